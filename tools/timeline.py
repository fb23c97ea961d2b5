#!/usr/bin/env python3
"""Per-launch kernel durations and, with --gantt N, the start / end of the last N launches relative to the first of
them (which kernels of consecutive runs overlap, where the device idles) from a rocprofv3 --kernel-trace CSV directory."""
import collections
import csv
import glob
import sys


def main():
    f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if 'qe::' in r['Kernel_Name']]
    d = collections.defaultdict(list)
    for r in rows:
        d[r['Kernel_Name'][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    for k, v in d.items():
        print(f"{k:46s} n={len(v):4d} min {min(v):7.3f} med {sorted(v)[len(v)//2]:7.3f} max {max(v):7.3f}  | " + " ".join(f"{x:.1f}" for x in v[:10]))
    if '--gantt' in sys.argv:
        n = int(sys.argv[sys.argv.index('--gantt') + 1])
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        last = rows[-n:]
        t0 = int(last[0]['Start_Timestamp'])
        busy, cur_end, span0 = 0, None, None
        for r in last:
            s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
            q = r.get('Queue_Id', '?')
            print(f"  {s/1e6:9.3f} -> {e/1e6:9.3f} ms  ({(e-s)/1e6:7.3f})  q{q:>3s}  {r['Kernel_Name'][:60]}")
            if cur_end is None or s > cur_end:
                busy += 0 if cur_end is None else 0
                if cur_end is not None:
                    busy += cur_end - span0
                span0, cur_end = s, e
            else:
                cur_end = max(cur_end, e)
        busy += cur_end - span0
        total = max(int(r['End_Timestamp']) for r in last) - t0
        print(f"  device busy {busy/1e6:.3f} of {total/1e6:.3f} ms ({100.0*busy/total:.1f} %)")


if __name__ == '__main__':
    main()
