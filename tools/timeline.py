#!/usr/bin/env python3
"""Print per-launch kernel durations from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'qe::' in r['Kernel_Name']]
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name'][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in d.items():
    print(f"{k:46s} n={len(v):3d} min {min(v):7.3f} med {sorted(v)[len(v)//2]:7.3f} max {max(v):7.3f}  | " + " ".join(f"{x:.1f}" for x in v[:14]))
