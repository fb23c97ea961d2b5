#!/bin/bash
# kernel trace of a stream of 12.5 k-pair BandEd runs: how many launches are on the device at a time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
one="--no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --workload ${2:-banded_score} --pairs 12500 $one --steps 60 --warmup 12 > $out/tr.log 2>&1
cp $out/tr/*/*kernel_trace.csv $out/share_kernel_trace.csv; rm -rf $out/tr
python3 - $out/share_kernel_trace.csv <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[r for r in rows if 'k_banded<false>' in r['Kernel_Name'] and int(r['End_Timestamp'])-int(r['Start_Timestamp'])>1e6]
big=big[-60:]
t0=int(big[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in big)
ev=[]
for r in big: ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort()
cur=0; last=t0; area=0; hist={}
for t,d in ev:
    area+=cur*(t-last); hist[cur]=hist.get(cur,0)+(t-last); last=t; cur+=d
print(f"{len(big)} launches over {(t1-t0)/1e6:.1f} ms; mean concurrency {area/(t1-t0):.2f}")
print("time share by number of k_banded launches on the device:", {k: round(v/(t1-t0),3) for k,v in sorted(hist.items())})
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in big]
print(f"launch duration min {min(d):.2f} median {sorted(d)[len(d)//2]:.2f} max {max(d):.2f} ms")
qs={}
for r in big: qs[r['Queue_Id']]=qs.get(r['Queue_Id'],0)+1
print("queues:", qs)
for r in big[-14:]:
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e6:8.2f} -> {(int(r['End_Timestamp'])-t0)/1e6:8.2f}  q{r['Queue_Id']}")
PY
