#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
one="--workload quicked --no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e --steps 1 --warmup 0 --sync-each-step"
for m in 0 1; do
  export QE_FILL_MULTI=$m
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_$m -- python3 bench.py $one > $out/pmc_$m.log 2>&1
  cp $out/pmc_$m/*/*counter_collection.csv $out/fill_multi_${m}_pmc.csv; rm -rf $out/pmc_$m
  python3 - $out/fill_multi_${m}_pmc.csv $m <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'k_banded<true>' not in k and 'k_traceback' not in k and 'k_windowed' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(k,r['Dispatch_Id']); 
    if key not in seen: seen.add(key); n[k]+=1
for k,v in acc.items():
    d=n[k]
    print(f"multi {sys.argv[2]} {k[:34]:34s} launches {d}: VALU insts {v['SQ_INSTS_VALU']/d/1e9:.3f} G  wave cycles {v['SQ_WAVE_CYCLES']/d/1e9:.3f} G  issuing {v['SQ_ACTIVE_INST_VALU']/v['SQ_WAVE_CYCLES']:.3f}  wait_any {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.3f}  wait_inst {v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f}")
PY
done | tee $out/summary.txt
unset QE_FILL_MULTI
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "planner_cuts or two_host or stage1 or bench_times" 2>&1 | tail -3
