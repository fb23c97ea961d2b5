#!/bin/bash
# SQ counters of config 4's kernels, each alone on the chip: gpurun -- bash tools/pmc_cfg4_sq.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
one="--workload quicked --pairs 10000 --length 100000 --error 0.1 --no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e --steps 1 --warmup 0 --sync-each-step"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $out/pmc -- python3 bench.py $one > $out/pmc.log 2>&1
cp $out/pmc/*/*counter_collection.csv $out/cfg4_sq.csv; rm -rf $out/pmc
python3 - $out/cfg4_sq.csv <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); seen=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'qe::' not in k: continue
    k=k.replace('void ','').replace('qe::','').split('(')[0]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
for k,v in acc.items():
    d=len(seen[k])
    if v['SQ_WAVE_CYCLES']<1e8: continue
    print(f"{k[:30]:30s} launches {d:3d}: VALU {v['SQ_INSTS_VALU']/d/1e9:7.3f} G/launch  LDS {v['SQ_INSTS_LDS']/d/1e9:6.3f} G  waves {v['SQ_WAVES']/d:8.0f}  issuing {v['SQ_ACTIVE_INST_VALU']/v['SQ_WAVE_CYCLES']:.3f}  wait_any {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.3f}  wait_inst {v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f}")
PY
