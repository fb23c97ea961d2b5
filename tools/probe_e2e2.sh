#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-e2e2}; mkdir -p $out
show() { python3 -c "
import sys,json
l=json.loads(sys.stdin.readlines()[-1])
print('  resident', round(l['value']), round(l['ms_per_step'],2), l['runs_in_flight'])
for k,v in l['e2e'].items():
    if isinstance(v,dict): print('  ',k, round(v['value']), {a:round(b,1) for a,b in v.get('host_ms_per_batch',{}).items()}, 'sets', v.get('pool_sets'))
"; }
for d in ${DS:-5 5 5}; do
  echo "== QE_DEPTH_CHAIN $d" >> $out/e2e.txt
  QE_DEPTH_CHAIN=$d timeout 600 python3 bench.py --workload quicked --no-workloads --no-cpu-baseline --no-strong --indel-pairs 0 --steps 20 --warmup 5 2>>$out/err.txt | show >> $out/e2e.txt
done
