#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-e2e_a}; mkdir -p $out
show() { python3 -c "
import sys,json
l=json.loads(sys.stdin.readlines()[-1])
print('  resident', round(l['value']), round(l['ms_per_step'],2))
for k,v in l['e2e'].items():
    if isinstance(v,dict): print('  ',k, round(v['value']), {a:round(b,1) for a,b in v.get('host_ms_per_batch',{}).items()}, 'inflight',v.get('inflight'),'slots',v.get('slots'),'upl',v.get('uploader_threads'))
"; }
for wl in banded_score quicked; do
  for cfg in ${CFGS:-"0 0 0" "6 3 3" "8 4 4"}; do
    set -- $cfg
    echo "== $wl slots $1 inflight $2 uploaders $3" >> $out/e2e.txt
    timeout 600 python3 bench.py --workload $wl --no-workloads --no-cpu-baseline --no-strong --steps 10 --warmup 2 --e2e-slots $1 --e2e-inflight $2 --e2e-uploaders $3 2>>$out/err.txt | show >> $out/e2e.txt
  done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $out/pytest.txt
