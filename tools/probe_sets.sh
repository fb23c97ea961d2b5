#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-sets_a}; mkdir -p $out
common="--no-cpu-baseline --no-e2e --no-strong --no-workloads --indel-pairs 0 --steps 40 --warmup 2"
rate() { python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'], l['runs_in_flight'])"; }
for np in 3 4 5; do
  for wl in quicked banded_score; do
    echo "== np_min $np $wl 100000" >> $out/rates.txt
    QE_NP_MIN=$np timeout 300 python3 bench.py --workload $wl $common 2>>$out/err.txt | rate >> $out/rates.txt
  done
done
