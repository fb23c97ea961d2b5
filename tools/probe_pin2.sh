#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-probe_pin2}; mkdir -p $out
common="--no-cpu-baseline --no-e2e --no-strong --no-workloads --steps 60 --warmup 2"
rate() { python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['kernel_ms_overlapped'], l['runs_in_flight'], l['single_batch_latency_ms'])"; }
for pin in 0 84000 70000; do
  if [ $pin = 0 ]; then unset QE_PIN_LDS; else export QE_PIN_LDS=$pin; fi
  for wl in quicked banded_score; do
    for n in 12500 32000; do
      for na in 0 8; do
        if [ $na = 0 ]; then unset QE_NA; else export QE_NA=$na; fi
        echo "== pin $pin $wl pairs $n QE_NA $na" >> $out/rates.txt
        timeout 300 python3 bench.py --pairs $n --workload $wl $common 2>>$out/err.txt | rate >> $out/rates.txt
      done
    done
  done
done
