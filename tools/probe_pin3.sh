#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
one="--no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e"
for pin in 0 53248 40960; do
  for wl in quicked banded_score; do
    if [ $pin = 0 ]; then unset QE_PIN_LDS; else export QE_PIN_LDS=$pin; fi
    python bench.py --workload $wl $one --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pin $pin $wl', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],2), 'ms  solo', round(d['roofline']['kernel_ms'],2), 'overlapped', round(d['roofline']['kernel_ms_overlapped'],2), 'in flight', d['runs_in_flight'])
"
  done
done | tee $out/rates.txt
