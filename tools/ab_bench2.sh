#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
one="--no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e"
for wl in quicked; do
python bench.py --workload $wl $one --steps 20 --warmup 5 2>$out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],2), 'ms  solo', round(d['roofline']['kernel_ms'],2), 'overlapped', round(d['roofline']['kernel_ms_overlapped'],2), 'in flight', d['runs_in_flight'], 'checksum', d['score_checksum'])
"
done
python bench.py --workload quicked --pairs 10000 --length 100000 --error 0.1 --steps 10 --warmup 3 $one 2>>$out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg4', round(d['value']/1e3,2), 'k', round(d['ms_per_step'],1), 'ms  fill solo', round(d['roofline']['kernel_ms'],2))
"
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
