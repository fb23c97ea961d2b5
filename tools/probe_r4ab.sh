#!/bin/bash
out=gpurun_out/r4ab; mkdir -p $out
for tall in 0 1; do
  QE_COOP_TALL_FILL=$tall timeout 300 python tools/probe_indel_overlap.py 20000 4 1,2 2>&1 | sed "s/^/tall $tall: /" >> $out/summary.txt
  QE_COOP_TALL_FILL=$tall STEPS=60 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/tall $tall: /" >> $out/summary.txt
  QE_COOP_TALL_FILL=$tall STEPS=96 SLOTS=14 timeout 300 python3 tools/probe_mixed.py 12500 0.01 1 2>/dev/null | sed "s/^/tall $tall: /" >> $out/summary.txt
  QE_COOP_TALL_FILL=$tall timeout 300 python bench.py --workload quicked --pairs 20000 --indels-num 4 --indels-len 800 --steps 8 --warmup 2 --no-workloads --no-strong --indel-pairs 0 --no-e2e --no-cpu-baseline --cfg4-pairs 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tall $tall: indel leg', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],1), 'ms; single', round(d['single_batch_latency_ms'],1))" >> $out/summary.txt
done
cat $out/summary.txt
