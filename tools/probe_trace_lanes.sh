#!/bin/bash
# lanes per leaf in the traceback (QE_TRACE_SYS: default selection, 16, 8, 4, 0) on streams of small batches and on one batch alone
out=gpurun_out/$1; mkdir -p $out
for n in 12500 4000 25000; do for t in default 16 8 0; do
  if [ $t = default ]; then unset QE_TRACE_SYS; else export QE_TRACE_SYS=$t; fi
  v=$(timeout 300 python bench.py --workload quicked --pairs $n --steps 120 --warmup 6 --no-e2e --no-cpu-baseline --no-strong --no-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.3f M/s  %.2f ms/step  alone %.2f ms' % (d['value'] / 1e6, d['ms_per_step'], d.get('single_batch_latency_ms', 0)))")
  echo "pairs $n QE_TRACE_SYS=$t: $v" | tee -a $out/trace_lanes.txt
done; done
