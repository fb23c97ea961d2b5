#!/bin/bash
# one lane or one wave per alignment in the CIGAR formatter (QE_FORMAT_WAVE), by read length and batch size
out=gpurun_out/$1; mkdir -p $out
for shape in "1000 100000" "1000 12500" "3000 100000" "300 100000"; do set -- $shape
  for f in default 1; do
    if [ $f = default ]; then unset QE_FORMAT_WAVE; else export QE_FORMAT_WAVE=1; fi
    v=$(timeout 300 python bench.py --workload quicked --length $1 --pairs $2 --steps 30 --warmup 4 --no-e2e --no-cpu-baseline --no-strong --no-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.3f M/s  alone %.2f ms' % (d['value'] / 1e6, d.get('single_batch_latency_ms', 0)))")
    echo "length $1 pairs $2 QE_FORMAT_WAVE=$f: $v" | tee -a $out/format_wave.txt
  done
done
