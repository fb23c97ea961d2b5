#!/bin/bash
# the indel-heavy leg (20 k pairs of 10 kb, 4 x 800-base indels) under the default selection and with each cooperative form forced:
#   gpurun --timeout 900 -- bash tools/probe_indel_forms.sh <tag>
out=gpurun_out/$1; mkdir -p $out
run() { # label, env...
  label=$1; shift
  v=$(env "$@" timeout 300 python bench.py --workload quicked --pairs ${PAIRS:-20000} --indels-num 4 --indels-len 800 --steps 10 --warmup 2 \
      --no-e2e --no-cpu-baseline --no-strong --no-workloads 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.3f M/s  %.2f ms/step  latency %.2f ms' % (d['value'] / 1e6, d['ms_per_step'], d.get('single_batch_latency_ms', 0)))")
  echo "$label: $v" | tee -a $out/indel_forms.txt
}
run default X=1
run windowed_sys QE_WINDOWED_SYS=1
run fill_sys QE_FILL_SYS=1
run score_sys QE_SCORE_SYS=1
run trace_sys4 QE_TRACE_SYS=4
run all_sys QE_WINDOWED_SYS=1 QE_FILL_SYS=1 QE_SCORE_SYS=1 QE_TRACE_SYS=4
run all_off QE_WINDOWED_SYS=0 QE_FILL_SYS=0 QE_SCORE_SYS=0 QE_TRACE_SYS=0 QE_STAGE3_DEVICE=0
