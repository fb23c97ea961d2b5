#!/bin/bash
# wave priority (s_setprio 3 in the cooperative kernels, QE_WAVE_PRIO) on the mixed stream, A / B alternating:
#   gpurun --timeout 900 -- bash tools/probe_prio.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2 3; do for w in 0 1; do
  echo "== QE_WAVE_PRIO=$w" | tee -a $out/prio.txt
  QE_WAVE_PRIO=$w STEPS=${STEPS:-96} timeout 300 python tools/probe_mixed.py 100000 0.01 3 2>&1 | tail -1 | tee -a $out/prio.txt
done; done
