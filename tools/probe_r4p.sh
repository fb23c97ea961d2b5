#!/bin/bash
out=gpurun_out/r4p; mkdir -p $out
for rep in 1 2; do
  QUICKED_HIP_LIB=$PWD/tools/bin/libquicked_hip_r03.so STEPS=120 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/r03 lib, 120 steps: /" >> $out/summary.txt
  STEPS=120 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/r04 lib, 120 steps: /" >> $out/summary.txt
done
QE_FINISH_MERGE=1 STEPS=120 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/r04 lib, merge 1, 120 steps: /" >> $out/summary.txt
cat $out/summary.txt
