#!/bin/bash
out=gpurun_out/r4l; mkdir -p $out
for m in 1 4 12; do for slots in 8 14; do
  QE_FINISH_MERGE=$m STEPS=96 SLOTS=$slots timeout 300 python3 tools/probe_mixed.py 12500 0.01 1 2>/dev/null | sed "s/^/12.5k pairs merge $m slots $slots: /" >> $out/summary.txt
done; done
cat $out/summary.txt
