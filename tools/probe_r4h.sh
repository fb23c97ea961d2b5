#!/bin/bash
out=gpurun_out/r4h; mkdir -p $out
for f in 1 2; do for slots in 4 6 8; do
  QE_FINISHERS=$f QE_FINISH_MERGE=6 STEPS=24 SLOTS=$slots timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/finishers $f merge 6 slots $slots: /" >> $out/summary.txt
done; done
QE_FINISHERS=3 QE_FINISH_MERGE=1 STEPS=24 SLOTS=4 QE_TRACE=1 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>$out/trace_f3.err | sed "s/^/finishers 3 merge 1 slots 4 (traced): /" >> $out/summary.txt
cat $out/summary.txt
