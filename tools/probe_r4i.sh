#!/bin/bash
out=gpurun_out/r4i; mkdir -p $out
for tb in 0 32 64; do for pin in 48 16; do
  [ $tb = 0 ] && [ $pin = 16 ] && continue
  QE_DBG_TINY_BLOCKS=$tb QE_DBG_TINY_PIN=$pin QE_FINISH_MERGE=1 STEPS=24 SLOTS=4 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/tiny $tb pin $pin slots 4: /" >> $out/summary.txt
  QE_DBG_TINY_BLOCKS=$tb QE_DBG_TINY_PIN=$pin QE_FINISH_MERGE=1 STEPS=24 SLOTS=6 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/tiny $tb pin $pin slots 6: /" >> $out/summary.txt
done; done
QE_DBG_TINY_BLOCKS=32 QE_DBG_TINY_PIN=48 QE_FINISH_MERGE=4 QE_FINISHERS=2 STEPS=24 SLOTS=6 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/tiny 32 pin 48 merge 4 finishers 2 slots 6: /" >> $out/summary.txt
cat $out/summary.txt
