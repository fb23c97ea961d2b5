#!/bin/bash
out=gpurun_out/r4k; mkdir -p $out
for g in 0 2 4 8 16; do
  if [ $g = 0 ]; then timeout 300 python tools/probe_indel_overlap.py 20000 4 1 2>&1 | sed "s/^/default: /" >> $out/summary.txt
  else QE_COOP_FILL_G=$g timeout 300 python tools/probe_indel_overlap.py 20000 4 1 2>&1 | sed "s/^/fill G=$g: /" >> $out/summary.txt; fi
done
for g in 4 8 16; do
  QE_COOP_FILL_G=$g timeout 300 python tools/probe_indel_overlap.py 1000 4 1 2>&1 | sed "s/^/1000 pairs fill G=$g: /" >> $out/summary.txt
done
timeout 300 python tools/probe_indel_overlap.py 1000 4 1 2>&1 | sed "s/^/1000 pairs default: /" >> $out/summary.txt
cat $out/summary.txt
