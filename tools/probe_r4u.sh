#!/bin/bash
out=gpurun_out/r4u; mkdir -p $out
for q in 14 16 20 24; do for w in A B; do GPU_MAX_HW_QUEUES=$q timeout 300 python3 tools/probe_leftover2.py $w 2>> $out/err.txt | sed "s/^/queues $q: /" >> $out/summary.txt; done; done
cat $out/summary.txt; tail -3 $out/err.txt
