#!/bin/bash
out=gpurun_out/r4t; mkdir -p $out
for w in A B; do QUICKED_HIP_LIB=$PWD/tools/bin/libquicked_hip_r03.so timeout 300 python3 tools/probe_leftover2.py $w 2>> $out/err.txt | sed "s/^/r03 lib: /" >> $out/summary.txt; done
cat $out/summary.txt; tail -3 $out/err.txt
