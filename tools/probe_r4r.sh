#!/bin/bash
out=gpurun_out/r4r; mkdir -p $out
for w in A B C; do timeout 300 python3 tools/probe_leftover2.py $w >> $out/summary.txt 2>> $out/err.txt; done
cat $out/summary.txt
