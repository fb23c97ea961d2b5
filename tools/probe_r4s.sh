#!/bin/bash
out=gpurun_out/r4s; mkdir -p $out
for w in twice events samebatch; do timeout 300 python3 tools/probe_leftover3.py $w >> $out/summary.txt 2>> $out/err.txt; done
cat $out/summary.txt
