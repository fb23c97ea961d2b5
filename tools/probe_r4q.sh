#!/bin/bash
out=gpurun_out/r4q; mkdir -p $out
timeout 600 python3 tools/probe_leftover.py > $out/summary.txt 2> $out/err.txt
cat $out/summary.txt; tail -3 $out/err.txt
