#!/bin/bash
out=gpurun_out/r4o; mkdir -p $out
QE_TRACE_POOL=1 STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 > $out/mixed.txt 2> $out/pool.err
cat $out/mixed.txt
grep -c "" $out/pool.err
tail -40 $out/pool.err | cut -c1-150
