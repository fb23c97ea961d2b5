#!/bin/bash
# config 4 with T host threads (each its own batches of 10 000 / T pairs), per LDS-pin policy: gpurun -- bash tools/probe_cfg4_threads.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for env in "" "QE_PIN_CHAIN=0" "QE_PIN_LDS=55296"; do
  echo "== ${env:-default}" >> $out/t.txt
  env $env python3 tools/probe_indel_threads.py 10000 4 100000 0.1 0 1,2 quicked >> $out/t.txt 2>> $out/t.err
done
cat $out/t.txt; tail -3 $out/t.err
