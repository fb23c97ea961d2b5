#!/bin/bash
# usage: gpurun -- bash tools/probe_indel_threads.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for w in windowed9 banded40c; do
  echo "== $w" >> $out/t.txt
  python3 tools/probe_indel_threads.py 20000 4 10000 0.05 4 1,2,4 $w >> $out/t.txt 2>> $out/t.err
done
echo "== windowed9 QE_PIN_LDS=1024" >> $out/t.txt
QE_PIN_LDS=1024 python3 tools/probe_indel_threads.py 20000 4 10000 0.05 4 1,2,4 windowed9 >> $out/t.txt 2>> $out/t.err
cat $out/t.txt; tail -3 $out/t.err
