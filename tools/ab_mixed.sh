#!/bin/bash
# the mixed stream (1 % hard pairs) with two builds of the library on one box, alternating:
#   gpurun -- bash tools/ab_mixed.sh <tag> <lib | default> <lib | default>
out=gpurun_out/$1; mkdir -p $out; shift
for rep in 1 2 3; do for lib in "$@"; do
  if [ $lib = default ]; then unset QUICKED_HIP_LIB; else export QUICKED_HIP_LIB=$PWD/$lib; fi
  for n in 100000 12500; do
    st=60; [ $n = 12500 ] && st=240
    SLOTS=14 STEPS=$st timeout 300 python3 tools/probe_mixed.py $n 0.01 1 2>>$out/err.txt | sed "s|^|$lib $n: |" | tee -a $out/ab_mixed.txt
  done
done; done
