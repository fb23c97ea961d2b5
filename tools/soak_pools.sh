#!/bin/bash
# the pool / thread test file N times in a row (fresh process each), stderr kept, slowest tests listed; summary under gpurun_out/$1
out=gpurun_out/${1:-soak}; N=${2:-8}; mkdir -p $out
export LD_PRELOAD=$PWD/tools/bin/libsegvtrace.so
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests/test_gpu_pools.py -q -m gpu -p no:faulthandler -s --durations=3 > $out/pools_$i.log 2>&1
  echo "pools $i rc=$? $(tail -1 $out/pools_$i.log) | $(grep -A3 'slowest 3' $out/pools_$i.log | tail -3 | awk '{print $1, $3}' | tr '\n' ' ')" >> $out/summary.txt
done
cat $out/summary.txt
