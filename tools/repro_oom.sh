#!/bin/bash
# round 4: the GPU suite in file order N times, then the thread-churn test alone M times, with the library's stderr kept (-s) and a
# native backtrace on abort (tools/segv_trace.c); logs under gpurun_out/$1
out=gpurun_out/${1:-r4}; N=${2:-5}; M=${3:-30}
mkdir -p $out
export LD_PRELOAD=$PWD/tools/bin/libsegvtrace.so
for i in $(seq 1 $N); do
  timeout 1200 python -m pytest tests -q -m gpu -p no:faulthandler -s > $out/suite_$i.log 2>&1
  echo "suite $i rc=$? $(tail -1 $out/suite_$i.log)" >> $out/summary.txt
done
for i in $(seq 1 $M); do
  QE_TRACE_POOL=1 timeout 300 python -m pytest tests/test_gpu_pools.py -q -m gpu -p no:faulthandler -s -k "threads_that_end" > $out/thr_$i.log 2>&1
  echo "thread test $i rc=$? $(tail -1 $out/thr_$i.log)" >> $out/summary.txt
done
cat $out/summary.txt
