#!/bin/bash
# threads that end with their pools allocated, many times over (the thread_local destructor used to call into HIP): gpurun -- bash tools/probe_exit_crash.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2 3 4 5 6; do
  QE_SEGV_TRACE=1 python3 tools/probe_indel_threads.py 20000 4 10000 0.05 4 1,2,4 windowed9 > $out/o_$rep.txt 2> $out/e_$rep.txt; echo "rep $rep rc $?"
done
cat $out/e_*.txt | head -40
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "threads or two_host or concurrent" 2>&1 | tail -3
