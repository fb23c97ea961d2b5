#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; mkdir -p $out
python3 tools/probe_two_threads.py 100000 quicked 1,2 8 | tee $out/rates.txt
python3 tools/probe_two_threads.py 100000 banded 1,2 8 | tee -a $out/rates.txt
python3 tools/probe_two_threads.py 50000 quicked 1,2 8 | tee -a $out/rates.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -- python3 tools/probe_two_threads.py 100000 quicked 2 8 > $out/tr.log 2>&1
cp $out/tr/*/*kernel_stats.csv $out/two_threads_kernel_stats.csv; cp $out/tr/*/*kernel_trace.csv $out/two_threads_kernel_trace.csv; rm -rf $out/tr
tail -2 $out/tr.log
