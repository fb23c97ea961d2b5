#!/bin/bash
out=gpurun_out/r4m; mkdir -p $out
export LD_PRELOAD=$PWD/tools/bin/libsegvtrace.so
timeout 1200 python -m pytest tests -q -m gpu -p no:faulthandler -s > $out/suite.log 2>&1
echo "suite rc=$? $(tail -1 $out/suite.log)" > $out/summary.txt
unset LD_PRELOAD
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err ) 2> $out/bench_line_time.txt
echo "bench rc=$? $(grep real $out/bench_line_time.txt)" >> $out/summary.txt
cat $out/summary.txt
