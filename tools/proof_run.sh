#!/bin/bash
# round 4, item 1 of VERDICT r03: the driver's bench command, then the GPU suite in file order 5 times and the thread-churn
# test alone 30 times in ONE lease; logs under gpurun_out/$1 (summary.txt is what profiles/ keeps)
out=gpurun_out/${1:-r4c}
mkdir -p $out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err ) 2> $out/bench_line_time.txt
echo "bench rc=$? $(grep real $out/bench_line_time.txt)" >> $out/summary.txt
bash tools/repro_oom.sh ${1:-r4c} ${2:-5} ${3:-30}
