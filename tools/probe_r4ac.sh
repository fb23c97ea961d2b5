#!/bin/bash
bash tools/soak_pools.sh r4soak2 6
out=gpurun_out/r4soak2
export LD_PRELOAD=$PWD/tools/bin/libsegvtrace.so
for i in 1 2; do
  timeout 1200 python -m pytest tests -q -m gpu -p no:faulthandler -s > $out/suite_$i.log 2>&1
  echo "suite $i rc=$? $(tail -1 $out/suite_$i.log)" >> $out/summary.txt
done
unset LD_PRELOAD
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
