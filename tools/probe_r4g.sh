#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
export LD_PRELOAD=$PWD/tools/bin/libsegvtrace.so
timeout 900 python -m pytest tests/test_gpu_pools.py -q -m gpu -p no:faulthandler -s > $out/pools.log 2>&1
echo "pools rc=$? $(tail -1 $out/pools.log)" > $out/summary.txt
unset LD_PRELOAD
for m in 4 1; do
  QE_FINISH_MERGE=$m STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/merge $m: /" >> $out/summary.txt
  QE_FINISH_MERGE=$m STEPS=24 SLOTS=6 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/merge $m slots 6: /" >> $out/summary.txt
  QE_FINISH_MERGE=$m timeout 300 python - >> $out/summary.txt 2>$out/stream_$m.err <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from quicked_amd import capi, datagen
o = bench.mixed_leg(capi, datagen, 20000, 10000, 0.05, 1.0, steps=16, slots=6)
print("indel stream merge", os.environ["QE_FINISH_MERGE"], {k: v for k, v in o.items() if k in ('value', 'ms_per_batch', 'error')}, capi.early_finish_stats())
PY
done
cat $out/summary.txt
