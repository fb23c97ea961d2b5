#!/bin/bash
out=gpurun_out/r4f; mkdir -p $out
QE_TRACE=1 STEPS=12 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 > $out/mixed.txt 2> $out/mixed_trace.err
STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 >> $out/mixed.txt 2>/dev/null
timeout 300 python - > $out/stream.txt 2>$out/stream.err <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from quicked_amd import capi, datagen
o = bench.mixed_leg(capi, datagen, 20000, 10000, 0.05, 1.0, steps=16, slots=6)
print({k: v for k, v in o.items() if k not in ('note', 'data')})
PY
cat $out/mixed.txt $out/stream.txt
