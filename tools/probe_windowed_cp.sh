#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_datasets or oracle_parity_random or ragged or randomised or counters or stage1 or stage3 or kats or long_reads or sam_cigar or packed_wire" > $out/tests.txt 2>&1
tail -5 $out/tests.txt
for cp in 0 1; do echo "== QE_WINDOWED_CP=$cp"; QE_WINDOWED_CP=$cp python tools/probe_windowed_n.py 2>&1 | grep "9,1"; done | tee $out/rates.txt
