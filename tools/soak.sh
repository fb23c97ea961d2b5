#!/bin/bash
# the two soaks of tests/ (not collected by pytest) on the GPU box: gpurun --timeout 1500 -- bash tools/soak.sh <tag> <first seed> <fuzz seeds> <long seeds>
out=gpurun_out/$1; mkdir -p $out
timeout 1200 python tests/soak_fuzz.py $2 $3 > $out/soak_fuzz.txt 2>&1
timeout 1200 python tests/soak_long.py $2 $4 > $out/soak_long.txt 2>&1
grep -h "MISMATCH\|^soak" $out/soak_fuzz.txt $out/soak_long.txt | tail -20
