#!/bin/bash
# QuickEd + CIGAR on batches with a small share of large-indel pairs, early finish on and off, one process per point:
#   gpurun -- bash tools/probe_mixed.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2; do for share in 0.01 0.05; do for f in 3 0; do
  QE_FINISHERS=$f STEPS=36 python3 tools/probe_mixed.py 100000 $share 1 2>>$out/err.txt | sed "s/^/finishers $f: /" | tee -a $out/mixed_rates.txt
done; done; done
grep -c "out of memory" $out/err.txt
