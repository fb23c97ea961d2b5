#!/bin/bash
# one workload with several builds of the library on one box: gpurun -- bash tools/ab_libs.sh <tag> <workload> <lib> [<lib> ...]   ("default" = the in-tree build)
out=gpurun_out/$1; mkdir -p $out; wl=$2; shift; shift
one="--no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e"
for rep in 1 2; do for lib in "$@"; do
  if [ $lib = default ]; then unset QUICKED_HIP_LIB; else export QUICKED_HIP_LIB=$PWD/$lib; fi
  python3 bench.py --workload $wl $one --steps 20 --warmup 5 2>>$out/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],2), 'ms  solo', round(d['roofline']['kernel_ms'],2), 'checksum', d['score_checksum'])
"
done; done | tee $out/rates.txt
