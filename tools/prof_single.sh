#!/bin/bash
# kernel + HIP API stats of the single-pair call: gpurun -- bash tools/prof_single.sh <tag> LENGTH banded|quicked
out=gpurun_out/$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; label=single_$2_$3
rm -rf /tmp/$label
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d /tmp/$label -o p -- python3 $R/tools/single_call_prof.py $2 $3 > /tmp/$label.log 2>&1
for k in kernel_stats hip_api_stats; do f=$(find /tmp/$label -name "*$k.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/${label}_$k.csv; done
tail -1 /tmp/$label.log
