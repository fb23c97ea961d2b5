#!/bin/bash
# kernel stats of one batch shape: gpurun -- bash tools/prof_shape.sh <tag> <label> <run_shape args...>   (env passes through)
out=gpurun_out/$1; label=$2; shift 2; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ps_$label
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$label -o p -- python3 $R/tools/run_shape.py "$@" > /tmp/ps_$label.log 2>&1
cp $(find /tmp/ps_$label -name "*kernel_stats.csv" | head -1) $R/$out/${label}_kernel_stats.csv
tail -1 /tmp/ps_$label.log
