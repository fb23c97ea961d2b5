#!/bin/bash
# kernel stats of the indel-heavy QuickEd leg (stages 2 / 3): gpurun -- bash tools/profile_indels.sh <tag>
tag=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
args="--workload quicked --pairs 20000 --indels-num 4 --indels-len 800 --no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/solo -- python3 bench.py $args --steps 6 --warmup 2 --sync-each-step > $out/solo.log 2>&1
cp $out/solo/*/*kernel_stats.csv $out/${tag}_indels_solo_kernel_stats.csv
cp $out/solo/*/*kernel_trace.csv $out/${tag}_indels_solo_kernel_trace.csv
rm -rf $out/solo
QE_TRACE=1 python3 bench.py $args --steps 3 --warmup 2 --sync-each-step > $out/trace.json 2> $out/trace.err
tail -3 $out/solo.log
