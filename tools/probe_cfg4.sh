#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-cfg4_a}; mkdir -p $out
args="--workload quicked --pairs 10000 --length 100000 --error 0.1 --no-workloads --no-cpu-baseline --no-e2e --no-strong"
for g in ${GS:-default}; do
  if [ $g = default ]; then unset QE_COOP_G; else export QE_COOP_G=$g; fi
  echo "== cfg4 G $g" >> $out/rates.txt
  timeout 600 python3 bench.py $args --steps 12 --warmup 3 2>>$out/err.txt | python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'], l['single_batch_latency_ms'], l['runs_in_flight'])" >> $out/rates.txt
done
unset QE_COOP_G
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -- python3 bench.py $args --steps 6 --warmup 3 > $out/tr.log 2>&1
python3 tools/timeline.py $out/tr --gantt 90 > $out/timeline.txt 2>&1
cp $out/tr/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/tr
