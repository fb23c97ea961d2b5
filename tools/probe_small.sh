#!/bin/bash
# small-batch regime (the per-GPU share of "100 k pairs over 8 GPUs"): rates per cooperative width, kernel timelines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-probe_small}; mkdir -p $out
common="--no-cpu-baseline --no-e2e --no-strong --steps 40 --warmup 4"
for n in 12500 32000; do
  for g in default 1 2 4 8; do
    if [ $g = default ]; then unset QE_COOP_G; else export QE_COOP_G=$g; fi
    echo "== banded_score pairs $n G $g" >> $out/rates.txt
    timeout 300 python3 bench.py --pairs $n $common 2>>$out/err.txt | python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'])" >> $out/rates.txt
  done
done
unset QE_COOP_G
echo "== quicked pairs 12500" >> $out/rates.txt
timeout 300 python3 bench.py --pairs 12500 --workload quicked $common 2>>$out/err.txt | python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'])" >> $out/rates.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr_banded -- python3 bench.py --pairs 12500 --no-cpu-baseline --no-e2e --no-strong --steps 12 --warmup 4 > $out/tr_banded.log 2>&1
python3 tools/timeline.py $out/tr_banded --gantt 40 > $out/timeline_banded.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr_quicked -- python3 bench.py --pairs 12500 --workload quicked --no-cpu-baseline --no-e2e --no-strong --steps 12 --warmup 4 > $out/tr_quicked.log 2>&1
python3 tools/timeline.py $out/tr_quicked --gantt 60 > $out/timeline_quicked.txt 2>&1
rm -rf $out/tr_banded $out/tr_quicked
