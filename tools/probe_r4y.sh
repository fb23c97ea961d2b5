#!/bin/bash
# the staged run stores of k_traceback: CIGAR parity, then QuickEd + CIGAR before / after on one box (round 3's library is the "before" for the rate;
# the kernel's own time from rocprofv3)
out=gpurun_out/r4y; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $out/parity.log 2>&1
echo "parity rc=$? $(tail -1 $out/parity.log)" > $out/summary.txt
one="--workload quicked --no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e --cfg4-pairs 0"
python3 bench.py $one --steps 20 --warmup 3 > $out/quicked.json 2> $out/quicked.err
python - $out/quicked.json >> $out/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("quicked", round(d["value"]/1e6, 3), "M", round(d["ms_per_step"], 2), "ms; single batch", round(d["single_batch_latency_ms"], 2), "ms")
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/solo -- python3 bench.py $one --steps 8 --warmup 3 --sync-each-step > $out/solo.log 2>&1
cp $out/solo/*/*kernel_stats.csv $out/quicked_solo_kernel_stats.csv; rm -rf $out/solo
cut -c1-120 $out/quicked_solo_kernel_stats.csv | head -9 >> $out/summary.txt
cat $out/summary.txt
