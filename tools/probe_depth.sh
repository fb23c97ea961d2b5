#!/bin/bash
# rates per batch size with the rotation depth the planner picks (and forced depths), per number of hardware queues
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-probe_depth}; mkdir -p $out
common="--no-cpu-baseline --no-e2e --no-strong --no-workloads --steps 60 --warmup 2"
rate() { python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['kernel_ms_overlapped'], l['runs_in_flight'], l['single_batch_latency_ms'])"; }
for hq in ${HQS:-16 32 64}; do
  export GPU_MAX_HW_QUEUES=$hq
  for wl in banded_score quicked; do
    for n in ${NS:-12500 32000 100000}; do
      for na in ${NAS:-0 6}; do
        if [ $na = 0 ]; then unset QE_NA; else export QE_NA=$na; fi
        echo "== hwq $hq $wl pairs $n QE_NA $na" >> $out/rates.txt
        timeout 300 python3 bench.py --pairs $n --workload $wl $common 2>>$out/err.txt | rate >> $out/rates.txt
      done
    done
  done
done
unset QE_NA; export GPU_MAX_HW_QUEUES=${TRACE_HQ:-16}
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr_quicked -- python3 bench.py --pairs 12500 --workload quicked $common > $out/tr_quicked.log 2>&1
python3 tools/timeline.py $out/tr_quicked --gantt 700 | head -300 | tail -120 > $out/timeline_quicked.txt 2>&1
rm -rf $out/tr_quicked
