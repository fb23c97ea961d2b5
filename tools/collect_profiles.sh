#!/bin/bash
# Collects what profiles/ holds for one milestone (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <tag> [banded_score quicked cfg4 share indels]   -> gpurun_out/<tag>/...
# kernel-trace/stats and every PMC counter set in separate passes, as MI355X_MICROARCH.md prescribes.
tag=${1:-r03_x}; shift
wls=${@:-banded_score quicked cfg4 share indels}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}      # the repository's root on the GPU box (or wherever the script lies)
cd /tmp && export TMPDIR=/tmp; cd "$R"
out=gpurun_out/$tag; mkdir -p $out
one="--no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e"
for wl in $wls; do
  case $wl in
    banded_score) args="--workload banded_score"; steps=10;;
    quicked)      args="--workload quicked"; steps=10;;
    cfg4)         args="--workload quicked --pairs 10000 --length 100000 --error 0.1"; steps=8;;
    share)        args="--workload banded_score --pairs 12500"; steps=40;;
    indels)       args="--workload quicked --pairs 20000 --indels-num 4 --indels-len 800"; steps=6;;
  esac
  common="$args $one"
  # overlapped (as benchmarked) and solo (--sync-each-step: a kernel's own duration) kernel stats
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 bench.py $common --steps $steps --warmup 3 > $out/stats_$wl.log 2>&1
  cp $out/stats_$wl/*/*kernel_stats.csv $out/${tag}_${wl}_kernel_stats.csv
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/solo_$wl -- python3 bench.py $common --steps $steps --warmup 3 --sync-each-step > $out/solo_$wl.log 2>&1
  cp $out/solo_$wl/*/*kernel_stats.csv $out/${tag}_${wl}_solo_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 bench.py $common --steps 1 --warmup 0 --sync-each-step > $out/pmc_${wl}_$c.log 2>&1
    cp $out/pmc_${wl}_$c/*/*counter_collection.csv $out/${tag}_${wl}_pmc_$c.csv
  done
  rm -rf $out/stats_$wl $out/solo_$wl $out/pmc_${wl}_FETCH_SIZE $out/pmc_${wl}_WRITE_SIZE
done
# SQ / GRBM counters of the BandEd score kernel alone (one pass per slot budget: 8 SQ, 2 GRBM)
if [[ " $wls " == *" banded_score "* ]]; then
  b="--workload banded_score $one --steps 1 --warmup 0 --sync-each-step"
  timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq1 -- python3 bench.py $b > $out/pmc_sq1.log 2>&1
  cp $out/pmc_sq1/*/*counter_collection.csv $out/${tag}_banded_score_pmc_sq1.csv
  timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_grbm -- python3 bench.py $b > $out/pmc_grbm.log 2>&1
  cp $out/pmc_grbm/*/*counter_collection.csv $out/${tag}_banded_score_pmc_grbm.csv
  rm -rf $out/pmc_sq1 $out/pmc_grbm
fi
# SQ wait / issue shares of the QuickEd kernels (WindowEd, fill, traceback), each alone on the chip
if [[ " $wls " == *" quicked "* ]]; then
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_qsq1 -- python3 bench.py --workload quicked $one --steps 1 --warmup 0 --sync-each-step > $out/pmc_qsq1.log 2>&1
  cp $out/pmc_qsq1/*/*counter_collection.csv $out/${tag}_quicked_pmc_sq1.csv
  rm -rf $out/pmc_qsq1
  # what the QuickEd kernels' waves wait for (k_traceback): memory instructions by kind, the time spent
  # with a memory instruction outstanding, L2 hits / misses -- separate passes
  q="--workload quicked $one --steps 1 --warmup 0 --sync-each-step"
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $out/pmc_qsq2 -- python3 bench.py $q > $out/pmc_qsq2.log 2>&1
  cp $out/pmc_qsq2/*/*counter_collection.csv $out/${tag}_quicked_pmc_sq2.csv
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_qtcc -- python3 bench.py $q > $out/pmc_qtcc.log 2>&1
  cp $out/pmc_qtcc/*/*counter_collection.csv $out/${tag}_quicked_pmc_tcc.csv
  timeout 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/pmc_qtcp -- python3 bench.py $q > $out/pmc_qtcp.log 2>&1
  cp $out/pmc_qtcp/*/*counter_collection.csv $out/${tag}_quicked_pmc_tcp.csv
  rm -rf $out/pmc_qsq2 $out/pmc_qtcc $out/pmc_qtcp
fi
./tools/bin/valu_rate ABC > $out/${tag}_valu_rates.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/calib_$c -- tools/bin/pmc_calib > $out/calib_$c.log 2>&1
  cp $out/calib_$c/*/*counter_collection.csv $out/${tag}_calib_pmc_$c.csv
  rm -rf $out/calib_$c
done
python3 tools/summarise_pmc.py $out $tag $wls > /dev/null
cp $out/pmc_traffic_latest.json profiles/pmc_traffic_latest.json      # (on the box's copy: the line below prints the traffic of THESE kernels)
# the full line (what the driver runs) and config 4 on its own
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_line.json 2> $out/bench_line.err ) 2> $out/${tag}_bench_line_time.txt
python3 bench.py --workload quicked --pairs 10000 --length 100000 --error 0.1 --steps 20 --warmup 3 --no-workloads --no-strong --indel-pairs 0 > $out/${tag}_bench_cfg4.json 2> $out/bench_cfg4.err
python3 tools/single_call_latency.py > $out/${tag}_single_call_latency.txt 2> $out/single_call_latency.err
# WindowEd on its own (the WINDOWED algorithm's default shape is 9 / 1: k_windowed_cp), history path against checkpoint path
for cp in 0 1; do echo "== QE_WINDOWED_CP=$cp"; QE_WINDOWED_CP=$cp python3 tools/probe_windowed_n.py 2>&1; done > $out/${tag}_windowed_rates.txt
