#!/bin/bash
# Collects what profiles/ holds for one milestone (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <tag> [banded_score quicked cfg4]   -> gpurun_out/<tag>/...
# kernel-trace/stats and every PMC counter set in separate passes, as MI355X_MICROARCH.md prescribes.
tag=${1:-r02_x}; shift
wls=${@:-banded_score quicked cfg4}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
for wl in $wls; do
  case $wl in
    banded_score) args="--workload banded_score"; steps=10;;
    quicked)      args="--workload quicked"; steps=10;;
    cfg4)         args="--workload quicked --pairs 10000 --length 100000 --error 0.1"; steps=10;;
  esac
  common="$args --no-cpu-baseline --no-e2e"
  # overlapped (as benchmarked) and solo (--sync-each-step: a kernel's own duration) kernel stats
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 bench.py $common --steps $steps --warmup 3 > $out/stats_$wl.log 2>&1
  cp $out/stats_$wl/*/*kernel_stats.csv $out/${tag}_${wl}_kernel_stats.csv
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/solo_$wl -- python3 bench.py $common --steps $steps --warmup 3 --sync-each-step > $out/solo_$wl.log 2>&1
  cp $out/solo_$wl/*/*kernel_stats.csv $out/${tag}_${wl}_solo_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 bench.py $common --steps 1 --warmup 0 --sync-each-step > $out/pmc_${wl}_$c.log 2>&1
    cp $out/pmc_${wl}_$c/*/*counter_collection.csv $out/${tag}_${wl}_pmc_$c.csv
  done
  python3 bench.py $args --steps $steps --warmup 3 > $out/${tag}_bench_$wl.json 2> $out/bench_$wl.err
done
# SQ / GRBM counters of the BandEd score kernel alone (one pass per slot budget: 8 SQ, 2 GRBM)
if [[ " $wls " == *" banded_score "* ]]; then
  timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq1 -- python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0 --sync-each-step > $out/pmc_sq1.log 2>&1
  cp $out/pmc_sq1/*/*counter_collection.csv $out/${tag}_banded_score_pmc_sq1.csv
  timeout 900 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $out/pmc_sq2 -- python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0 --sync-each-step > $out/pmc_sq2.log 2>&1
  cp $out/pmc_sq2/*/*counter_collection.csv $out/${tag}_banded_score_pmc_sq2.csv
  timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_grbm -- python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0 --sync-each-step > $out/pmc_grbm.log 2>&1
  cp $out/pmc_grbm/*/*counter_collection.csv $out/${tag}_banded_score_pmc_grbm.csv
fi
# SQ wait / issue shares of the QuickEd kernels (WindowEd, fill, traceback), each alone on the chip
if [[ " $wls " == *" quicked "* ]]; then
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_qsq1 -- python3 bench.py --workload quicked --no-cpu-baseline --no-e2e --no-strong --steps 1 --warmup 0 --sync-each-step > $out/pmc_qsq1.log 2>&1
  cp $out/pmc_qsq1/*/*counter_collection.csv $out/${tag}_quicked_pmc_sq1.csv
fi
./tools/bin/valu_rate ABC > $out/${tag}_valu_rates.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/calib_$c -- tools/bin/pmc_calib > $out/calib_$c.log 2>&1
  cp $out/calib_$c/*/*counter_collection.csv $out/${tag}_calib_pmc_$c.csv
done
python3 tools/summarise_pmc.py $out $tag $wls
