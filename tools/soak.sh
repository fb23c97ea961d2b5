#!/bin/bash
# the two soaks of tests/ (not collected by pytest) on the GPU box, default kernel forms and forced ones:
#   gpurun --timeout 2700 -- bash tools/soak.sh <tag> <first seed> <fuzz seeds> <long seeds>
out=gpurun_out/$1; mkdir -p $out
timeout 1200 python tests/soak_fuzz.py $2 $3 > $out/soak_fuzz.txt 2>&1
timeout 1200 python tests/soak_long.py $2 $4 > $out/soak_long.txt 2>&1
n=$(( $3 / 3 )); m=$(( $4 / 3 ))
QE_WINDOWED_CP=0 QE_FILL_MULTI=0 timeout 900 python tests/soak_fuzz.py $(( $2 + 1000 )) $n > $out/soak_fuzz_history_single.txt 2>&1
QE_COOP_G=1 QE_QUICKED_EST=40 timeout 900 python tests/soak_fuzz.py $(( $2 + 2000 )) $n > $out/soak_fuzz_onelane_smallest.txt 2>&1
QE_FILL_MULTI=0 timeout 900 python tests/soak_long.py $(( $2 + 1000 )) $m > $out/soak_long_single.txt 2>&1
QE_WINDOWED_CP=0 timeout 900 python tests/soak_long.py $(( $2 + 2000 )) $m > $out/soak_long_history.txt 2>&1
# round 5's cooperative forms: forced wherever eligible (with 8 and 4 lanes per leaf in the traceback), and switched off
QE_WINDOWED_QUAD=1 QE_WINDOWED_SYS=1 QE_FILL_SYS=1 QE_SCORE_SYS=1 QE_TRACE_SYS=1 timeout 900 python tests/soak_fuzz.py $(( $2 + 3000 )) $n > $out/soak_fuzz_sys_forced.txt 2>&1
QE_WINDOWED_QUAD=1 QE_WINDOWED_SYS=1 QE_FILL_SYS=1 QE_SCORE_SYS=1 QE_TRACE_SYS=8 timeout 900 python tests/soak_long.py $(( $2 + 3000 )) $m > $out/soak_long_sys_forced.txt 2>&1
QE_TRACE_SYS=4 QE_LANE_REL=2 timeout 900 python tests/soak_long.py $(( $2 + 4000 )) $m > $out/soak_long_trace4.txt 2>&1
QE_WINDOWED_QUAD=0 QE_WINDOWED_SYS=0 QE_FILL_SYS=0 QE_SCORE_SYS=0 QE_TRACE_SYS=0 QE_STAGE3_DEVICE=0 QE_FORMAT_WAVE=0 QE_WAVE_PRIO=0 timeout 900 python tests/soak_fuzz.py $(( $2 + 5000 )) $n > $out/soak_fuzz_sys_off.txt 2>&1
# round 6's wave formatter (64 consecutive runs per step, segmented scan) on every alignment, however few its runs
QE_FORMAT_WAVE=1 timeout 900 python tests/soak_fuzz.py $(( $2 + 6000 )) $n > $out/soak_fuzz_format_wave.txt 2>&1
QE_FORMAT_WAVE=1 timeout 900 python tests/soak_long.py $(( $2 + 6000 )) $m > $out/soak_long_format_wave.txt 2>&1
# round 6, late: QuickEd with only_score from one score pass over the fill's cells (the default): at the end of the host-driven
# flow only, with a small forced estimate; and switched off (the align step)
QE_QUICKED_SCORE_PASS_FAST=0 QE_QUICKED_EST=40 timeout 900 python tests/soak_fuzz.py $(( $2 + 7000 )) $n > $out/soak_fuzz_score_pass_classic.txt 2>&1
QE_QUICKED_SCORE_PASS=0 timeout 900 python tests/soak_fuzz.py $(( $2 + 8000 )) $n > $out/soak_fuzz_score_pass_off.txt 2>&1
grep -h "MISMATCH\|^soak" $out/soak_*.txt | tail -30
