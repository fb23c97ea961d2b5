#!/bin/bash
# what does k_traceback wait for?  Builds of the library whose k_traceback leaves one kind of memory access out
# (-DQE_TB_ELIDE=<bits>: 1 run stores, 2 checkpoint loads, 4 carry-word loads, 8 band edges + text / pattern planes; results are
# garbage, the kernel's duration alone on the chip is the datum).  Build them first:
#   for n in 1 2 4 8 14 15; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DQE_TB_ELIDE=$n -Iinclude \
#     -Iquicked_amd/csrc quicked_amd/csrc/qe_driver.hip -Wl,quicked_amd/qe_hostpack.o -Wl,quicked_amd/qe_capi.o -lpthread -o tools/bin/libq_elide_$n.so; done
out=gpurun_out/${1:-r4z}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one="--workload quicked --no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e --cfg4-pairs 0 --steps 6 --warmup 2 --sync-each-step"
for n in 0 1 2 4 8 14 15; do
  if [ $n = 0 ]; then unset QUICKED_HIP_LIB; else export QUICKED_HIP_LIB=$PWD/tools/bin/libq_elide_$n.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e$n -- python3 bench.py $one > $out/e$n.log 2>&1
  f=$(ls $out/e$n/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "elide $n: $(grep k_traceback $f | cut -d, -f2-4) | fill $(grep 'k_banded<true>' $f | cut -d, -f4) | format $(grep 'k_format_segs<true>' $f | cut -d, -f4)" >> $out/summary.txt
  rm -rf $out/e$n
done
cat $out/summary.txt
