#!/bin/bash
# workgroup placement at small grids: LDS pin that allows two workgroups per CU (54 KB) against one (84 KB)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-probe_pin}; mkdir -p $out
common="--no-cpu-baseline --no-e2e --no-strong --steps 40 --warmup 4"
rate() { python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['kernel_ms'])"; }
for pin in 55296 84000; do
  export QE_PIN_LDS=$pin
  for n in 12500 32000; do
    for g in default 1 2 4; do
      if [ $g = default ]; then unset QE_COOP_G; else export QE_COOP_G=$g; fi
      echo "== pin $pin banded_score pairs $n G $g" >> $out/rates.txt
      timeout 300 python3 bench.py --pairs $n $common 2>>$out/err.txt | rate >> $out/rates.txt
      echo "   solo:" >> $out/rates.txt
      timeout 300 python3 bench.py --pairs $n $common --sync-each-step 2>>$out/err.txt | rate >> $out/rates.txt
    done
  done
  unset QE_COOP_G
  for n in 12500 32000; do
    echo "== pin $pin quicked pairs $n" >> $out/rates.txt
    timeout 300 python3 bench.py --pairs $n --workload quicked $common 2>>$out/err.txt | rate >> $out/rates.txt
    echo "   solo:" >> $out/rates.txt
    timeout 300 python3 bench.py --pairs $n --workload quicked $common --sync-each-step 2>>$out/err.txt | rate >> $out/rates.txt
  done
done
