#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-coop_b}; mkdir -p $out
for tall in 3 2 1; do
  echo "=== QE_COOP_TALL $tall" >> $out/probe.txt
  QE_COOP_TALL=$tall timeout 600 python3 tools/probe_coop.py 4000:10000:0.05 12500:10000:0.05 32000:10000:0.05 2000:100000:0.10 2>&1 | grep -v "G    1:\|G    2:" >> $out/probe.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $out/pytest.txt
