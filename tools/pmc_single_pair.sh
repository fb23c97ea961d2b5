#!/bin/bash
# SQ counters of the kernels of ONE 10 kb QuickEd pair (a lone wave per kernel): instructions, cycles, share issuing / waiting
out=gpurun_out/$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/sp1 -o p -- python3 $R/tools/run_shape.py 1 10000 0.05 quicked 3 > /tmp/sp1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/sp2 -o p -- python3 $R/tools/run_shape.py 1 10000 0.05 quicked 3 > /tmp/sp2.log 2>&1
cp $(find /tmp/sp1 -name "*counter_collection.csv" | head -1) $R/$out/single_pair_pmc_sq1.csv
cp $(find /tmp/sp2 -name "*counter_collection.csv" | head -1) $R/$out/single_pair_pmc_sq2.csv

