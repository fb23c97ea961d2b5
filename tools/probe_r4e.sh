#!/bin/bash
# round 4 diagnosis: why the small-batch stream and the early-finish legs lost rate with the new pool code
out=gpurun_out/r4e; mkdir -p $out
share() { python bench.py --pairs 12500 --steps 160 --warmup 1 --no-e2e --no-strong --no-workloads --no-cpu-baseline 2>$out/share_$1.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,3), 'M/s', round(d['ms_per_step'],3), 'ms rif', d['runs_in_flight'])"; }
share base >> $out/summary.txt
QE_DBG_NOSTASHWAIT=1 share nostashwait >> $out/summary.txt
QE_DBG_NOEVLAST=1 share noevlast >> $out/summary.txt
QE_DBG_NOSTASHWAIT=1 QE_DBG_NOEVLAST=1 share neither >> $out/summary.txt
QE_TRACE_POOL=1 STEPS=12 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 > $out/mixed.txt 2> $out/mixed_pool.err
cat $out/mixed.txt >> $out/summary.txt
QE_DBG_NOSTASHWAIT=1 QE_DBG_NOEVLAST=1 STEPS=12 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed 's/^/neither: /' >> $out/summary.txt
grep -c "release_all\|reclaim" $out/mixed_pool.err >> $out/summary.txt
cat $out/summary.txt
