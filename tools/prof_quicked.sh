#!/bin/bash
# kernel-time summary of the QuickEd workload (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-q}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --workload quicked --no-cpu-baseline --sync-each-step > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('gpurun_out/prof_$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'qe::' in r['Name']: print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
