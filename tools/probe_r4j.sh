#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
QE_FINISH_MERGE=1 STEPS=12 SLOTS=4 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/probe_mixed.py 100000 0.01 1 > $out/mixed.txt 2> $out/mixed.err
cat $out/mixed.txt
ls -R $out/trace | head
f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
python3 - $f > $out/gantt.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'qe::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-700:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e6:9.3f} {e/1e6:9.3f} {(e-s)/1e6:8.3f} q{r.get('Queue_Id','?'):>3s} s{r.get('Stream_Id','?'):>3s} g{r.get('Grid_Size','?'):>8s} {r['Kernel_Name'][:48]}")
PY
wc -l $out/gantt.txt
rm -rf $out/trace
