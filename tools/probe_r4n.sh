#!/bin/bash
# A/B on one box: round 3's library against this one on the mixed probe; then the full bench line with the stream retirement
out=gpurun_out/r4n; mkdir -p $out
for rep in 1 2; do
  QUICKED_HIP_LIB=$PWD/tools/bin/libquicked_hip_r03.so STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/r03 lib: /" >> $out/summary.txt
  STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/r04 lib: /" >> $out/summary.txt
  QE_LANE_REL=0 STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>/dev/null | sed "s/^/r04 lib, lane_rel 0: /" >> $out/summary.txt
done
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err ) 2> $out/bench_line_time.txt
python - $out/bench_line.json >> $out/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); w = d["workloads"]; s = d["strong_share"]
print("line: head", round(d["value"]/1e6,2), "quicked", round(w["quicked"]["value"]/1e6,2), "share", round(s["banded_score"]["value"]/1e6,2), round(s["quicked"]["value"]/1e6,2), round(s.get("quicked_mixed",{}).get("value",0)/1e6,2),
      "mixed", round(w["quicked_mixed"].get("value",0)/1e6,2), "indel", round(w["quicked_indels"]["value"]/1e6,3), "stream", round(w["quicked_indels"].get("fetched_stream",{}).get("value",0)/1e6,3), "cfg4", round(w.get("cfg4",{}).get("value",0)/1e3,1))
PY
grep real $out/bench_line_time.txt >> $out/summary.txt
cat $out/summary.txt
