#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
for q in 16 18; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/line_$q.json 2> $out/line_$q.err
  python - $out/line_$q.json $q >> $out/summary.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); w = d["workloads"]; s = d["strong_share"]
print("queues", sys.argv[2], "head", round(d["value"]/1e6,2), "quicked", round(w["quicked"]["value"]/1e6,2), "share", round(s["banded_score"]["value"]/1e6,2), round(s["quicked"]["value"]/1e6,2), round(s.get("quicked_mixed",{}).get("value",0)/1e6,2),
      "mixed", round(w["quicked_mixed"].get("value",0)/1e6,2), "indel", round(w["quicked_indels"]["value"]/1e6,3), "stream", round(w["quicked_indels"].get("fetched_stream",{}).get("value",0)/1e6,3), "cfg4", round(w.get("cfg4",{}).get("value",0)/1e3,1),
      "e2e", {k: round(v["value"]/1e6,2) for k, v in d["e2e"].items() if isinstance(v, dict)}, {k: round(v["value"]/1e6,2) for k, v in w["quicked"]["e2e"].items() if isinstance(v, dict)})
PY
done
cat $out/summary.txt
