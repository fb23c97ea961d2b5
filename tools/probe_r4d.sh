#!/bin/bash
# round 4: the lane-relative band walk (QE_LANE_REL) before / after on the workloads whose lanes' bands lie apart, and whether
# host-driven QuickEd chains of different threads overlap.  Logs under gpurun_out/r4d
out=gpurun_out/r4d; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lane_relative or fill_multi or golden_datasets or randomised or counters" > $out/tests.log 2>&1
echo "tests rc=$? $(tail -1 $out/tests.log)" > $out/summary.txt
for rel in 0 1; do
  QE_LANE_REL=$rel timeout 300 python tools/probe_indel_overlap.py 20000 4 1,2,3,4 > $out/overlap_rel$rel.txt 2>&1
  QE_LANE_REL=$rel timeout 300 python bench.py --steps 10 --warmup 2 --no-e2e --no-strong --cfg4-pairs 0 --no-cpu-baseline > $out/bench_rel$rel.json 2> $out/bench_rel$rel.err
  QE_LANE_REL=$rel timeout 300 python bench.py --workload quicked --pairs 10000 --length 100000 --error 0.1 --steps 8 --warmup 2 --no-workloads --no-strong --indel-pairs 0 --no-e2e --no-cpu-baseline > $out/cfg4_rel$rel.json 2> $out/cfg4_rel$rel.err
done
for f in 3 5; do for slots in 4 7; do
  QE_FINISHERS=$f SLOTS=$slots STEPS=24 timeout 300 python3 tools/probe_mixed.py 100000 0.01 1 2>>$out/mixed_err.txt | sed "s/^/finishers $f slots $slots: /" >> $out/mixed_rates.txt
done; done
python - <<'PY' >> gpurun_out/r4d/summary.txt
import json
for rel in (0, 1):
    try:
        d = json.loads(open(f"gpurun_out/r4d/bench_rel{rel}.json").read().strip().splitlines()[-1])
        w = d["workloads"]
        print(f"rel={rel}: banded {d['value']/1e6:.2f} M quicked {w['quicked']['value']/1e6:.2f} M ({w['quicked']['ms_per_step']:.2f} ms) indels {w['quicked_indels']['value']/1e6:.3f} M ({w['quicked_indels']['ms_per_step']:.1f} ms) stream {w['quicked_indels'].get('fetched_stream',{}).get('value',0)/1e6:.3f} M mixed {w['quicked_mixed'].get('value',0)/1e6:.2f} M")
    except Exception as e:
        print("bench", rel, repr(e))
    try:
        d = json.loads(open(f"gpurun_out/r4d/cfg4_rel{rel}.json").read().strip().splitlines()[-1])
        print(f"rel={rel}: cfg4 {d['value']/1e3:.1f} k ({d['ms_per_step']:.1f} ms)")
    except Exception as e:
        print("cfg4", rel, repr(e))
PY
cat $out/overlap_rel0.txt $out/overlap_rel1.txt $out/mixed_rates.txt >> $out/summary.txt
for tag in full nocfg4; do
  extra=""; [ $tag = nocfg4 ] && extra="--cfg4-pairs 0"
  ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e $extra > $out/line_$tag.json 2> $out/line_$tag.err ) 2> $out/line_${tag}_time.txt
  python - $out/line_$tag.json $tag <<'PY' >> gpurun_out/r4d/summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); w = d["workloads"]; s = d["strong_share"]
print(sys.argv[2], "share", round(s["banded_score"]["value"]/1e6,2), round(s["quicked"]["value"]/1e6,2), "mixed", round(w["quicked_mixed"].get("value",0)/1e6,2),
      "indel", round(w["quicked_indels"]["value"]/1e6,3), "stream", round(w["quicked_indels"].get("fetched_stream",{}).get("value",0)/1e6,3), "cfg4", round(w.get("cfg4",{}).get("value",0)/1e3,1))
PY
done
cat $out/summary.txt
