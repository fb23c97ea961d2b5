#!/bin/bash
# the driver's bench line + the parity suite: gpurun -- bash tools/ab_bench.sh <tag> [pytest -k expression]
out=gpurun_out/$1; mkdir -p $out
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
q=d['workloads']['quicked']; s=d['strong_share']
print(f"banded {d['value']/1e6:.3f} M ({d['ms_per_step']:.2f} ms; solo {d['roofline']['kernel_ms']:.2f})  quicked {q['value']/1e6:.3f} M ({q['ms_per_step']:.2f} ms; fill solo {q['roofline']['kernel_ms']:.2f})")
print("e2e banded", {k: round(v['value']/1e6,2) for k,v in d['e2e'].items() if isinstance(v,dict)}, "quicked", {k: round(v['value']/1e6,2) for k,v in q['e2e'].items() if isinstance(v,dict)})
print(f"share banded {s['banded_score']['value']/1e6:.2f} M single {s['banded_score']['single_batch_latency_ms']:.2f} ms; quicked {s['quicked']['value']/1e6:.2f} M single {s['quicked']['single_batch_latency_ms']:.2f} ms; indels {d['workloads']['quicked_indels']['value']/1e6:.3f} M")
PY
if [ -n "$2" ]; then timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" 2>&1 | tail -3; fi
