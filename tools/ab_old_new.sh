#!/bin/bash
# the same bench on the same box with two builds of the library: gpurun -- bash tools/ab_old_new.sh <tag> <old .so>
out=gpurun_out/$1; mkdir -p $out
old=$PWD/$2
sum() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
q=d['workloads']['quicked']; s=d['strong_share']
print(sys.argv[2], f\"banded {d['value']/1e6:.3f} M ({d['ms_per_step']:.2f} ms; solo {d['roofline']['kernel_ms']:.2f})  quicked {q['value']/1e6:.3f} M ({q['ms_per_step']:.2f} ms; fill solo {q['roofline']['kernel_ms']:.2f})\")
print(sys.argv[2], 'e2e banded', {k: round(v['value']/1e6,2) for k,v in d['e2e'].items() if isinstance(v,dict)}, 'quicked', {k: round(v['value']/1e6,2) for k,v in q['e2e'].items() if isinstance(v,dict)})
print(sys.argv[2], f\"share banded {s['banded_score']['value']/1e6:.2f} M single {s['banded_score']['single_batch_latency_ms']:.2f} ms; quicked {s['quicked']['value']/1e6:.2f} M single {s['quicked']['single_batch_latency_ms']:.2f} ms; indels {d['workloads']['quicked_indels']['value']/1e6:.3f} M\")
" $1 $2; }
for rep in 1 2; do
  QUICKED_HIP_LIB=$old python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/old_$rep.json 2> $out/old_$rep.err; sum $out/old_$rep.json old$rep
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/new_$rep.json 2> $out/new_$rep.err; sum $out/new_$rep.json new$rep
done | tee $out/summary.txt
