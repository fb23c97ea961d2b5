"""host-side stage timers (QE_TRACE=1) of config-4 runs: where the host sits between the Hirschberg levels"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
p = capi.make_params(algo=capi.QUICKED)
b = datagen.generate(10000, 100000, 0.10, seed=0x51CED)
rb = capi.ResidentBatch(b)
for _ in range(3):
    rb.run(p, sync=True)
print("=== async runs", file=sys.stderr, flush=True)
t0 = time.perf_counter()
for _ in range(4):
    rb.run(p, sync=False)
    print(f"--- run call returned at +{1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
rb.sync()
print(f"--- all done at +{1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
