"""host-side stage timers of a few asynchronous runs (QE_TRACE=1): where the run call spends its time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
wl = sys.argv[2] if len(sys.argv) > 2 else "quicked"
p = capi.make_params(algo=capi.QUICKED) if wl == "quicked" else capi.make_params(algo=capi.BANDED, only_score=True)
b = datagen.generate(n, 10000, 0.05, seed=5)
rb = capi.ResidentBatch(b)
for _ in range(3):
    rb.run(p, sync=True)
for _ in range(14):
    rb.run(p, sync=False)
rb.sync()
t0 = time.perf_counter()
for _ in range(24):
    rb.run(p, sync=False)
t1 = time.perf_counter()
rb.sync()
t2 = time.perf_counter()
print(f"{wl} {n}: host {1e3 * (t1 - t0) / 24:.3f} ms per run call, {1e3 * (t2 - t0) / 24:.3f} ms per step incl. drain", file=sys.stderr)
os.environ["QE_TRACE"] = "1"
