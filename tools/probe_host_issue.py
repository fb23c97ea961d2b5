"""host time of one quicked_batch_run(sync = 0) call against the step time of a stream of such runs: is a stream of small
batches bound by the host's issue rate?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
for n in (12500, 32000, 100000):
    b = datagen.generate(n, 10000, 0.05, seed=0x51CED)
    rb = capi.ResidentBatch(b)
    for name, p in (("banded_score", capi.make_params(algo=capi.BANDED, only_score=True)), ("quicked", capi.make_params(algo=capi.QUICKED))):
        rb.run(p, sync=True)
        for _ in range(24):
            rb.run(p, sync=False)
        rb.sync()
        calls = []
        t0 = time.perf_counter()
        for _ in range(48):
            c0 = time.perf_counter(); rb.run(p, sync=False); calls.append(time.perf_counter() - c0)
        t_issue = time.perf_counter() - t0
        rb.sync()
        dt = time.perf_counter() - t0
        calls.sort()
        print(f"n {n:6d} {name:12s}: step {dt / 48 * 1e3:6.3f} ms ({n * 48 / dt / 1e6:5.2f} M/s)  run call median {calls[24] * 1e3:6.3f} ms  min {calls[0] * 1e3:6.3f}  "
              f"all 48 issued after {t_issue * 1e3:7.1f} ms of {dt * 1e3:7.1f}", flush=True)
    rb.close()
