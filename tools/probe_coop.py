"""single-batch latency of score-only BandEd per cooperative width and kernel form (k_banded_coop_lds / k_banded_coop)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quicked_amd import capi, datagen
p = capi.make_params(algo=capi.BANDED, only_score=True)
shapes = [(1000, 10000, 0.05), (4000, 10000, 0.05), (12500, 10000, 0.05), (32000, 10000, 0.05), (2000, 100000, 0.10)]
if len(sys.argv) > 1:
    shapes = [tuple(float(x) if "." in x else int(x) for x in a.split(":")) for a in sys.argv[1:]]
for n, length, err in shapes:
    b = datagen.generate(int(n), int(length), err, seed=5)
    rb = capi.ResidentBatch(b)
    ref = None
    for lds in ("1", "0"):
        os.environ["QE_COOP_LDS"] = lds
        for G in ("", "1", "2", "4", "8", "16", "32", "64"):
            if G == "1" and lds == "0":
                continue
            if G:
                os.environ["QE_COOP_G"] = G
            else:
                os.environ.pop("QE_COOP_G", None)
            os.environ["QE_WAVE"] = "0"
            for _ in range(3):
                rb.run(p, sync=True)
            s = rb.scores()[0].copy()
            if ref is None:
                ref = s
            ok = bool((s == ref).all())
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); rb.run(p, sync=True); ts.append(time.perf_counter() - t0)
            c = rb.counters()
            dt = sorted(ts)[len(ts) // 2]
            print(f"n {n:6d} len {length:6d} lds {lds} G {G or 'auto':>4s}: {dt * 1e3:8.3f} ms  {n / dt / 1e6:7.3f} M/s  {c[0] / dt / 1e12:6.3f}e12 bc/s  fallback tasks {int(c[6])}  same {ok}", flush=True)
    rb.close()
