"""few-alignment regime of score-only BandEd: one wavefront per alignment (k_banded_wave) against the lane-per-alignment /
cooperative kernels, resident batches, kernel + launch + score download per run"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
p = capi.make_params(algo=capi.BANDED, only_score=True)
for length in (1000, 10000):
    for n in (1, 16, 64, 256, 1000, 1500, 2000, 4000, 8000):
        b = datagen.generate(n, length, 0.05, seed=5)
        rb = capi.ResidentBatch(b)
        res = {}
        for mode in ("0", "1"):
            os.environ["QE_WAVE"] = mode
            for _ in range(4):                      # every stream / pool set of the rotation has allocated
                rb.run(p, sync=True)
            s0 = rb.scores()[0].copy()
            reps = 20 if n * length <= 4_000_000 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                rb.run(p, sync=True)
            res[mode] = (time.perf_counter() - t0) / reps
            assert (rb.scores()[0] == s0).all()
        del os.environ["QE_WAVE"]
        rb.close()
        print(f"len {length:6d} pairs {n:5d}: lane/coop {res['0']*1e3:8.3f} ms ({n/res['0']:10.0f}/s)   wave {res['1']*1e3:8.3f} ms ({n/res['1']:10.0f}/s)", flush=True)
