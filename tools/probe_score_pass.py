#!/usr/bin/env python3
"""QuickEd with only_score: the align step (fill, traceback, edit count) against one score pass over the fill's cells
(QE_QUICKED_SCORE_PASS = 0 / 1), a stream of queued runs and one run alone, at several batch sizes.  The scores of the two
must be the same array."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quicked_amd import capi, datagen

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 2048, 4096, 8192, 12500, 25000, 50000, 100000]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
length = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
error = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
for n in sizes:
    b = datagen.generate(n, length, error, seed=datagen.DEFAULT_SEED)
    rb = capi.ResidentBatch(b)
    ref = None
    coop = [("1", "1", "coop%d" % g) for g in (2, 4, 8)] if os.environ.get("PROBE_COOP") else []
    for mode, fast, sys_ in [("0", "1", None), (None, "1", None), ("1", "0", None), ("1", "1", None), ("1", "1", "0")] + coop:      # (sys_ "0": also QE_COOP_FILL_G = 1, no cooperative form)
        os.environ.pop("QE_QUICKED_SCORE_PASS", None)
        if mode is not None:
            os.environ["QE_QUICKED_SCORE_PASS"] = mode       # None: the library's own choice; 1: the pass wherever the results allow it
        os.environ["QE_QUICKED_SCORE_PASS_FAST"] = fast      # 0: synchronous runs take the pass at the end of the host-driven flow
        os.environ.pop("QE_SCORE_SYS", None); os.environ.pop("QE_COOP_FILL_G", None)
        os.environ.pop("QE_SCORE_PASS_COOP_G", None)
        if sys_ is not None and sys_.startswith("coop"):
            os.environ["QE_SCORE_SYS"] = "0"; os.environ["QE_SCORE_PASS_COOP_G"] = sys_[4:]      # the pass in the cooperative LDS form
        elif sys_ is not None:
            os.environ["QE_SCORE_SYS"] = sys_                # 0: one lane per alignment whatever the launch's size
            os.environ["QE_COOP_FILL_G"] = "1"
        capi.reload_env()
        p = capi.make_params(algo=capi.QUICKED, only_score=True)
        for _ in range(2):
            assert rb.run(p, sync=True) >= 0
        sc = rb.scores()[0].copy()
        if ref is None: ref = sc
        same = bool(np.array_equal(ref, sc))
        ctr = [int(x) for x in rb.counters()] if hasattr(rb, "counters") else None
        lat = []
        for _ in range(5):
            t0 = time.perf_counter(); assert rb.run(p, sync=True) >= 0; lat.append(time.perf_counter() - t0)
        for _ in range(6):
            assert rb.run(p, sync=False) >= 0
        rb.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            assert rb.run(p, sync=False) >= 0
        rb.sync()
        dt = time.perf_counter() - t0
        print(f"{n:7d} pairs of {length}, score pass {mode} (sync runs in the fast flow {fast}, QE_SCORE_SYS {sys_}): stream {n * steps / dt / 1e6:.3f} M alignments/s, alone {min(lat) * 1e3:.2f} ms; "
              f"scores identical to the align step's: {same}; counters {ctr}", flush=True)
    rb.close()
    capi.pool_trim()
