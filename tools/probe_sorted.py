#!/usr/bin/env python3
"""Does the ORDER of the pairs in a batch matter?  A wave's 64 lanes walk the union of their bands, and k_banded<false> takes its
4-slot passes only where every lane has all four slots: lanes with the same band geometry (prolog and height follow from
plen - tlen and the cutoff, bpm_banded.c:121-135) agree more often.  The same 100 k pairs in generator order and sorted by
plen - tlen, BandEd score-only and QuickEd + CIGAR, a stream of queued runs each."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quicked_amd import capi, datagen
from quicked_amd.datagen import PairBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
b = datagen.generate(n, 10000, 0.05, seed=datagen.DEFAULT_SEED)
diff = b.pattern_len.astype(np.int64) - b.text_len.astype(np.int64)
orders = {"generator order": np.arange(n), "sorted by plen - tlen": np.argsort(diff, kind="stable"),
          "sorted by |plen - tlen|, sign": np.lexsort((np.abs(diff), diff < 0))}
for label, idx in orders.items():
    idx = np.ascontiguousarray(idx)
    pb = PairBatch(b.pattern_pool, np.ascontiguousarray(b.pattern_off[idx]), np.ascontiguousarray(b.pattern_len[idx]),
                   b.text_pool, np.ascontiguousarray(b.text_off[idx]), np.ascontiguousarray(b.text_len[idx]))
    rb = capi.ResidentBatch(pb)
    for name, kw in (("BandEd score-only", dict(algo=capi.BANDED, only_score=True, bandwidth=15)), ("QuickEd + CIGAR", dict(algo=capi.QUICKED))):
        p = capi.make_params(**kw)
        for _ in range(2):
            assert rb.run(p, sync=True) >= 0
        chk = int(rb.scores()[0].astype("int64").sum())
        for _ in range(6):
            assert rb.run(p, sync=False) >= 0
        rb.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            assert rb.run(p, sync=False) >= 0
        rb.sync()
        dt = time.perf_counter() - t0
        print(f"{label:32s} {name:18s}: {n * steps / dt / 1e6:.3f} M alignments/s ({dt / steps * 1e3:.2f} ms per step), score checksum {chk}", flush=True)
    rb.close()
    capi.pool_trim()
