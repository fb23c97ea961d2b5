"""WindowEd(9,1) score-only and a tall-band BandEd fill on indel-heavy pairs: duration of one synchronous run against the
number of pairs (latency-bound launches keep their duration while the chip has idle SIMDs; throughput-bound ones do not)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
cases = [("WindowEd(9,1) score-only", capi.make_params(algo=capi.WINDOWED, window_size=9, overlap_size=1, only_score=True)),
         ("WindowEd(9,1) + CIGAR", capi.make_params(algo=capi.WINDOWED, window_size=9, overlap_size=1)),
         ("WindowEd(4,2) score-only", capi.make_params(algo=capi.WINDOWED, window_size=4, overlap_size=2, only_score=True)),
         ("WindowEd(2,1) score-only", capi.make_params(algo=capi.WINDOWED, window_size=2, overlap_size=1, only_score=True))]
for n, indels in ((500, 4), (15000, 4), (100000, 4), (100000, 0)):
    b = datagen.generate(n, 10000, 0.05, seed=0x51CED, indels_num=indels, indels_len=800 if indels else 0)
    rb = capi.ResidentBatch(b)
    for name, p in cases:
        for _ in range(2):
            rb.run(p, sync=True)
        t0 = time.perf_counter()
        for _ in range(3):
            rb.run(p, sync=True)
        dt = (time.perf_counter() - t0) / 3
        print(f"n {n:6d} indels {indels} {name:26s}: {dt * 1e3:8.2f} ms per run  {n / dt / 1e6:7.3f} M pairs/s", flush=True)
    rb.close()
