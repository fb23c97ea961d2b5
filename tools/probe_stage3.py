"""QuickEd's stage 3 in isolation: score-only BandEd at the bandwidth cutoff (and doubled) on indel-heavy pairs whose
distance exceeds it -- how many tasks the cooperative kernel hands to the one-lane fallback, and what each form costs"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
b = datagen.generate(n, 10000, 0.05, seed=0x51CED, indels_num=4, indels_len=800)
rb = capi.ResidentBatch(b)
for bw in (15, 30, 60):
    p = capi.make_params(algo=capi.BANDED, only_score=True, bandwidth=bw)
    for _ in range(2):
        rb.run(p, sync=True)
    t0 = time.perf_counter()
    for _ in range(3):
        rb.run(p, sync=True)
    dt = (time.perf_counter() - t0) / 3
    c = rb.counters()
    s, st = rb.scores()
    print(f"QE_COOP_G={os.environ.get('QE_COOP_G', 'auto'):4s} n {n} bandwidth {bw:2d} %: {dt * 1e3:7.2f} ms per run, fallback tasks {int(c[6])}, "
          f"scores < 0 (band never reached the last block): {int((s < 0).sum())}, block advances {int(c[0]) / 1e9:.2f} G", flush=True)
rb.close()
