"""One synchronous run of a SMALL batch, per stage form: WindowEd(2,1) score-only, QuickEd + CIGAR and BandEd score-only on
12.5 k / 4 k / 500 / 1 pairs of 10 kb, and config 4's stage 1 (10 k pairs of 100 kb), with the cooperative forms on and off
(QE_WINDOWED_QUAD).  Duration of one run = the latency a caller with exactly that many pairs sees."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
cases = [("WindowEd(2,1) score-only", dict(algo=capi.WINDOWED, window_size=2, overlap_size=1, only_score=True)),
         ("QuickEd + CIGAR", dict(algo=capi.QUICKED)),
         ("BandEd score-only", dict(algo=capi.BANDED, only_score=True))]
shapes = [(12500, 10000, 0.05), (4000, 10000, 0.05), (500, 10000, 0.05), (1, 10000, 0.05), (1000, 1000, 0.05)]
if "--cfg4" in sys.argv:
    shapes = [(10000, 100000, 0.10)]
for n, L, e in shapes:
    b = datagen.generate(n, L, e, seed=0x51CED)
    rb = capi.ResidentBatch(b)
    for name, kw in cases:
        if L > 50000 and name != "WindowEd(2,1) score-only":
            continue
        p = capi.make_params(**kw)
        for quad in ("0", "1"):
            os.environ["QE_WINDOWED_QUAD"] = quad
            for _ in range(2):
                rb.run(p, sync=True)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); rb.run(p, sync=True); ts.append(time.perf_counter() - t0)
            dt = min(ts)
            print(f"n {n:6d} x {L:6d} {name:26s} quad {quad}: {dt * 1e3:8.3f} ms per run  {n / dt / 1e6:7.3f} M pairs/s", flush=True)
    rb.close()
