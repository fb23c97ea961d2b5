"""N synchronous runs of one batch shape (for rocprofv3 --kernel-trace --stats):
python3 tools/run_shape.py PAIRS LENGTH ERROR ALGO [RUNS [INDELS_NUM INDELS_LEN]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
n, L, e, algo = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 6
kw = dict(quicked=dict(algo=capi.QUICKED), banded=dict(algo=capi.BANDED, only_score=True),
          windowed=dict(algo=capi.WINDOWED, window_size=2, overlap_size=1, only_score=True))[algo]
ind = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 0)
b = datagen.generate(n, L, e, seed=0x51CED, indels_num=ind[0], indels_len=ind[1])
rb = capi.ResidentBatch(b)
p = capi.make_params(**kw)
ts = []
for _ in range(runs):
    t0 = time.perf_counter(); rb.run(p, sync=True); ts.append(time.perf_counter() - t0)
print(f"{n} x {L} {algo}: min {min(ts) * 1e3:.3f} ms per run")
rb.close()
