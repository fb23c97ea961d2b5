"""300 quicked_new / quicked_align / quicked_free calls of one shape (for rocprofv3 --kernel-trace --hip-trace --stats):
python3 tools/single_call_prof.py LENGTH banded|quicked"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
length, algo = int(sys.argv[1]), sys.argv[2]
from quicked_amd import capi, datagen
lib = capi.lib()
pairs = list(datagen.generate(300, length, 0.05, seed=3).pairs())
p = capi.make_params(algo=capi.BANDED if algo == "banded" else capi.QUICKED, only_score=algo == "banded")
ts = []
for pt in pairs:
    t0 = time.perf_counter()
    a = capi.Aligner()
    lib.quicked_new(C.byref(a), C.byref(p))
    lib.quicked_align(C.byref(a), pt[0], len(pt[0]), pt[1], len(pt[1]))
    lib.quicked_free(C.byref(a))
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[40:])
print(f"{length} {algo}: median {ts[len(ts) // 2] * 1e3:.3f} ms per call")
