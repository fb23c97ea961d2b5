"""per-call cost of the drop-in single-pair path (quicked_new / quicked_align / quicked_free per pair,
as tools/align_benchmark/benchmark/benchmark_edit.c:45-87 does)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
lib = capi.lib()
for length, algo, only in ((1000, capi.QUICKED, False), (1000, capi.BANDED, True), (10000, capi.QUICKED, False), (10000, capi.BANDED, True)):
    pairs = list(datagen.generate(200, length, 0.05, seed=3).pairs())
    p = capi.make_params(algo=algo, only_score=only)
    def once(pt):
        a = capi.Aligner()
        lib.quicked_new(C.byref(a), C.byref(p))
        lib.quicked_align(C.byref(a), pt[0], len(pt[0]), pt[1], len(pt[1]))
        s = a.score
        lib.quicked_free(C.byref(a))
        return s
    once(pairs[0])
    t0 = time.perf_counter()
    for pt in pairs: once(pt)
    dt = (time.perf_counter() - t0) / len(pairs)
    print(f"len {length} algo {algo} only_score {only}: {dt*1e3:.3f} ms per new+align+free")
