"""QE_TRACE of a few single quicked_align calls (1 kb BandEd score-only): where one call's time goes"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
lib = capi.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pairs = list(datagen.generate(60, L, 0.05, seed=3).pairs())
p = capi.make_params(algo=capi.BANDED, only_score=True)
for i, pt in enumerate(pairs):
    if i == 57:
        sys.stderr.write("==== traced calls\n"); sys.stderr.flush()
    a = capi.Aligner()
    lib.quicked_new(C.byref(a), C.byref(p))
    lib.quicked_align(C.byref(a), pt[0], len(pt[0]), pt[1], len(pt[1]))
    lib.quicked_free(C.byref(a))
