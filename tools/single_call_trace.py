"""QE_TRACE of one quicked_new / quicked_align / quicked_free call after a warm-up: python tools/single_call_trace.py LENGTH banded|quicked"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
length, algo = int(sys.argv[1]), sys.argv[2]
from quicked_amd import capi, datagen
lib = capi.lib()
pairs = list(datagen.generate(64, length, 0.05, seed=3).pairs())
p = capi.make_params(algo=capi.BANDED if algo == "banded" else capi.QUICKED, only_score=algo == "banded")
for i, pt in enumerate(pairs):
    if i == 60:
        sys.stderr.flush(); os.write(2, b"==== traced calls\n")
    a = capi.Aligner()
    lib.quicked_new(C.byref(a), C.byref(p))
    lib.quicked_align(C.byref(a), pt[0], len(pt[0]), pt[1], len(pt[1]))
    lib.quicked_free(C.byref(a))
