import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
from quicked_amd import capi, datagen
lib = capi.lib()
length = int(sys.argv[1]); algo = int(sys.argv[2]); only = bool(int(sys.argv[3]))
pairs = list(datagen.generate(60, length, 0.05, seed=3).pairs())
p = capi.make_params(algo=algo, only_score=only)
def once(pt):
    a = capi.Aligner()
    lib.quicked_new(C.byref(a), C.byref(p))
    lib.quicked_align(C.byref(a), pt[0], len(pt[0]), pt[1], len(pt[1]))
    lib.quicked_free(C.byref(a))
for pt in pairs[:50]: once(pt)

for pt in pairs[50:53]:
    t0 = time.perf_counter(); once(pt); print("call ms", (time.perf_counter()-t0)*1e3, file=sys.stderr)
