"""per-call cost of the drop-in single-pair path (quicked_new / quicked_align / quicked_free per pair, as
tools/align_benchmark/benchmark/benchmark_edit.c:45-87 does), and BASELINE config 1 (1000 pairs of 1 kb, BandEd
score-only) through that per-pair loop and through one batch call.  Median and minimum over the calls after a warm-up."""
import os, sys, time, ctypes as C, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
lib = capi.lib()


def once(p, pt):
    a = capi.Aligner()
    lib.quicked_new(C.byref(a), C.byref(p))
    lib.quicked_align(C.byref(a), pt[0], len(pt[0]), pt[1], len(pt[1]))
    s = a.score
    lib.quicked_free(C.byref(a))
    return s


for length, algo, only in ((1000, capi.BANDED, True), (1000, capi.QUICKED, False), (10000, capi.BANDED, True), (10000, capi.QUICKED, False),
                           (1000, capi.QUICKED, True), (10000, capi.QUICKED, True)):
    pairs = list(datagen.generate(300, length, 0.05, seed=3).pairs())
    p = capi.make_params(algo=algo, only_score=only)
    for pt in pairs[:40]:
        once(p, pt)                                       # warm: code objects, pools, clocks
    ts = []
    for pt in pairs[40:]:
        t0 = time.perf_counter(); once(p, pt); ts.append(time.perf_counter() - t0)
    print(f"len {length:6d} {('BandEd score-only' if algo == capi.BANDED else 'QuickEd score-only') if only else 'QuickEd + CIGAR  '}: median {statistics.median(ts)*1e3:7.3f} ms  min {min(ts)*1e3:7.3f} ms per new+align+free")

# BASELINE config 1: 1000 pairs of 1 kb at 5 %, BandEd score-only
b = datagen.generate(1000, 1000, 0.05)
pairs = list(b.pairs())
p = capi.make_params(algo=capi.BANDED, only_score=True)
for pt in pairs[:50]:
    once(p, pt)
t0 = time.perf_counter()
s_loop = [once(p, pt) for pt in pairs]
t_loop = time.perf_counter() - t0
al = capi.QuickedAligner(); al.setAlgorithm(capi.BANDED); al.setOnlyScore(True)
al.alignBatch(pairs)
t0 = time.perf_counter(); st, out = al.alignBatch(pairs); t_batch = time.perf_counter() - t0
assert [o[1] for o in out] == s_loop
rb = capi.ResidentBatch(b)
for _ in range(4):
    rb.run(p, sync=True)
t0 = time.perf_counter()
for _ in range(20):
    rb.run(p, sync=True)
t_res = (time.perf_counter() - t0) / 20
rb.close()
print(f"config 1 (1000 x 1 kb, BandEd score-only): per-pair loop {t_loop*1e3:.1f} ms ({1000/t_loop:,.0f} pairs/s), "
      f"one quicked_align_batch call {t_batch*1e3:.2f} ms ({1000/t_batch:,.0f} pairs/s), resident batch run {t_res*1e3:.3f} ms ({1000/t_res:,.0f} pairs/s)")

# The reference's parallel mode over the per-pair ABI (align_benchmark.c:246-284: N OpenMP threads, an aligner per thread and
# pair): T host threads, each with its own quicked_new / quicked_align / quicked_free per pair.  Every thread has its own
# context (streams, pools), so the calls of different threads are on the device at the same time.
import threading
for length, algo, only, label in ((1000, capi.BANDED, True, "1 kb BandEd score-only"), (10000, capi.QUICKED, False, "10 kb QuickEd + CIGAR")):
    pairs = list(datagen.generate(400, length, 0.05, seed=5).pairs())
    p = capi.make_params(algo=algo, only_score=only)
    want = [once(p, pt) for pt in pairs[:64]]
    for T in (1, 2, 4, 8, 16):
        got = [None] * T
        go = threading.Barrier(T + 1)

        def work(i):
            for pt in pairs[:8]:
                once(p, pt)                               # this thread's context: streams, pinned block
            go.wait()
            got[i] = [once(p, pt) for pt in pairs]

        ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        [t.start() for t in ths]
        go.wait()
        t0 = time.perf_counter()
        [t.join() for t in ths]
        dt = time.perf_counter() - t0
        assert all(g[:64] == want for g in got)
        print(f"{label}, {T:2d} host threads x {len(pairs)} new+align+free each: {T * len(pairs) / dt:9,.0f} calls/s ({dt / len(pairs) * 1e3:.3f} ms per call and thread)", flush=True)
    capi.pool_trim()
