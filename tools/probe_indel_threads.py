"""QuickEd + CIGAR on indel-heavy pairs (stages 2 / 3 are host-driven: every run blocks its host thread on the stage
results) with T host threads, each with its own batches -- the reference harness' own parallel mode (one aligner per
OpenMP thread, align_benchmark.c:246-284).  Aggregate alignments/s per T, total pairs per round fixed."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen

total = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
length = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
error = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
indels = int(sys.argv[5]) if len(sys.argv) > 5 else 4
which = sys.argv[7] if len(sys.argv) > 7 else "quicked"
p = {"quicked": lambda: capi.make_params(algo=capi.QUICKED),
     "windowed9": lambda: capi.make_params(algo=capi.WINDOWED, window_size=9, overlap_size=1, only_score=True),
     "banded40": lambda: capi.make_params(algo=capi.BANDED, bandwidth=40, only_score=True),
     "banded40c": lambda: capi.make_params(algo=capi.BANDED, bandwidth=40)}[which]()
ilen = 800 if indels else 0
ref = None
Ts = [int(x) for x in sys.argv[6].split(',')] if len(sys.argv) > 6 else [1, 2, 3, 4, 6, 8]
for T in Ts:
    per = total // T
    shards = [datagen.generate(per, length, error, seed=0x51CED, first=i * per, indels_num=indels, indels_len=ilen) for i in range(T)]
    sums = [0] * T
    bar = threading.Barrier(T + 1)
    err = []

    def work(i):
        try:
            rb = capi.ResidentBatch(shards[i])
            for _ in range(2):
                rb.run(p, sync=True)
            bar.wait()
            for _ in range(rounds):
                st = rb.run(p, sync=True)
                if st < 0:
                    raise RuntimeError("run failed")
            bar.wait()
            sc, _ = rb.scores()
            sums[i] = int(sum(int(x) for x in sc))
            rb.close()
            capi.pool_trim()
        except Exception as e:       # noqa: BLE001
            err.append(e)
            try:
                bar.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for th in ths:
        th.start()
    try:
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = float("nan")
    for th in ths:
        th.join()
    if err:
        print(f"T {T}: {err[0]!r}", flush=True)
        continue
    s = sum(sums)
    if ref is None and T * per == total:
        ref = s
    print(f"T {T:2d} x {per:6d} pairs: {dt / rounds * 1e3:8.1f} ms per round  {T * per * rounds / dt / 1e6:7.3f} M alignments/s  checksum {s}"
          f"{'' if T * per != total else ('  same' if s == ref else '  DIFFERENT')}", flush=True)
