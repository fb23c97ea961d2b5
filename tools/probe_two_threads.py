"""T host threads, each streaming runs of its own resident batch (QuickEd + CIGAR or BandEd score-only): aggregate rate per T
-- what `one aligner per thread` (align_benchmark.c:246-284) costs or gains against one thread with a deep rotation"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
which = sys.argv[2] if len(sys.argv) > 2 else "quicked"
Ts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
p = capi.make_params(algo=capi.QUICKED) if which == "quicked" else capi.make_params(algo=capi.BANDED, only_score=True)
batch = datagen.generate(n, 10000, 0.05, seed=0x51CED)
for T in Ts:
    bar = threading.Barrier(T + 1)
    err = []

    def work():
        try:
            rb = capi.ResidentBatch(batch)
            rb.run(p, sync=True)
            for _ in range(6):
                rb.run(p, sync=False)
            rb.sync()
            bar.wait()
            for _ in range(steps):
                rb.run(p, sync=False)
            rb.sync()
            bar.wait()
            rb.close()
            capi.pool_trim()
        except Exception as e:      # noqa: BLE001
            err.append(e); bar.abort()

    ths = [threading.Thread(target=work) for _ in range(T)]
    for th in ths:
        th.start()
    try:
        bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = float("nan")
    for th in ths:
        th.join()
    print(f"{which} T {T} x {n} pairs x {steps} runs: {T * n * steps / dt / 1e6:6.3f} M alignments/s  ({dt / steps * 1e3:7.2f} ms per round of T runs) {err[:1]}", flush=True)
