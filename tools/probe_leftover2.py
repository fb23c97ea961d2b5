"""isolating what slows a later stream of small batches: (A) the main thread's own big allocations that came and went;
(B) short-lived threads with TINY batches (contexts and streams, no big memory); (C) short-lived threads with big batches"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen

small = datagen.generate(12500, 10000, 0.05, seed=0x51CED)
tiny = datagen.generate(64, 1000, 0.05, seed=0x51CED)
big = datagen.generate(100000, 10000, 0.05, seed=0x51CED)
p = capi.make_params(algo=capi.BANDED, only_score=True)


def share_rate(tag):
    rb = capi.ResidentBatch(small)
    for _ in range(3):
        rb.run(p, sync=True)
    for _ in range(24):
        rb.run(p, sync=False)
    rb.sync()
    t0 = time.perf_counter()
    for _ in range(160):
        rb.run(p, sync=False)
    rb.sync()
    dt = time.perf_counter() - t0
    st = capi.pool_stats()
    rb.close()
    print(f"{tag}: {12500 * 160 / dt / 1e6:.2f} M alignments/s ({dt / 160 * 1e3:.2f} ms per step), contexts {st.get('contexts')}", flush=True)


def threads(n, batch, runs=3):
    def work():
        rb = capi.ResidentBatch(batch)
        for _ in range(runs):
            rb.run(p, sync=False)
        rb.fetch()
        rb.close()
    for k in range(0, n, 4):
        ths = [threading.Thread(target=work) for _ in range(min(4, n - k))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()


which = sys.argv[1]
share_rate("fresh")
if which == "A":
    for _ in range(3):
        rbs = [capi.ResidentBatch(big) for _ in range(4)]
        for rb in rbs:
            rb.run(p, sync=True)
        for rb in rbs:
            rb.close()
    share_rate("A: after the main thread's own 12 big batch objects came and went")
elif which == "B":
    threads(8, tiny)
    share_rate("B: after 8 short threads with tiny batches")
    threads(8, tiny, runs=12)
    share_rate("B: after 8 more, 12 queued runs each (more streams)")
elif which == "C":
    threads(4, big)
    share_rate("C: after 4 short threads with big batches")
