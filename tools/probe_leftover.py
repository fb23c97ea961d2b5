"""What do host threads that have ended leave behind for a later stream of small batches?  The rate of 160 queued BandEd runs of
a 12.5 k-pair batch (11 pool sets in rotation): fresh; after 8 short-lived threads that each ran a batch of their own; after
those and quicked_pool_trim; after big allocations have come and gone (10 batch objects of 10 GB)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen

small = datagen.generate(12500, 10000, 0.05, seed=0x51CED)
big = datagen.generate(100000, 10000, 0.05, seed=0x51CED)
p = capi.make_params(algo=capi.BANDED, only_score=True)
pq = capi.make_params(algo=capi.QUICKED)


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
    print(f"{tag}: {12500 * 160 / dt / 1e6:.2f} M alignments/s ({dt / 160 * 1e3:.2f} ms per step), sets {st['sets']}, contexts {st['contexts']} ({st['contexts_leased']} leased)", flush=True)


def short_threads(n, quick):
    def work():
        rb = capi.ResidentBatch(big)
        for _ in range(3):
            rb.run(pq if quick else p, sync=False)
        rb.fetch()
        rb.close()
    for k in range(0, n, 4):
        ths = [threading.Thread(target=work) for _ in range(min(4, n - k))]
        for th in ths:
            th.start()
        for th in ths:
            th.join()


share_rate("fresh")
short_threads(8, False)
share_rate("after 8 short threads (BandEd)")
capi.pool_trim()
share_rate("... and quicked_pool_trim")
short_threads(8, True)
share_rate("after 8 short threads (QuickEd + CIGAR)")
capi.pool_trim()
share_rate("... and quicked_pool_trim")
rbs = [capi.ResidentBatch(big) for _ in range(10)]
for rb in rbs:
    rb.run(pq, sync=True)
for rb in rbs:
    rb.close()
capi.pool_trim()
share_rate("after 10 batch objects of 100 k pairs came and went (+ trim)")
