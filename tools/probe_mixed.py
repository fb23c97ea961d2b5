"""QuickEd + CIGAR on MIXED data: 100 k pairs of 10 kb of which a small share carries 4 x 800-base indels (those pairs
leave stage 1; the fast flow aligns them when the run is fetched).  A stream of queued runs, every run fetched: rate per
share of hard pairs and per number of fetching threads."""
import os, sys, time, threading, queue
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
shares = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.0, 0.01, 0.05]
fetchers = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 3]
steps = int(os.environ.get("STEPS", "12"))
p = capi.make_params(algo=capi.QUICKED)


for share in shares:
    hard = int(n * share)
    easy = datagen.generate(n - hard, 10000, 0.05, seed=0x51CED)
    batch = easy.concat(datagen.generate(hard, 10000, 0.05, seed=0x51CED, first=n, indels_num=4, indels_len=800)) if hard else easy
    for F in fetchers:
        slots = int(os.environ.get("SLOTS", max(4, F + 3)))
        rbs = [capi.ResidentBatch(batch) for _ in range(slots)]
        for rb in rbs:
            rb.run(p, sync=True)
        for rb in rbs:
            rb.run(p, sync=False)
        for rb in rbs:
            rb.fetch()
        q = queue.Queue()
        free = queue.Queue()
        for rb in rbs:
            free.put(rb)
        deferred = []

        def fetcher():
            while True:
                rb = q.get()
                if rb is None:
                    return
                rb.fetch()
                deferred.append(rb.deferred_pairs())
                free.put(rb)

        ths = [threading.Thread(target=fetcher) for _ in range(F)]
        for th in ths:
            th.start()
        t0 = time.perf_counter()
        for _ in range(steps):
            rb = free.get()
            rb.run(p, sync=False)
            q.put(rb)
        for _ in ths:
            q.put(None)
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        print(f"hard pairs {share * 100:4.1f} % ({hard}), {F} fetching thread(s), {slots} batch objects: {n * steps / dt / 1e6:6.3f} M alignments/s "
              f"({dt / steps * 1e3:7.2f} ms per batch; pairs aligned at fetch per run: {max(deferred) if deferred else 0})", flush=True)
        for rb in rbs:
            rb.close()
