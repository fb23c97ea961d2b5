#!/usr/bin/env python3
"""Config 4 (10 k pairs of 100 kb at 10 %, QuickEd + Hirschberg CIGAR) driven by T host threads, each with its OWN resident batch
object of the full 10 k pairs (the reference's parallel mode, align_benchmark.c:246-284: one aligner per thread): a splitting
run is host-driven level by level (bpm_hirschberg.c:63-65 needs its cutoffs on the host), so ONE thread leaves the chip to one
run's chain of stages; do the chains of several threads overlap?   python3 tools/probe_cfg4_threads.py [pairs] [steps]"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quicked_amd import capi, datagen

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
batch = datagen.generate(pairs, 100000, 0.10, seed=datagen.DEFAULT_SEED)
p = capi.make_params(algo=capi.QUICKED)
for T in (1, 2, 3):
    rbs = [capi.ResidentBatch(batch) for _ in range(T)]
    sums = [None] * T
    ready, go = threading.Barrier(T + 1), threading.Barrier(T + 1)

    def thread(i):
        # warm-up and timed runs in the SAME thread: a context (streams, pools) is its thread's, and one that loses its thread
        # is fair game for the next allocation that finds the device full
        for _ in range(2):
            assert rbs[i].run(p, sync=True) >= 0
        sums[i] = int(rbs[i].scores()[0].astype("int64").sum())
        assert rbs[i].run(p, sync=False) >= 0
        rbs[i].sync()
        for _ in range(capi.pool_stats()["sets"] + 1):      # a stream of queued runs rotates over more pool sets; every set allocates once
            assert rbs[i].run(p, sync=False) >= 0
        rbs[i].sync()
        ready.wait()
        go.wait()
        for _ in range(steps):
            assert rbs[i].run(p, sync=False) >= 0
        rbs[i].sync()

    ths = [threading.Thread(target=thread, args=(i,)) for i in range(T)]
    [t.start() for t in ths]
    ready.wait()
    t0 = time.perf_counter()
    go.wait()
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    st = capi.pool_stats()
    print(f"T={T}: {pairs * steps * T / dt:,.0f} alignments/s  ({dt / steps * 1e3:.1f} ms per round of {T} runs; checksums equal: {len(set(sums)) == 1}; "
          f"device pools hold {st['device_pool_bytes'] / 1e9:.0f} GB, reclaim events {st['reclaim_events']})", flush=True)
    for rb in rbs:
        rb.close()
    capi.pool_trim()
