"""QuickEd + CIGAR on indel-heavy pairs (stages 2 / 3, band doubling): pool planner behaviour run by run"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = capi.make_params(algo=capi.QUICKED)
b = datagen.generate(n, 10000, 0.05, seed=0x51CED, indels_num=4, indels_len=800)
rb = capi.ResidentBatch(b)
for k, sync in enumerate([True, True, True, False, False, False, False, True]):
    t0 = time.perf_counter()
    st = rb.run(p, sync=sync)
    if not sync and k == 6:
        rb.sync()
    dt = time.perf_counter() - t0
    ps = capi.pool_stats()
    c = rb.counters()
    print(f"run {k} sync {sync}: status {st} {dt * 1e3:9.1f} ms  sets {ps['sets']} sub {ps['sub_batches']} pools {ps['pool_bytes'] / 2**30:6.1f} GiB budget {ps['pool_budget'] / 2**30:6.1f} GiB "
          f"reclaims {ps['reclaim_events']} stage2 {int(c[6])} stage3 {int(c[7])} deferred {rb.deferred_pairs()}", file=sys.stderr, flush=True)
    if st < 0:
        break
