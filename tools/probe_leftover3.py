import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
small = datagen.generate(12500, 10000, 0.05, seed=0x51CED)
p = capi.make_params(algo=capi.BANDED, only_score=True)

def share_rate(tag, reset_events=False, reuse=None):
    rb = reuse or capi.ResidentBatch(small)
    for _ in range(3):
        rb.run(p, sync=True)
    for _ in range(24):
        rb.run(p, sync=False)
    rb.sync()
    if reset_events:
        rb.kernel_time()
    t0 = time.perf_counter()
    for _ in range(160):
        rb.run(p, sync=False)
    rb.sync()
    dt = time.perf_counter() - t0
    if reset_events:
        rb.kernel_time()
    if reuse is None:
        rb.close()
    print(f"{tag}: {12500 * 160 / dt / 1e6:.2f} M alignments/s ({dt / 160 * 1e3:.2f} ms per step)", flush=True)

which = sys.argv[1]
if which == "twice":
    for k in range(4):
        share_rate(f"call {k}")
elif which == "events":
    for k in range(4):
        share_rate(f"call {k}, kernel events collected", reset_events=True)
elif which == "samebatch":
    rb = capi.ResidentBatch(small)
    for k in range(4):
        share_rate(f"call {k}, same batch object", reuse=rb)
