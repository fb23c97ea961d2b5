"""probe: aggregate throughput of N host threads, each with its own resident batch (own stream + pool)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
STEPS = 6
batches = [datagen.generate(PAIRS, 10000, 0.05, seed=0x51CED, first=i * PAIRS) for i in range(NT)]
p = capi.make_params(algo=capi.BANDED, only_score=True)
rbs = [None] * NT
def setup(i):
    rbs[i] = capi.ResidentBatch(batches[i]); rbs[i].run(p, sync=True)
def work(i):
    for _ in range(STEPS): rbs[i].run(p, sync=False)
    rbs[i].sync()
ths = [threading.Thread(target=setup, args=(i,)) for i in range(NT)]
[t.start() for t in ths]; [t.join() for t in ths]
# NOTE: batch objects are bound to the creating thread's context; run them from new threads is fine (context is per calling thread)
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in ths]; [t.join() for t in ths]
dt = time.perf_counter() - t0
print(f"threads {NT} pairs/batch {PAIRS}: {NT*PAIRS*STEPS/dt:,.0f} alignments/s aggregate, {dt/STEPS*1e3:.2f} ms per round")
