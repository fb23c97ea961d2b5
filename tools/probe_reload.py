"""what a reload costs while the device is busy: QE_TRACE stage timers of quicked_batch_reload_packed from a second thread
while the main thread keeps a stream of runs going"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quicked_amd import capi, datagen
wl = sys.argv[1] if len(sys.argv) > 1 else "quicked"
p = capi.make_params(algo=capi.QUICKED) if wl == "quicked" else capi.make_params(algo=capi.BANDED, only_score=True)
b = datagen.generate(100000, 10000, 0.05, seed=5)
pw, po = capi.wire_pack_pool(b.pattern_pool, b.pattern_off, b.pattern_len, capi.WIRE_2BIT)
tw, to = capi.wire_pack_pool(b.text_pool, b.text_off, b.text_len, capi.WIRE_2BIT)
(pwp, h1), (twp, h2) = capi.pinned_array(pw), capi.pinned_array(tw)
rb = capi.ResidentBatch.from_wire(b, capi.WIRE_2BIT, pwp, po, twp, to)
other = capi.ResidentBatch.from_wire(b, capi.WIRE_2BIT, pwp, po, twp, to)
for _ in range(3):
    rb.run(p, sync=True)
stop = [False]
def loader():
    time.sleep(0.3)
    os.environ["QE_TRACE"] = "1"
    for k in range(6):
        t0 = time.perf_counter()
        other.reload_wire(b, capi.WIRE_2BIT, pwp, po, twp, to)
        print(f"reload {k}: {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
    stop[0] = True
th = threading.Thread(target=loader); th.start()
n = 0
while not stop[0]:
    rb.run(p, sync=False); n += 1
rb.sync(); th.join()
print(f"{wl}: {n} runs queued meanwhile", file=sys.stderr)
