"""a process that ends while an early-finish job may still be running (a queued QuickEd run with deferred pairs, never
fetched, batch never closed): it must exit cleanly"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen
b = datagen.generate(20000, 10000, 0.05, seed=1).concat(datagen.generate(600, 10000, 0.05, seed=2, indels_num=4, indels_len=800))
rb = capi.ResidentBatch(b)
p = capi.make_params(algo=capi.QUICKED)
rb.run(p, sync=True)
rb.run(p, sync=False)
rb.run(p, sync=False)
print("exiting with a run in flight", flush=True)
os._exit(0) if len(sys.argv) > 1 else None
