"""debug helper: k_banded_coop (QE_COOP_G=G) vs the oracle on seeded sets"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from quicked_amd import capi, datagen

def run(gen, bw, G):
    os.environ["QE_COOP_G"] = str(G)
    batch = datagen.generate(**gen)
    rb = capi.ResidentBatch(batch)
    p = capi.make_params(algo=capi.BANDED, only_score=True, bandwidth=bw)
    rb.run(p, sync=True)
    s, st = rb.scores(); cnt = rb.counters(); rb.close()
    exp = np.array([O.oracle_align(a, b, algo=2, only_score=True, bandwidth=bw)[1] for a, b in batch.pairs()])
    bad = np.nonzero(s != exp)[0]
    print(f"gen={gen['length']}x{gen['count']} bw={bw} G={G}: mismatches={len(bad)} fallback_tasks={cnt[6]}", bad[:8], s[bad[:4]], exp[bad[:4]])

for gen in (dict(count=256, length=10000, error=0.05, seed=0x51CED), dict(count=64, length=1000, error=0.05, seed=0x51CED),
            dict(count=40, length=3000, error=0.3, seed=305), dict(count=16, length=100000, error=0.1, seed=5)):
    for bw in (5, 15, 30):
        for G in (2, 4, 8, 16, 32):
            run(gen, bw, G)
