"""first difference between the GPU's CIGARs and the oracle's on a generated batch: python tools/debug_cigar.py COUNT LENGTH ERROR [SEED [ALGO]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from quicked_amd import capi, datagen
import oracle_lib as O
count, length, error = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
algo = int(sys.argv[5]) if len(sys.argv) > 5 else capi.QUICKED
b = datagen.generate(count, length, error, seed=seed)
al = capi.QuickedAligner(); al.setAlgorithm(algo)
pairs = list(b.pairs())
st, out = al.alignBatch(pairs)
bad = 0
for i, ((p, t), o) in enumerate(zip(pairs, out)):
    s2, sc, cg = O.oracle_align(p, t, algo=algo)
    g = o[2]
    if g != cg or o[1] != sc:
        bad += 1
        if bad <= 3:
            k = next((k for k in range(min(len(g), len(cg))) if g[k] != cg[k]), min(len(g), len(cg)))
            print(f"pair {i}: status {o[0]} / {s2}, score {o[1]} / {sc}, len {len(g)} / {len(cg)}, first difference at op {k} of {len(cg)}")
            print("  gpu   ", g[max(0, k - 30):k + 30]); print("  oracle", cg[max(0, k - 30):k + 30])
print(f"{bad} of {count} differ")
