"""One-off soak (not collected by pytest): long reads through every CIGAR path -- QuickEd (device-side stage 1 where no
pair may split, classic where one may; forced through stages 2 / 3 by HEW parameters), Hirschberg with real splits, BandEd,
WindowEd (9 / 1 and 5 / 2 windows: k_windowed_cp) -- against the oracle.
   python tests/soak_long.py FIRST_SEED COUNT"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle_lib as O
from quicked_amd import capi, datagen
from test_gpu_parity import _pools

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    t0 = time.time()
    rng = np.random.default_rng(seed)
    pairs = []
    for i in range(20):
        L = int(rng.choice([3000, 7000, 12000, 26000, 45000]))
        e = float(rng.choice([0.01, 0.05, 0.12]))
        kw = {}
        if rng.random() < 0.25:
            kw = dict(indels_num=int(rng.integers(1, 4)), indels_len=int(rng.choice([100, 400, 1500])))
        p, t = next(datagen.generate(1, L, e, seed=seed * 977 + i, **kw).pairs())
        pairs.append((p, t))
    batch = datagen.PairBatch(*_pools(pairs))
    rb = capi.ResidentBatch(batch)
    for algo, kwp in ((0, {}), (0, {}), (3, dict(bandwidth=20)), (2, dict(bandwidth=20)), (1, {}), (1, dict(window_size=5, overlap_size=2)),
                      (0, dict(hew_threshold=(10, 10), hew_percentage=(1, 1)))):
        prm = capi.make_params(algo=algo, **kwp)
        assert rb.run(prm, sync=False) >= 0
        assert rb.fetch() >= 0
        s, st = rb.scores(); cg = rb.cigars()
        for i, (p, t) in enumerate(pairs):
            est, esc, ecg = O.oracle_align(p, t, algo=algo, **kwp)
            dist_ok = algo in (0, 1) or O.oracle().qo_exact_distance(p, len(p), t, len(t)) <= max(len(p), len(t)) * 20 // 100
            if not dist_ok or est < 0:
                continue
            if (st[i], s[i], cg[i]) != (est, esc, ecg):
                bad += 1
                print("MISMATCH seed", seed, "algo", algo, "pair", i, len(p), len(t), (st[i], s[i]), (est, esc), flush=True)
    rb.close()
    print(f"seed {seed}: {time.time() - t0:.1f} s", flush=True)
print(f"soak_long: seeds {first}..{first + count - 1}, mismatches: {bad}")
