"""One-off soak (not collected by pytest): the randomised-shape parity fuzz of test_gpu_parity.py over many
seeds, optionally with QE_COOP_G=1 so that every score-only BandEd pass goes through k_banded<false>'s
multi-slot walk.   python tests/soak_fuzz.py FIRST_SEED COUNT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_parity as T


class _MP:
    def setenv(self, k, v):
        os.environ[k] = v
        T.capi.reload_env()          # the library parses its switches once (qe_pool.h: SwitchTable)


first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
import time
for seed in range(first, first + count):
    t0 = time.time()
    try:
        T.test_randomised_shapes_and_params(seed, _MP())
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
    print(f"seed {seed}: {time.time() - t0:.1f} s", flush=True)
print(f"soak: seeds {first}..{first + count - 1}, mismatching seeds: {bad}")
