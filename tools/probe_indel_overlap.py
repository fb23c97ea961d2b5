"""Do the host-driven QuickEd chains of DIFFERENT host threads overlap on the device?  T threads, each with its OWN batch of
`per` indel-heavy pairs (4 x 800-base indels per 10 kb pair: stages 2 / 3, band doubling, tall fills), synchronous runs.
The work grows with T; if the chains overlap the aggregate rate grows with it.  (tools/probe_indel_threads.py keeps the
total fixed: it shows that splitting a batch does not help, not whether chains overlap.)
    python tools/probe_indel_overlap.py [per=20000] [rounds=4] [Ts=1,2,3,4]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quicked_amd import capi, datagen

per = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Ts = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 3, 4]
p = capi.make_params(algo=capi.QUICKED)
shards = [datagen.generate(per, 10000, 0.05, seed=0x51CED, first=i * per, indels_num=4, indels_len=800) for i in range(max(Ts))]
os.environ["QE_QUICKED_FAST"] = "0"          # the host-driven flow, as bench.py times it on this data
for T in Ts:
    bar = threading.Barrier(T + 1)
    err = []

    def work(i):
        try:
            rb = capi.ResidentBatch(shards[i])
            for _ in range(2):
                rb.run(p, sync=True)
            bar.wait()
            for _ in range(rounds):
                if rb.run(p, sync=True) < 0:
                    raise RuntimeError("run failed")
            bar.wait()
            rb.close()
            capi.pool_trim()
        except Exception as e:       # noqa: BLE001
            err.append(e)
            try:
                bar.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for th in ths:
        th.start()
    try:
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = float("nan")
    for th in ths:
        th.join()
    print(f"T={T}: {T * per * rounds / dt / 1e6:.3f} M alignments/s, {dt / rounds * 1e3:.1f} ms per round of {T} x {per} pairs {err if err else ''}", flush=True)
