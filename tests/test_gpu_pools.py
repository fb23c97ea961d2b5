"""Device-pool, planner and host-thread tests: the memory and concurrency side of the library (SURVEY 8 a21, 8b's
threading contract).  Collected AFTER tests/test_gpu_parity.py, so that a failure here can never hide a kernel-form parity
test; every test still compares the HIP path's results with the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from quicked_amd import capi, datagen
from test_gpu_parity import _pools

pytestmark = pytest.mark.gpu


def device_free_fraction():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free_b, total_b = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
    return free_b.value / total_b.value


def test_concurrent_host_threads_one_aligner_each():
    """the reference's threading contract: no locks, no globals, one aligner per thread
    (align_benchmark.c:246-284).  Here every host thread gets its own streams and device pools; results must not
    depend on what the other threads are doing."""
    import ctypes as C
    import threading
    lib = capi.lib()
    batch = datagen.generate(count=48, length=1500, error=0.07, seed=515)
    pairs = list(batch.pairs())
    expect = {}
    for algo, only in ((capi.QUICKED, False), (capi.BANDED, True), (capi.HIRSCHBERG, False), (capi.WINDOWED, False)):
        expect[(algo, only)] = [O.oracle_align(p, t, algo=algo, only_score=only) for p, t in pairs]
    errors = []

    def worker(algo, only, lo, hi):
        try:
            prm = capi.make_params(algo=algo, only_score=only)
            a = capi.Aligner()
            assert lib.quicked_new(C.byref(a), C.byref(prm)) == capi.QUICKED_WIP
            for i in range(lo, hi):
                p, t = pairs[i]
                st = lib.quicked_align(C.byref(a), p, len(p), t, len(t))
                est, esc, ecg = expect[(algo, only)][i]
                got = (st, a.score, a.cigar.decode() if (a.cigar and not only) else None)
                if got != (est, esc, None if only else ecg):
                    errors.append((algo, only, i, got[:2]))
            lib.quicked_free(C.byref(a))
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = []
    for k, (algo, only) in enumerate(expect):
        for half in range(2):
            threads.append(threading.Thread(target=worker, args=(algo, only, 24 * half, 24 * half + 24)))
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


def test_streaming_reload_from_an_uploader_thread():
    """bench.py's end-to-end pattern: an uploader thread reloads batch objects while the main thread runs and fetches
    the others (ASCII and 2-bit input)"""
    import threading
    sets = [datagen.generate(count=256, length=900, error=0.05, seed=800 + k) for k in range(6)]
    expect = [[O.oracle_align(p, t, algo=2, only_score=True)[1] for p, t in b.pairs()] for b in sets]
    prm = capi.make_params(algo=capi.BANDED, only_score=True)
    for wire in (None, capi.WIRE_2BIT):
        slots = 3
        words = None
        if wire is None:
            rbs = [capi.ResidentBatch(sets[0]) for _ in range(slots)]
        else:
            words = [capi.wire_pack_pool(b.pattern_pool, b.pattern_off, b.pattern_len, wire) +
                     capi.wire_pack_pool(b.text_pool, b.text_off, b.text_len, wire) for b in sets]
            rbs = [capi.ResidentBatch.from_wire(sets[0], wire, *words[0]) for _ in range(slots)]
        up = [threading.Event() for _ in sets]
        done = [threading.Event() for _ in sets]
        errs = []

        def uploader():
            try:
                for k, b in enumerate(sets):
                    if k >= slots:
                        done[k - slots].wait()
                    st = rbs[k % slots].reload(b) if wire is None else rbs[k % slots].reload_wire(b, wire, *words[k])
                    assert st >= 0
                    up[k].set()
            except Exception as e:      # noqa: BLE001
                errs.append(e)
                for ev in up:
                    ev.set()

        th = threading.Thread(target=uploader)
        th.start()
        got = {}

        def finish(k):
            assert rbs[k % slots].fetch() >= 0
            got[k] = rbs[k % slots].scores()[0].tolist()
            done[k].set()

        for k in range(len(sets)):
            up[k].wait()
            assert not errs, errs
            assert rbs[k % slots].run(prm, sync=False) >= 0
            if k >= 1:
                finish(k - 1)
        finish(len(sets) - 1)
        th.join()
        for k in range(len(sets)):
            assert got[k] == expect[k], (wire, k)
        for rb in rbs:
            rb.close()


def test_planner_cuts_a_large_cigar_batch_before_it_runs_out_of_memory():
    """a21: the device-pool planner.  400 k pairs of 10 kb through QuickEd + CIGAR need ~92 GB of fill checkpoints per
    run; three pool sets of that do not fit 288 GB.  The planner must pick the rotation depth and the fill sub-batches
    up front: no out-of-memory reclaim event, results identical to the oracle's, and the rate stays in the millions."""
    import time
    n = 400000
    batch = datagen.generate(count=n, length=10000, error=0.05, seed=0x51CED)
    before = capi.pool_stats()["reclaim_events"]
    rb = capi.ResidentBatch(batch)
    p = capi.make_params(algo=capi.QUICKED)
    assert rb.run(p, sync=True) >= 0                      # sizes the pools
    first = capi.pool_stats()
    for _ in range(3):                                    # every set of the rotation allocates its pools once (tens of GB each)
        assert rb.run(p, sync=False) >= 0
    rb.sync()
    t0 = time.perf_counter()
    steps = 4
    for _ in range(steps):
        assert rb.run(p, sync=False) >= 0
    rb.sync()
    dt = time.perf_counter() - t0
    assert rb.run(p, sync=True) >= 0
    s, st = rb.scores()
    cg = rb.cigars()
    stats = capi.pool_stats()
    rb.close()
    rate = n * steps / dt
    print(f"planner: {rate / 1e6:.2f} M pairs/s, sets {stats['sets']}, fill sub-batches {stats['sub_batches']}, "
          f"pools {stats['pool_bytes'] / 2**30:.1f} GiB, first run {first}")
    assert stats["reclaim_events"] == before, stats
    assert (st == capi.QUICKED_WIP).all()
    for i in list(range(0, 48)) + list(range(n - 16, n)):
        pt, tt = batch.pattern(i), batch.text(i)
        assert (st[i], s[i], cg[i]) == O.oracle_align(pt, tt, algo=0), i
    assert rate > 3.0e6, rate


def test_two_host_threads_plan_hbm_together():
    """the process-wide HBM ledger: two host threads, each with 150 k pairs of 10 kb through QuickEd + CIGAR (~35 GB of fill
    checkpoints per run and pool set), plan their device pools against what the OTHER has planned, not against the whole
    device each: no out-of-memory reclaim event, both threads' results equal to the oracle's on a stride.  A thread that
    is done gives its pools back (quicked_pool_trim)."""
    import threading
    n = 150000
    batch = datagen.generate(count=n, length=10000, error=0.05, seed=0x51CED)
    assert capi.pool_trim() == 0            # what this (idle) thread's pools hold from earlier tests goes back to the device
    before = capi.pool_stats()["reclaim_events"]
    errors, stats, rates = [], {}, {}
    gate = threading.Barrier(2)

    def worker(name):
        try:
            import time
            rb = capi.ResidentBatch(batch)
            p = capi.make_params(algo=capi.QUICKED)
            gate.wait()
            assert rb.run(p, sync=True) >= 0
            for _ in range(3):                        # every set of the rotation allocates its pools once (tens of GB each, seconds)
                assert rb.run(p, sync=False) >= 0
            rb.sync()
            gate.wait()
            t0 = time.perf_counter()
            for _ in range(4):
                assert rb.run(p, sync=False) >= 0
            rb.sync()
            rates[name] = n * 4 / (time.perf_counter() - t0)
            assert rb.run(p, sync=True) >= 0
            s, st = rb.scores()
            cg = rb.cigars()
            stats[name] = capi.pool_stats()
            rb.close()
            assert capi.pool_trim() == 0          # this thread is done: its pools go back to the device
            assert (st == capi.QUICKED_WIP).all()
            for i in list(range(0, n, n // 24)) + [n - 1]:
                assert (st[i], s[i], cg[i]) == O.oracle_align(batch.pattern(i), batch.text(i), algo=0), (name, i)
        except Exception as e:      # noqa: BLE001
            errors.append((name, repr(e)))
            try:
                gate.abort()
            except Exception:      # noqa: BLE001
                pass

    ths = [threading.Thread(target=worker, args=(k,)) for k in ("a", "b")]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    print(f"two threads: {rates}, {stats}")
    for k in ("a", "b"):
        assert stats[k]["reclaim_events"] == before, stats
        # neither thread was allowed to plan for the whole device: the budgets of their pool sets add up to less than it
    total = 288 * 2**30
    assert sum(stats[k]["pool_budget"] * stats[k]["sets"] for k in ("a", "b")) < total, stats
    assert sum(stats[k]["pool_bytes"] for k in ("a", "b")) < total, stats
    assert all(stats[k]["sets"] >= 2 and stats[k]["sub_batches"] <= 2 for k in ("a", "b")), stats      # nobody was starved
    assert min(rates.values()) > 0.4e6, rates           # a ledger test, not a benchmark: four runs each, two threads on one chip
    # the threads trimmed their pools before they ended: the device is free again
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free_b, total_b = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
    assert free_b.value > 0.8 * total_b.value, (free_b.value, total_b.value)


@pytest.mark.parametrize("finishers", ["3", "0"])
def test_mixed_batches_finish_early(finishers, monkeypatch):
    """ordinary reads with a few large-indel pairs among them: the QuickEd fast flow sizes its align step for the ordinary
    ones (estimate within twice the median bound), the others leave it and are aligned through the host-driven stages -- by
    the library's early-finish threads as soon as the run is over (QE_FINISHERS = 3, default) or by the caller's fetch
    (0).  A stream of queued runs over several batch objects, fetched by one thread: every result equal to the oracle's,
    the deferred pairs counted, no pool grown to the outliers' size."""
    monkeypatch.setenv("QE_FINISHERS", finishers)
    import numpy as np
    easy = datagen.generate(count=6000, length=4000, error=0.05, seed=555)
    hard = datagen.generate(count=120, length=4000, error=0.05, seed=556, indels_num=3, indels_len=400)
    pairs = list(easy.pairs()) + list(hard.pairs())
    batch = datagen.PairBatch(*_pools(pairs))
    want = {i: O.oracle_align(*pairs[i], algo=0) for i in list(range(0, 6000, 500)) + list(range(6000, 6120, 7))}
    prm = capi.make_params(algo=capi.QUICKED)
    rbs = [capi.ResidentBatch(batch) for _ in range(3)]
    for rb in rbs:
        assert rb.run(prm, sync=True) >= 0
    deferred = []
    for rnd in range(4):
        for rb in rbs:
            assert rb.run(prm, sync=False) >= 0
        for rb in rbs:
            assert rb.fetch() >= 0
            deferred.append(rb.deferred_pairs())
            s, st = rb.scores(); cg = rb.cigars()
            for i, w in want.items():
                assert (st[i], s[i], cg[i]) == w, (finishers, rnd, i)
    assert min(deferred) > 0, deferred                    # the outliers did leave the fast flow ...
    assert max(deferred) < 600, deferred                  # ... and only they (and the few ordinary pairs above the estimate)
    stats = capi.pool_stats()
    for rb in rbs:
        rb.close()
    assert stats["pool_bytes"] < 40 * 2**30, stats        # this thread's pools: sized for the ordinary pairs


def test_threads_that_end_without_trimming_leave_no_pools_behind():
    """A host thread that just ends (no quicked_pool_trim, batches closed) must neither crash in its thread-local
    destructors -- HIP may not be called from there -- nor strand its device pools: its context's lease ends, and the next
    thread that needs a context takes it over, streams and pools and all.  Thirty short-lived threads with pools of a few
    GB each therefore never hold more than three threads' worth of pools, every one of their fetches succeeds, the number
    of contexts stays that of the threads alive at once, and a live thread's quicked_pool_trim gives everything back."""
    import threading
    import time
    assert capi.pool_trim() == 0
    batch = datagen.generate(count=20000, length=4000, error=0.05, seed=321)
    want = [O.oracle_align(p, t, algo=0) for p, t in list(batch.pairs())[:8]]
    errors = []
    before = capi.pool_stats()

    def worker():
        try:
            rb = capi.ResidentBatch(batch)
            prm = capi.make_params(algo=capi.QUICKED)
            for _ in range(2):
                assert rb.run(prm, sync=False) >= 0
            assert rb.fetch() >= 0
            s, st = rb.scores(); cg = rb.cigars()
            for i in range(8):
                assert (st[i], s[i], cg[i]) == want[i]
            rb.close()                            # ... and the thread ends with its pools allocated
        except Exception as e:                    # noqa: BLE001
            errors.append(e)

    peak = 0
    for _ in range(10):
        ths = [threading.Thread(target=worker) for _ in range(3)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        peak = max(peak, capi.pool_stats()["device_pool_bytes"])
    assert not errors, errors
    after = capi.pool_stats()
    # contexts are reused, not multiplied: at most the three workers' on top of what existed (CPython's join() may return a
    # moment before the OS thread has run its thread-local destructors, so a round can find a lease not yet ended)
    assert after["contexts"] <= before["contexts"] + 6, (before, after)
    assert after["reclaim_events"] == before["reclaim_events"], (before, after)
    assert peak < 150 * 2**30, peak
    for _ in range(50):                           # see above: the last three leases end within microseconds of join()
        assert capi.pool_trim() == 0              # a live thread: releases what the ended ones held
        if device_free_fraction() > 0.8:
            break
        time.sleep(0.02)
    assert device_free_fraction() > 0.8


def test_an_idle_threads_pools_are_taken_when_another_thread_needs_them():
    """Out of memory is a path, not an error (qe_pool.h).  Thread A runs a batch, keeps its pools (tens of GB) and sits
    idle -- alive, no call in progress, no trim.  The rest of the device is then taken by a ballast allocation, so that
    thread B cannot even create its batch object next to A's pools: B's allocation goes through the reclaim levels and takes
    A's pools (A's streams are drained first).  B's results are right, A's pools are gone, and A -- at work again after the
    ballast is freed -- allocates anew and is right too."""
    import ctypes as C
    import threading
    assert capi.pool_trim() == 0
    n = 100000
    batch = datagen.generate(count=n, length=10000, error=0.05, seed=0xA11CE)
    idx = list(range(0, n, n // 16)) + [n - 1]
    want = {i: O.oracle_align(batch.pattern(i), batch.text(i), algo=0) for i in idx}
    errors, seen = [], {}
    a_ran, b_done = threading.Event(), threading.Event()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]

    def check(rb):
        s, st = rb.scores(); cg = rb.cigars()
        for i in idx:
            assert (st[i], s[i], cg[i]) == want[i], i

    def thread_a():
        try:
            rb = capi.ResidentBatch(batch)
            p = capi.make_params(algo=capi.QUICKED)
            assert rb.run(p, sync=True) >= 0
            for _ in range(5):                     # the whole rotation allocates
                assert rb.run(p, sync=False) >= 0
            assert rb.fetch() >= 0
            check(rb)
            seen["a_before"] = capi.pool_stats()
            a_ran.set()
            assert b_done.wait(600)                # idle, pools allocated, no call in progress
            seen["a_after"] = capi.pool_stats()
            assert rb.run(p, sync=True) >= 0       # ... and at work again
            check(rb)
            rb.close()
            assert capi.pool_trim() == 0
        except Exception as e:                     # noqa: BLE001
            errors.append(("a", repr(e))); a_ran.set()

    def thread_b():
        ballast = C.c_void_p()
        try:
            assert a_ran.wait(600)
            free_b, total_b = C.c_size_t(), C.c_size_t()
            assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
            assert hip.hipMalloc(C.byref(ballast), free_b.value - (3 << 30)) == 0      # 3 GB left: less than B's batch object
            seen["b_before"] = capi.pool_stats()
            rb = capi.ResidentBatch(batch)
            p = capi.make_params(algo=capi.QUICKED)
            assert rb.run(p, sync=True) >= 0
            for _ in range(3):
                assert rb.run(p, sync=False) >= 0
            assert rb.fetch() >= 0
            check(rb)
            seen["b_after"] = capi.pool_stats()
            rb.close()
            assert capi.pool_trim() == 0
        except Exception as e:                     # noqa: BLE001
            errors.append(("b", repr(e)))
        finally:
            if ballast.value:
                hip.hipFree(ballast)
            b_done.set()

    ths = [threading.Thread(target=thread_a), threading.Thread(target=thread_b)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, (errors, seen)
    print("idle-thread test:", seen)
    assert seen["a_before"]["pool_bytes"] > 20 * 2**30, seen
    assert seen["a_after"]["pool_bytes"] < seen["a_before"]["pool_bytes"] // 2, seen          # B took them
    assert seen["b_after"]["reclaim_events"] > seen["b_before"]["reclaim_events"], seen       # ... through the out-of-memory path
    assert capi.pool_trim() == 0
    assert device_free_fraction() > 0.8


def test_queued_runs_of_different_kinds_on_one_batch_keep_their_order():
    """The batch has one result arena for queued runs.  A QuickEd + CIGAR run (a long chain of kernels) followed at once by
    a score-only BandEd run (one kernel, another stream of the rotation) on the SAME batch, no fetch in between: the fetch
    must bring the second run's results, complete and unmixed, whichever run finishes first on the device."""
    batch = datagen.generate(count=30000, length=3000, error=0.06, seed=777)
    pairs = list(batch.pairs())
    pq = capi.make_params(algo=capi.QUICKED)
    pb = capi.make_params(algo=capi.BANDED, only_score=True)
    rb = capi.ResidentBatch(batch)
    assert rb.run(pq, sync=True) >= 0                      # sizes the pools; the next QuickEd run takes the queued (fast) flow
    idx = list(range(0, len(pairs), 997)) + [len(pairs) - 1]
    want_b = {i: O.oracle_align(*pairs[i], algo=2, only_score=True) for i in idx}
    want_q = {i: O.oracle_align(*pairs[i], algo=0) for i in idx}
    for rnd in range(6):
        assert rb.run(pq, sync=False) >= 0
        assert rb.run(pb, sync=False) >= 0
        assert rb.fetch() >= 0
        s, st = rb.scores()
        for i in idx:
            assert (st[i], s[i]) == want_b[i][:2], (rnd, i)
        assert rb.run(pb, sync=False) >= 0
        assert rb.run(pq, sync=False) >= 0
        assert rb.fetch() >= 0
        s, st = rb.scores(); cg = rb.cigars()
        for i in idx:
            assert (st[i], s[i], cg[i]) == want_q[i], (rnd, i)
    rb.close()


def test_getters_are_stable_while_a_queued_run_finishes_early(monkeypatch):
    """quicked_batch.h: a sync == 0 run leaves the getters' data -- and the zero-copy CIGAR view -- untouched until the
    caller fetches.  With pairs that leave the fast flow, an early-finish thread brings the queued run's results to the
    host as soon as the run is over; it must do so into the batch's shadow set, not into what the caller is reading."""
    import time
    monkeypatch.setenv("QE_FINISHERS", "3")
    easy = datagen.generate(count=6000, length=4000, error=0.05, seed=555)
    hard = datagen.generate(count=120, length=4000, error=0.05, seed=556, indels_num=3, indels_len=400)
    pairs = list(easy.pairs()) + list(hard.pairs())
    batch = datagen.PairBatch(*_pools(pairs))
    prm = capi.make_params(algo=capi.QUICKED)
    rb = capi.ResidentBatch(batch)
    assert rb.run(prm, sync=True) >= 0
    s0, st0 = rb.scores(); c0 = rb.cigars()
    pv, ov = rb.cigar_view()
    snap_pool, snap_off = pv.copy(), ov.copy()
    for rnd in range(3):
        assert rb.run(prm, sync=False) >= 0                # deferred pairs: a finisher works on this batch object now
        t_end = time.time() + 0.5
        while time.time() < t_end:                         # ... while the caller keeps reading the previous results
            s, st = rb.scores()
            assert (s == s0).all() and (st == st0).all()
            pv2, ov2 = rb.cigar_view()
            assert (ov2 == snap_off).all() and pv2.shape == snap_pool.shape and (pv2 == snap_pool).all()
        assert rb.fetch() >= 0
        assert rb.deferred_pairs() > 0
        s, st = rb.scores()
        assert (s == s0).all() and (st == st0).all() and rb.cigars() == c0
        pv, ov = rb.cigar_view()
        snap_pool, snap_off = pv.copy(), ov.copy()
    rb.close()


def test_merged_early_finish_gives_every_batch_its_own(monkeypatch):
    """The pairs that the queued QuickEd runs of SEVERAL batch objects left for the host-driven stages are aligned by one flow
    when those runs are over by the time an early-finish thread gets to work (qe_driver.hip merged_finish: a stand-in batch
    object with gathered bit-planes).  Every batch must get exactly what its own fetch would have produced: scores,
    statuses, CIGARs, the count of deferred pairs and every work counter -- compared here with QE_FINISHERS = 0, where the
    caller's fetch runs the flow per batch.  Batches of different sizes and lengths, one of them from packed wire words."""
    import time
    rng = np.random.default_rng(31)
    batches = []
    for k, (n_easy, n_hard, length) in enumerate(((3000, 90, 4000), (1500, 200, 2500), (2500, 40, 6000))):
        easy = datagen.generate(count=n_easy, length=length, error=0.05, seed=700 + k)
        hard = datagen.generate(count=n_hard, length=length, error=0.05, seed=800 + k, indels_num=3, indels_len=300)
        pairs = list(easy.pairs()) + list(hard.pairs())
        order = rng.permutation(len(pairs))
        batches.append(datagen.PairBatch(*_pools([pairs[i] for i in order])))
    prm = capi.make_params(algo=capi.QUICKED)

    def collect(finishers, wire):
        monkeypatch.setenv("QE_FINISHERS", finishers)
        rbs = [capi.ResidentBatch(b, wire=(wire if k == 2 else None)) for k, b in enumerate(batches)]
        for rb in rbs:
            assert rb.run(prm, sync=True) >= 0
        out = []
        for rnd in range(3):
            for rb in rbs:
                assert rb.run(prm, sync=False) >= 0
            for rb in rbs:
                rb.sync()
            if finishers != "0":
                time.sleep(0.6)                     # the early-finish thread takes the first job alone, the other two together
            got = []
            for rb in rbs:
                assert rb.fetch() >= 0
                s, st = rb.scores()
                got.append((s.copy(), st.copy(), rb.cigars(), rb.counters()[[0, 1, 2, 3, 4, 6, 7]].copy(), rb.deferred_pairs()))
            out.append(got)
        for rb in rbs:
            rb.close()
        return out

    before = capi.early_finish_stats()
    own = collect("0", capi.WIRE_2BIT)
    assert capi.early_finish_stats() == before
    merged = collect("1", capi.WIRE_2BIT)
    after = capi.early_finish_stats()
    assert after["merged_flows"] > before["merged_flows"] and after["merged_batches"] >= before["merged_batches"] + 2, (before, after)
    for rnd in range(3):
        for k in range(3):
            a, b = own[rnd][k], merged[rnd][k]
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), (rnd, k)
            assert a[2] == b[2], (rnd, k)
            assert (a[3] == b[3]).all(), (rnd, k, a[3], b[3])
            assert a[4] == b[4] and a[4] > 0, (rnd, k, a[4], b[4])
    # ... and the oracle agrees on a stride
    for k, bt in enumerate(batches):
        for i in range(0, len(bt), 211):
            assert (merged[2][k][1][i], merged[2][k][0][i], merged[2][k][2][i]) == O.oracle_align(bt.pattern(i), bt.text(i), algo=0), (k, i)
