#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic read pairs whose ASCII bytes are already resident in
HBM: pack -> kernels -> results in HBM.  The headline (`value`) is configs[1] of BASELINE.json: BandEd score-only,
100 k pairs of 10 kb at 5 % error, reference default bandwidth 15 %.  The same JSON line also carries

  workloads.quicked  configs[2]: QuickEd bound-and-align + CIGAR on the same pairs, with its own roofline / e2e /
                     cpu_baseline objects (this is the HBM-relevant workload of the path)
  workloads.quicked_score  (N = 1) the same pairs with only_score: bound, then one score-only pass over the fill's cells
  strong_share       (N = 1) both workloads at 12 500 pairs per step: the per-GPU share of BASELINE.json's "100 k pairs
                     at 8 GPUs", i.e. the rate one GPU of the 8-GPU strong-scaling target sees
  strong             (N > 1) both workloads with `--pairs` pairs IN TOTAL split over the ranks
  roofline           dominant kernel: algorithmic bytes / its launch duration ALONE on the chip (HIP events, a few
                     synchronous steps); `kernel_ms_overlapped` is the same launch's duration inside the timed region,
                     where up to `sets` runs share the chip; `aggregate_*` divide by the step time
  valu               the same kernel against the chip's VALU issue rate (its real bound), step-time based
  e2e                PCIe-inclusive rates: host buffers -> HBM -> run -> results on the host, batch after batch;
                     ASCII over the link, 2-bit words the caller already holds, and ASCII packed to 2 bits by the
                     library's SIMD host packer inside the clock (`ascii_hostpacked`)
  cpu_baseline       the compiled reference (or the oracle port) on the host cores, N = 1 only
  ranks_seen         all-reduce SUM of 1 over the ranks

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under a launcher (WORLD_SIZE set: the driver's `python -m torch.distributed.run ...
bench.py --gpus N`) this process is one rank; without one, `--gpus N` starts the N ranks itself as a child
torch.distributed.run BEFORE anything touches the GPU and relays its JSON line.  Every rank owns its own shard of the
seeded dataset (quicked_amd/shard.py), there is no data-path collective, and RCCL only reduces {pairs, cells, checksum}
(SUM) and the elapsed time (MAX) at the end.
"""
import argparse
import glob
import json
import os
import re
import sys
import threading
import time

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one
# serialise; a thread's runs rotate over up to 12 stream sets.  The runtime reads the variable when it initialises, so it is
# set here, before anything imports torch (a value the user has set is left alone).  20, not more: once a process has
# created 24 hardware queues -- they stay for its lifetime, whatever happens to the streams -- an 11-deep stream of
# 12.5 k-pair batches runs at 4.3 instead of 6.4 M alignments/s (tools/probe_leftover2.py: 14, 16 and 20 are immune).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from quicked_amd import shard  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue: a SIMD issues one wave64 VALU instruction per 2 cycles at best (MI355X_MICROARCH.md, "Wave scheduling":
# SIMD-32, 2 passes per wave64 op); 1024 SIMDs at the 2.4 GHz peak clock.  The block step of k_banded<false> is
# INSTR_PER_BLOCK_COLUMN VALU instructions per 64-row block per text column (ISA count of the 4-slot loop),
# some of which (v_lshl_add_u64, v_bfe, v_alignbit) hold the SIMD for 4 cycles: ISSUE_CYCLES_PER_BLOCK_COLUMN is the sum
# over the loop at the guide's rates (2 / 4 cycles), the bound nothing can beat without removing instructions.
SIMDS, PEAK_CLOCK_HZ = 1024, 2.4e9
INSTR_PER_BLOCK_COLUMN = 26.1
INSTR_PER_BLOCK_COLUMN_COOP = 33.4       # k_banded_coop_lds<false>, per LIVE block-column (profiles/r03_g_cfg4_sq_counters.txt)
ISSUE_CYCLES_PER_BLOCK_COLUMN = 61.0     # (2 766 x 2 + 570 x 4) / 128 block-columns of the unrolled 4-slot loop
MEASURED_LOOP_BLOCK_COLUMNS_PER_S = 1.70e12  # run64_skew<4> on registers at two waves per SIMD: 91 cycles per block-column (profiles/r06_b_skew_asm_bench.txt)
# reference anchors of BASELINE.md section 2 (one core of the survey container's 2.1 GHz Xeon, AVX2 build)
CPU_ANCHOR_PER_CORE = {"banded_score": 2463.0, "quicked": 1680.0}
FILL_BYTES_PER_BLOCK_COLUMN = 1.25       # 16 B of {Pv, Mv} per 16 columns + 16 B of carry words per 64 (qe_types.h: QE_CP_COLS)
STRONG_SHARE_PAIRS = 12500       # 100 k pairs over 8 GPUs (BASELINE.json north_star)
CFG5_MAX_PER_GPU = 250000        # configs[4] is 1 M pairs over 8 GPUs = 125 k per GPU; a rank never takes more than this


def visible_gpus():
    """GPUs of this node WITHOUT touching HIP (the launcher parent must not initialise the GPU): KFD topology nodes that
    have SIMDs, else DRM render nodes; capped by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set"""
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                m = re.search(r"^simd_count\s+(\d+)", f.read(), re.M)
            if m and int(m.group(1)) > 0:
                n += 1
        except OSError:
            continue
    if n == 0:
        n = len(glob.glob("/dev/dri/renderD*"))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def measured_copy_bandwidth(nbytes=1 << 30, reps=8):
    """device-to-device copy rate on this box (print the measured bandwidth next to the 8 TB/s spec):
    read + write bytes of hipMemcpyDtoD per second, GB/s, through the HIP runtime the library already loaded"""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        a, b, e0, e1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        if hip.hipMalloc(C.byref(a), C.c_size_t(nbytes)) or hip.hipMalloc(C.byref(b), C.c_size_t(nbytes)):
            return None
        hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
        hip.hipMemset(a, 1, C.c_size_t(nbytes))
        hip.hipMemcpyDtoD(b, a, C.c_size_t(nbytes))
        hip.hipDeviceSynchronize()
        hip.hipEventRecord(e0, None)
        for _ in range(reps):
            hip.hipMemcpyDtoDAsync(b, a, C.c_size_t(nbytes), None)
        hip.hipEventRecord(e1, None)
        hip.hipEventSynchronize(e1)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        hip.hipFree(a); hip.hipFree(b); hip.hipEventDestroy(e0); hip.hipEventDestroy(e1)
        return 2.0 * nbytes * reps / (ms.value * 1e-3) / 1e9 if ms.value > 0 else None
    except Exception:      # noqa: BLE001 -- a report field, not part of the path
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """threads this process can really run at once: the scheduler affinity mask, capped by the cgroup CPU quota
    (cpu.max = "<quota> <period>"), not os.cpu_count() (which counts the machine's CPUs, not ours)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        quota = q / float(g.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def strided_sample(batch, m):
    """`m` pairs of the batch taken at equal strides (same pools, other offset arrays): a sample that keeps the mix of a
    batch whose hard pairs are not at the front"""
    from quicked_amd.datagen import PairBatch
    idx = np.unique(np.linspace(0, len(batch) - 1, min(m, len(batch))).astype(np.int64))
    return PairBatch(batch.pattern_pool, np.ascontiguousarray(batch.pattern_off[idx]), np.ascontiguousarray(batch.pattern_len[idx]),
                     batch.text_pool, np.ascontiguousarray(batch.text_off[idx]), np.ascontiguousarray(batch.text_len[idx])), idx


def cpu_baseline(batch, params_kw, workload, budget_s=10.0, anchored=True):
    """The compiled reference (oracle/_ref, kind "reference") or the oracle restatement (kind "port") on the host cores:
    oracle/cpu_bench.c, one aligner per OpenMP thread over disjoint pair ranges (the reference's own model,
    align_benchmark.c:246-284), on a bounded prefix of the same workload.  Threads = what this process may really use."""
    import ctypes as C
    import subprocess
    import oracle_lib as O
    so = os.path.join(O.ORACLE_DIR, "libcpubench.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", O.ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.cpu_bench_run.restype = C.c_double
    lib.cpu_bench_run.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
    kind = "reference" if O.have_ref() else "port"
    ref_so = O.REF_SO.encode() if kind == "reference" else None
    cores, quota = usable_cpus()

    def run(n, threads):
        scores = np.zeros(n, dtype=np.int32)
        wall = lib.cpu_bench_run(ref_so, n, batch.pattern_pool.ctypes.data, batch.pattern_off.ctypes.data,
                                 batch.pattern_len.ctypes.data, batch.text_pool.ctypes.data, batch.text_off.ctypes.data,
                                 batch.text_len.ctypes.data, params_kw["algo"], 1 if params_kw.get("only_score") else 0,
                                 params_kw.get("bandwidth", 15), threads, scores.ctypes.data)
        assert wall > 0, "cpu_bench_run failed"
        return wall, scores

    run(min(len(batch), cores), cores)                   # warm the library, the arenas and the page cache
    # single-thread calibration, best of three, sized to ~0.3 s per try (64 pairs of 10 kb; 4 of 100 kb)
    per0 = run(min(len(batch), 2), 1)[0] / min(len(batch), 2)
    n1 = int(min(len(batch), max(2, min(64, 0.3 / max(per0, 1e-9)))))
    w1 = min(run(n1, 1)[0] for _ in range(3))
    per = w1 / n1
    n = int(min(len(batch), max(cores * 8, budget_s / per * cores)))
    wall, scores = run(n, cores)
    single = 1.0 / per
    per_thread = n / wall / cores
    anchor = CPU_ANCHOR_PER_CORE.get(workload) if anchored else None      # the anchors are for 10 kb / 5 % pairs
    out = {"value": n / wall, "unit": "alignments/s", "cores": cores, "cpu_model": cpu_model(), "kind": kind,
           "sample": f"first {n} pairs of the same workload, {cores} OpenMP threads (affinity mask"
                     f"{'' if quota is None else f', cgroup quota {quota:.1f} CPUs'}; os.cpu_count() = {os.cpu_count()}), "
                     f"one aligner per thread, {wall:.1f} s of wall clock",
           "single_thread_value": single, "per_thread_value": per_thread,
           "reference_anchor_per_core": anchor,
           # a host whose cores run the reference far below the survey's anchor (shared / throttled
           # cores) must be called out, not used to inflate a speed-up
           "suspect": bool(anchor and (single < 0.5 * anchor or per_thread < 0.25 * anchor))}
    return out, scores.astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------
# end to end: host buffers -> HBM (H2D) -> run -> results on the host (D2H), batch after batch
# ---------------------------------------------------------------------------------------------------------------
def e2e_leg(capi, batch, params, fmt, nbatches, expect_checksum, slots=4, inflight=2, uploaders=2, expect_cigar_bytes=None):
    """`nbatches` batches of the same host data through `slots` resident batch objects: `uploaders` uploader threads
    reload (quicked_batch_reload*, H2D) batches k+1.. (uploader u takes the batches with k % uploaders == u) while the main
    thread queues run k (sync = 0) and a fetcher thread brings the results of run k over (quicked_batch_fetch, D2H).
    fmt: ascii_pinned (ASCII pools over the link), 2bit_pinned (2-bit words the caller already holds),
    ascii_hostpacked (ASCII pools packed to 2-bit words by quicked_wire_pack_pool on the uploader threads, inside the
    clock, then shipped).  Returns alignments/s over everything between the first reload and the last fetch; batch
    creation (hipMalloc) and the first upload are warm-up."""
    L = capi.lib()
    n = len(batch)
    frees = []
    pack_threads = 0
    src = None
    if fmt == "ascii_pinned":
        src = capi.pinned_copy(batch)
        make = lambda: capi.ResidentBatch(src)                                       # noqa: E731
        reload_ = lambda rb, slot: rb.reload(src)                                    # noqa: E731
        nbytes = int(batch.pattern_len.astype(np.int64).sum() + batch.text_len.astype(np.int64).sum())
    elif fmt == "2bit_pinned":
        pw, po = capi.wire_pack_pool(batch.pattern_pool, batch.pattern_off, batch.pattern_len, capi.WIRE_2BIT)
        tw, to = capi.wire_pack_pool(batch.text_pool, batch.text_off, batch.text_len, capi.WIRE_2BIT)
        (pwp, h1), (twp, h2) = capi.pinned_array(pw), capi.pinned_array(tw)
        frees += [h1, h2]
        make = lambda: capi.ResidentBatch.from_wire(batch, capi.WIRE_2BIT, pwp, po, twp, to)       # noqa: E731
        reload_ = lambda rb, slot: rb.reload_wire(batch, capi.WIRE_2BIT, pwp, po, twp, to)         # noqa: E731
        nbytes = int(pw.nbytes + tw.nbytes)
    elif fmt == "ascii_hostpacked":
        # the caller holds ASCII (pageable is fine: the CPU reads it); every batch is packed into the slot's own pinned word
        # buffers by the uploader thread that then ships them
        po, ptotal = capi.wire_offsets(batch.pattern_len, capi.WIRE_2BIT)
        to, ttotal = capi.wire_offsets(batch.text_len, capi.WIRE_2BIT)
        cores, _ = usable_cpus()
        cores = max(1, cores // max(1, int(os.environ.get("WORLD_SIZE", "1"))))      # every rank of the node packs: the CPUs are shared
        pack_threads = max(1, cores // max(uploaders, 1))
        bufs = []
        for _ in range(slots):
            (pwp, h1), (twp, h2) = capi.pinned_array(np.zeros(ptotal + 1, np.uint64)), capi.pinned_array(np.zeros(ttotal + 1, np.uint64))
            frees += [h1, h2]
            bufs.append((pwp, twp))

        def pack(slot):
            pwp, twp = bufs[slot]
            capi.wire_pack_pool(batch.pattern_pool, batch.pattern_off, batch.pattern_len, capi.WIRE_2BIT, threads=pack_threads, out=pwp)
            capi.wire_pack_pool(batch.text_pool, batch.text_off, batch.text_len, capi.WIRE_2BIT, threads=pack_threads, out=twp)
            return pwp, twp

        def make():
            pwp, twp = pack(0)
            return capi.ResidentBatch.from_wire(batch, capi.WIRE_2BIT, pwp, po, twp, to)

        def reload_(rb, slot):
            pwp, twp = pack(slot)
            return rb.reload_wire(batch, capi.WIRE_2BIT, pwp, po, twp, to)
        nbytes = int(8 * (ptotal + ttotal))
    else:
        raise ValueError(fmt)
    rbs = []
    try:
        for _ in range(slots):
            rbs.append(make())
    except RuntimeError:
        # no room for `slots` resident batches next to the pools of the resident leg (config 4 holds ~94 GB per pool set)
        for rb in rbs:
            rb.close()
        if src is not None:
            capi.pinned_free(src)
        for h in frees:
            L.quicked_host_free(h)
        raise
    for rb in rbs:                                        # warm: code objects, pools of the rotation (asynchronous runs:
        if rb.run(params, sync=False) < 0 or rb.fetch() < 0:      # the planner's depth for a stream of runs)
            raise RuntimeError("end-to-end warm-up run failed")
    # a queued run leaves its results in its batch object's own memory (include/quicked_batch.h): what bounds the runs in
    # flight is the number of batch objects, not the thread's rotating pool sets
    sets = capi.pool_stats()["sets"]
    inflight = max(1, min(inflight, slots - 1))
    uploaded = [threading.Event() for _ in range(nbatches)]
    fetched = [threading.Event() for _ in range(nbatches)]      # results checked: the batch object may be reloaded
    on_host = [threading.Event() for _ in range(nbatches)]      # quicked_batch_fetch returned
    queued = [threading.Event() for _ in range(nbatches)]
    err = []

    def fail(e):
        err.append(e)
        for ev in uploaded + fetched + on_host + queued:
            ev.set()

    tm = {"reload": 0.0, "run": 0.0, "fetch": 0.0}
    lock = threading.Lock()

    def uploader(u):
        try:
            for k in range(u, nbatches, uploaders):
                if k >= slots:
                    fetched[k - slots].wait()
                if err:
                    return
                tu = time.perf_counter()
                st = reload_(rbs[k % slots], k % slots)
                with lock:
                    tm["reload"] += time.perf_counter() - tu
                if st < 0:
                    raise RuntimeError(f"quicked_batch_reload: {st}")
                uploaded[k].set()
        except Exception as e:      # noqa: BLE001
            fail(e)

    checks = []
    d2h_bytes = [0]

    def finish(k):
        rb = rbs[k % slots]
        tf = time.perf_counter()
        if rb.fetch() < 0:
            raise RuntimeError("quicked_batch_fetch failed")
        tm["fetch"] += time.perf_counter() - tf
        on_host[k].set()                                   # the run's device results are no longer needed: its pool set may be reused
        s, st = rb.scores()
        if not (st >= 0).all():
            raise RuntimeError("end-to-end run: some pairs failed")
        checks.append(int(s.astype(np.int64).sum()))
        if expect_cigar_bytes is not None:                 # CIGAR strings arrive with the fetch (one DMA into pinned memory)
            pool, off = rb.cigar_view()
            if pool.nbytes != expect_cigar_bytes or int((off >= 0).sum()) != n:
                raise RuntimeError("end-to-end CIGARs differ from the resident run's")
            d2h_bytes[0] += pool.nbytes
        fetched[k].set()

    def fetcher():
        # its own thread: waiting for run k and bringing its results over does not keep the main thread from queueing run k+1
        try:
            for k in range(nbatches):
                queued[k].wait()
                if err:
                    return
                finish(k)
        except Exception as e:      # noqa: BLE001
            fail(e)

    t0 = time.perf_counter()
    ths = [threading.Thread(target=uploader, args=(u,)) for u in range(uploaders)]
    for th in ths:
        th.start()
    fth = threading.Thread(target=fetcher)
    fth.start()
    try:
        for k in range(nbatches):
            uploaded[k].wait()
            if k >= inflight:
                on_host[k - inflight].wait()                # at most `inflight` runs queued and not fetched
            if err:
                break
            tr = time.perf_counter()
            if rbs[k % slots].run(params, sync=False) < 0:
                raise RuntimeError("quicked_batch_run failed")
            tm["run"] += time.perf_counter() - tr
            queued[k].set()
    except Exception as e:      # noqa: BLE001
        fail(e)
    fth.join()
    elapsed = time.perf_counter() - t0
    for th in ths:
        th.join()
    for rb in rbs:
        rb.close()
    if src is not None:
        capi.pinned_free(src)
    for h in frees:
        L.quicked_host_free(h)
    if err:
        raise err[0] if isinstance(err[0], RuntimeError) else RuntimeError(repr(err[0]))
    if not all(c == expect_checksum for c in checks):
        raise AssertionError("end-to-end scores differ from the resident run's")
    out = {"value": n * nbatches / elapsed, "unit": "alignments/s", "batches": nbatches, "ms_per_batch": elapsed / nbatches * 1e3,
           "h2d_bytes_per_batch": nbytes, "h2d_GBs": nbytes * nbatches / elapsed / 1e9,
           "host_ms_per_batch": {k: v / nbatches * 1e3 for k, v in tm.items()}, "slots": slots, "inflight": inflight,
           "pool_sets": sets, "uploader_threads": uploaders, "d2h_cigar_bytes_per_batch": d2h_bytes[0] // max(nbatches, 1)}
    if pack_threads:
        out["host_pack_threads_per_uploader"] = pack_threads
        out["host_cpus_per_rank"] = cores
        out["host_pack_kernel"] = {0: "scalar", 1: "avx2+bmi2", 2: "avx512bw+bmi2"}.get(L.quicked_wire_pack_isa(-1), "?")
    return out


def mixed_leg(capi, datagen, pairs, length, error, share, steps=24, slots=4, cpu_budget=0.0):
    """QuickEd + CIGAR on a MIXED batch: `pairs` ordinary pairs of which `share` carry 4 x 800-base indels (they leave the
    fast flow and go through the host-driven stages 2 / 3).  `slots` resident batch objects of the same data, runs queued
    with sync == 0 by this thread, every run fetched by ONE other thread -- the pattern of the end-to-end legs without the
    copies in.  The library's early-finish threads (QE_FINISHERS) align the pairs that left the fast flow."""
    import queue
    hard = min(pairs, max(1, int(pairs * share)))
    batch = datagen.generate(hard, length, error, seed=datagen.DEFAULT_SEED, first=pairs if hard < pairs else 0, indels_num=4, indels_len=800)
    if hard < pairs:
        batch = datagen.generate(pairs - hard, length, error, seed=datagen.DEFAULT_SEED).concat(batch)
    p = capi.make_params(algo=capi.QUICKED)
    rbs = [capi.ResidentBatch(batch) for _ in range(slots)]
    try:
        checks = []
        for rb in rbs:
            if rb.run(p, sync=True) < 0:
                raise RuntimeError("quicked_batch_run failed")
            gpu_scores = rb.scores()[0]
            checks.append(int(gpu_scores.astype("int64").sum()))
        for _ in range(2):                                  # the fast flow, twice, outside the clock: pools, estimates, and both host-side
            for rb in rbs:                                  # result sets of every batch object (its first early finish allocates the second:
                rb.run(p, sync=False)                       # ~0.1 s of pinned memory for 100 k pairs' strings)
            for rb in rbs:
                rb.fetch()
        todo, free, deferred, bad = queue.Queue(), queue.Queue(), [], []
        for rb in rbs:
            free.put(rb)

        def fetcher():
            while True:
                rb = todo.get()
                if rb is None:
                    return
                try:
                    if rb.fetch() < 0:
                        bad.append("fetch")
                    deferred.append(int(rb.deferred_pairs()))
                    if int(rb.scores()[0].astype("int64").sum()) != checks[0]:
                        bad.append("checksum")
                except Exception as e:      # noqa: BLE001  (the queueing thread must get its batch object back whatever happens)
                    bad.append(repr(e))
                free.put(rb)

        th = threading.Thread(target=fetcher)
        th.start()
        t0 = time.perf_counter()
        try:
            for _ in range(steps):
                rb = free.get(timeout=300)
                if rb.run(p, sync=False) < 0:
                    bad.append("run")
                todo.put(rb)
        except queue.Empty:
            bad.append("no batch object came back within 300 s")
        finally:
            todo.put(None)
            th.join(timeout=300)
        dt = time.perf_counter() - t0
    finally:
        for rb in rbs:
            rb.close()
    if bad or len(set(checks)) != 1:
        return {"error": f"mixed leg failed: {sorted(set(bad))}"}
    out = {"value": pairs * steps / dt, "unit": "alignments/s", "ms_per_batch": dt / steps * 1e3, "pairs_per_gpu": pairs, "batches": steps,
            "hard_pairs": hard, "pairs_finished_outside_the_fast_flow_per_run": max(deferred) if deferred else 0,
            "batch_objects": slots, "fetching_threads": 1, "early_finish_threads": int(os.environ.get("QE_FINISHERS", "3")),
            "score_checksum": checks[0],
            "data": f"{hard} of {pairs} pairs with 4 x 800-base indels on top of the {error * 100:g} % edits, the rest as in the headline",
            "note": "a stream of queued runs over resident batch objects, every run fetched (results on the host side of the "
                    "library, strings left in the batch's pinned pool); the pairs that leave the fast flow are aligned through "
                    "the host-driven stages by the library's early-finish threads"}
    if cpu_budget > 0:
        # the reference on the host cores over a sample that keeps the mix (every k-th pair: the hard ones are at the end)
        cores, _ = usable_cpus()
        sample, idx = strided_sample(batch, max(cores * 8, min(pairs, int(cpu_budget * cores * 400))))
        try:
            base, ref_scores = cpu_baseline(sample, dict(algo=capi.QUICKED, only_score=False, bandwidth=15), "quicked_mixed",
                                            budget_s=cpu_budget, anchored=False)
            k = len(ref_scores)
            base["sample"] = f"every {max(1, pairs // len(idx))}-th pair of the batch ({len(idx)} pairs, {int((idx[:k] >= pairs - hard).sum())} of them hard); " + base["sample"]
            base["gpu_scores_identical_on_sample"] = bool((gpu_scores[idx[:k]].astype(np.int64) == ref_scores).all())
            out["cpu_baseline"] = base
        except Exception as e:          # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


class Bench:
    """one rank's legs; every leg reduces over the ranks through quicked_amd/shard.py"""

    def __init__(self, args, rank, world, local_rank, dist, torch, device, share):
        from quicked_amd import capi, datagen
        self.args, self.rank, self.world, self.dist, self.torch, self.device = args, rank, world, dist, torch, device
        self.capi, self.datagen = capi, datagen
        self._cache = None
        self.parallelism = f"pairs sharded over {world} GPU(s), no data-path collective" + (" [test: ranks share device 0]" if share else "")
        if capi.lib().quicked_set_device(local_rank) < 0:
            sys.exit(f"bench.py: rank {rank}: no HIP device {local_rank}")

    def kw(self, workload):
        capi = self.capi
        if workload == "banded_score":
            return dict(algo=capi.BANDED, only_score=True, bandwidth=self.args.bandwidth)
        return dict(algo=capi.QUICKED, only_score=False, bandwidth=self.args.bandwidth)

    def barrier(self, rb):
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()
        rb.sync()

    def reduce(self, pairs, cells, checksum, elapsed, extra=()):
        return shard.reduce_totals(self.dist, self.torch, self.device, pairs, cells, checksum, elapsed, extra_sum=extra)

    # -----------------------------------------------------------------------------------------------------------
    def timed_resident(self, workload, first, count, steps, warmup, solo_steps=3, kind=None):
        """the contract's loop: `warmup` untimed steps, then exactly `steps` steps between two barrier + synchronize.
        -> dict(batch, scores, counters, elapsed, kernel ms overlapped / solo, cigar bytes, flow, latency)"""
        args, capi = self.args, self.capi
        params = capi.make_params(**self.kw(workload))
        if self._cache is None or self._cache[0] != (first, count):      # the workloads of one line run on the same pairs
            self._cache = ((first, count), self.datagen.generate(count, args.length, args.error, seed=args.seed, first=first,
                                                                 indels_num=args.indels_num, indels_len=args.indels_len))
        batch = self._cache[1]
        rb = capi.ResidentBatch(batch)           # H2D happens here, outside the timed region
        quick = workload == "quicked"
        if kind is None:
            kind = 1 if quick else 0             # quicked_batch_kernel_times: 0 score-only passes, 1 fills, 2 Hirschberg half passes
        flow = {}

        def kernel_time():
            ms, n = rb.kernel_times()
            return float(ms[kind]), int(n[kind])
        saved_fast = os.environ.get("QE_QUICKED_FAST")

        def run_checked(sync):
            st = rb.run(params, sync=sync)
            if st < 0:
                raise RuntimeError(f"quicked_batch_run failed: {capi.lib().quicked_status_msg(st).decode().strip()}")

        def timed_loop():
            self.barrier(rb)
            t0 = time.perf_counter()
            for _ in range(steps):
                run_checked(args.sync_each_step)      # results stay resident in HBM; the driver still syncs where a stage needs host decisions
            rb.sync()
            self.barrier(rb)
            return time.perf_counter() - t0

        try:
            # QuickEd: at least two untimed synchronous runs -- the first run of a batch is always a classic one and sets
            # the bound estimate, the second shows whether the fast flow defers pairs to the fetch
            for _ in range(max(warmup, 2 if quick else 0)):
                run_checked(True)
            if quick:
                # A QuickEd run queued with sync = 0 leaves the pairs that go past stage 1 (or exceed the planned buffers) to
                # the fetch.  The timed loop never fetches: if this data has such pairs, time the host-driven flow instead,
                # which does all the work inside the run.
                wc = rb.counters()
                flow["stage2_pairs"], flow["stage3_pairs"] = int(wc[6]), int(wc[7])
                flow["deferred_pairs"] = rb.deferred_pairs()
                if flow["stage2_pairs"] or flow["deferred_pairs"]:
                    os.environ["QE_QUICKED_FAST"] = "0"; capi.reload_env()
                    flow["timed_flow"] = "classic (host-driven stages): pairs leave stage 1 on this data"
                else:
                    flow["timed_flow"] = ("stage-1 rule on the device, align step queued with it, where no pair may split (else host-driven "
                                          "stage by stage: reads of >~ 20 kb); no pair deferred to the fetch")
            if not args.sync_each_step:              # asynchronous warm-up: a stream of runs rotates over more sets than a
                run_checked(False)                   # synchronous one (planner, qe_driver.hip); every set allocates once
                rb.sync()
                for _ in range(capi.pool_stats()["sets"]):
                    run_checked(False)
                rb.sync()
            kernel_time()                            # drop the warm-up launches
            elapsed = timed_loop()
            kern_ms, kern_n = kernel_time()
            sets = capi.pool_stats()["sets"]
            # one synchronous run to fetch results + work counters for the report ...
            tl0 = time.perf_counter()
            run_checked(True)
            latency = time.perf_counter() - tl0
            if quick and os.environ.get("QE_QUICKED_FAST") != "0" and rb.deferred_pairs():
                # ... and to catch what the warm-up runs could not: pairs deferred to a fetch the timed loop never made
                os.environ["QE_QUICKED_FAST"] = "0"; capi.reload_env()
                flow["timed_flow"] = "classic (host-driven stages): the fast flow deferred pairs to the fetch on this data (re-timed)"
                flow["deferred_pairs"] = rb.deferred_pairs()
                run_checked(True)
                kernel_time()
                elapsed = timed_loop()
                kern_ms, kern_n = kernel_time()
                run_checked(True)
            scores, status = rb.scores()
            assert (status >= 0).all(), "some pairs failed"
            counters = rb.counters()
            cig = capi.lib().quicked_batch_cigar_bytes(rb._h) if quick else None
            # ... then the dominant kernel ALONE on the chip (no other run in flight): what the roofline fraction divides by
            kernel_time()
            lat = []
            for _ in range(solo_steps):
                tl0 = time.perf_counter()
                run_checked(True)
                lat.append(time.perf_counter() - tl0)
            solo_ms, solo_n = kernel_time()
            latency = min([latency] + lat)
        finally:
            if saved_fast is None:
                os.environ.pop("QE_QUICKED_FAST", None)
            else:
                os.environ["QE_QUICKED_FAST"] = saved_fast
            capi.reload_env()
            rb.close()
        return dict(batch=batch, scores=scores, counters=counters, elapsed=elapsed, kern_ms=kern_ms, kern_n=kern_n,
                    solo_ms=solo_ms, solo_n=solo_n, cigar_bytes=cig, flow=flow, sets=sets, latency_s=latency, params=params)

    # -----------------------------------------------------------------------------------------------------------
    def roofline(self, workload, r, pairs, step_s):
        """roofline + valu objects of one resident leg"""
        args = self.args
        batch, counters = r["batch"], r["counters"]
        per_launch_bytes = float((batch.pattern_len.astype(np.int64) + batch.text_len.astype(np.int64) + 4).sum())
        solo_s = (r["solo_ms"] / 1e3 / r["solo_n"]) if r["solo_n"] else float("nan")
        over_s = (r["kern_ms"] / 1e3 / r["kern_n"]) if r["kern_n"] else float("nan")
        extra = {}
        instr_per_bc = INSTR_PER_BLOCK_COLUMN
        if workload == "banded_score":
            # B_so = plen + tlen + 4 per pair (ASCII in, int32 score out)
            kernel, alg_bytes, work_blocks = "k_banded<false> (BandEd score-only)", per_launch_bytes, int(counters[0])
        elif r.get("kind") == 2:
            # Long reads: the dominant launches are the score-only half passes of Hirschberg's split levels
            # (bpm_hirschberg.c:85-100), G lanes per alignment with the band state in LDS.  One level = a forward and a reverse
            # launch that together read every pair's bit-planes once (3 bits per base) and leave one stopped band per node:
            # per launch half of that.  counters[0] = block-columns of all half passes of the step.
            launches_per_step = max(1, r["solo_n"] // max(1, r.get("solo_steps", 1)))
            alg_bytes = 0.375 * float((batch.pattern_len.astype(np.int64) + batch.text_len.astype(np.int64)).sum()) / 2.0
            kernel, work_blocks = "k_banded_coop_lds<false> (Hirschberg half passes, band state in LDS)", int(counters[0])
            instr_per_bc = INSTR_PER_BLOCK_COLUMN_COOP
            extra["launches_per_step"] = launches_per_step
        else:
            # The fill stores a 16-byte checkpoint per (slot, 16 columns) and the 16-byte carry words per (slot, chunk):
            # 1.25 B per block-column (round 2: a checkpoint every 8 columns, 2.25 B), and reads its inputs as bit-planes
            # (3 bits per base); the traceback recomputes 16-column tiles from those.  SURVEY 8(d)'s figure (every column stored, 16 B per block-column and per
            # traceback step) is what the reference's layout would move: kept as survey_equivalent_bytes, never divided
            # by the time of a kernel that does not move those bytes
            planes_in = 0.375 * float((batch.pattern_len.astype(np.int64) + batch.text_len.astype(np.int64)).sum())
            alg_bytes = planes_in + FILL_BYTES_PER_BLOCK_COLUMN * float(counters[1])
            extra["survey_equivalent_bytes"] = per_launch_bytes + 16.0 * counters[1] + 16.0 * counters[3] + float(counters[4])
            kernel, work_blocks = "k_banded<true> (BandEd fill, checkpointed)", int(counters[1])
        traffic, traffic_src = None, None
        try:      # HBM bytes per launch from the committed PMC passes of this same command (tools/collect_profiles.sh) ...
            import hashlib
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
                pm = json.load(f)
            with open(os.path.join(ROOT, "quicked_amd", "csrc", "qe_kernels.hip"), "rb") as f:
                now = hashlib.sha256(f.read()).hexdigest()
            ent = pm.get(f"{'cfg4' if r.get('kind') == 2 else workload}:{pairs}x{args.length}")
            if pm.get("kernels_sha256") != now:      # ... of THESE kernels: a figure measured with another qe_kernels.hip is not printed
                traffic_src = "stale: profiles/pmc_traffic_latest.json was collected with another qe_kernels.hip (re-run tools/collect_profiles.sh)"
            elif ent:
                traffic, traffic_src = ent["hbm_bytes"], ent.get("source")
        except Exception:      # noqa: BLE001
            traffic = None
        achieved = alg_bytes / solo_s / 1e9
        # the VALU view compares a STEP's block-columns with a step's worth of the dominant launches (kind 2: several per step)
        solo_step_s = solo_s * extra.get("launches_per_step", 1)
        # work_blocks counts block-columns per LANE (one alignment); a wave64 instruction serves 64 of them
        wave_instr = work_blocks / 64.0 * instr_per_bc
        peak_instr = SIMDS * PEAK_CLOCK_HZ / 2.0
        roof = dict({"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms": solo_s * 1e3, "kernel_ms_source": f"HIP events around the launch, {r['solo_n']} launches, one run at a time",
                     "kernel_ms_overlapped": over_s * 1e3, "runs_in_flight": r["sets"],
                     "aggregate_achieved": alg_bytes / step_s / 1e9, "aggregate_frac": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                     "note": "the BandEd kernels are integer-VALU-bound, not HBM-bound (SURVEY 8d); see valu.  frac = algorithmic bytes / "
                             "the kernel's duration alone on the chip; aggregate_* divide by the step time (launches of consecutive "
                             "runs overlap, so a launch inside the timed region lasts kernel_ms_overlapped)"}, **extra)
        valu = {"bound": "VALU issue, one wave64 instruction per SIMD per 2 cycles", "unit": "wave-instructions/s",
                "peak": peak_instr, "aggregate_achieved": wave_instr / step_s, "aggregate_frac": wave_instr / step_s / peak_instr,
                "solo_kernel_frac": wave_instr / solo_step_s / peak_instr,
                "block_columns_per_launch": work_blocks, "instructions_per_block_column": instr_per_bc,
                "issue_cycles_per_block_column": ISSUE_CYCLES_PER_BLOCK_COLUMN,
                "instruction_mix_bound_block_columns_per_s": SIMDS * PEAK_CLOCK_HZ * 64 / ISSUE_CYCLES_PER_BLOCK_COLUMN,
                "aggregate_block_columns_per_s": work_blocks / step_s,
                # what the pass's own instruction stream does on registers, no memory, two waves per SIMD (the occupancy the
                # kernel runs at): profiles/r06_b_skew_asm_bench.txt / r06_b_issue_classes.md -- every variant of the pass issues
                # at 3.2-3.6 cycles per instruction there whatever its class mix, not at the nominal 2 / 4, so this, not
                # instruction_mix_bound, is the loop's real bound
                "measured_loop_block_columns_per_s": MEASURED_LOOP_BLOCK_COLUMNS_PER_S,
                "aggregate_frac_of_measured_loop": work_blocks / step_s / MEASURED_LOOP_BLOCK_COLUMNS_PER_S,
                "note": "aggregate = per-step work / step time; peak assumes the 2.4 GHz peak clock (the chip holds less under this "
                        "load: DESIGN.md 4.1)"}
        return roof, valu, work_blocks

    # -----------------------------------------------------------------------------------------------------------
    def e2e(self, workload, r, checksum):
        args, capi = self.args, self.capi
        quick = workload != "banded_score"
        out = {}
        for fmt in ("ascii_pinned", "2bit_pinned", "ascii_hostpacked"):
            res = None
            # QuickEd's run call does more host work per run: one more run in flight and one more uploader hide it
            # the host-packed leg packs on the uploader threads: one more of them
            want = (args.e2e_slots or (6 if quick else (5 if fmt == "ascii_hostpacked" else 4)), args.e2e_inflight or (3 if quick else 2),
                    args.e2e_uploaders or (3 if (quick or fmt == "ascii_hostpacked") else 2))
            for slots, inflight, uploaders in (want, (3, 2, 1)):
                try:
                    res = e2e_leg(capi, r["batch"], r["params"], fmt, args.e2e_batches, checksum, slots=slots, inflight=inflight,
                                  uploaders=uploaders, expect_cigar_bytes=r["cigar_bytes"])
                    break
                except (RuntimeError, MemoryError) as e:
                    print(f"[bench] end-to-end leg ({workload}, {fmt}, {slots} resident batches): {e}", file=sys.stderr)
            if res is None:      # HBM is held by the resident leg's pools: reported, not hidden
                res = {"value": 0.0, "h2d_GBs": 0.0, "skipped": "no HBM left for resident batch objects next to the run's device pools"}
            _, _, _, _, ext = self.reduce(0, 0, 0, 0.0, extra=(res["value"], res["h2d_GBs"]))
            res["value"], res["h2d_GBs"] = ext[0], ext[1]
            out[fmt] = res
        out["note"] = ("reload (H2D) of batch k+1 overlapped with the run of batch k, scores (and CIGAR strings, where the workload "
                       "makes them) fetched (D2H) per batch; summed over ranks.  ascii_pinned ships the bytes the reference's API "
                       "takes (PCIe Gen5 x16 moves ~47-55 GB/s from pinned memory: ~2.7 M alignments/s of 10 kb); 2bit_pinned is a "
                       "caller that already holds 2-bit words; ascii_hostpacked packs the caller's ASCII to 2-bit words on the host "
                       "cores inside the clock (quicked_wire_pack_pool) and ships those")
        return out

    # -----------------------------------------------------------------------------------------------------------
    def workload_object(self, workload, pairs, steps, warmup, with_e2e, with_cpu, scaling="weak", kind=None, solo_steps=3):
        """one workload at `pairs` pairs per GPU (weak) or in total (strong): resident rate (+ roofline, e2e, cpu baseline)"""
        args = self.args
        first, count, _ = shard.plan(pairs, self.rank, self.world, scaling)
        r = self.timed_resident(workload, first, count, steps, warmup, solo_steps=solo_steps, kind=kind)
        r["kind"], r["solo_steps"] = kind, solo_steps
        cells = r["batch"].cells()
        checksum = int(r["scores"].astype(np.int64).sum())
        tot_pairs, tot_cells, tot_checksum, max_elapsed, _ = self.reduce(count, cells, checksum, r["elapsed"])
        step_s = max_elapsed / steps
        roof, valu, work_blocks = self.roofline(workload, r, pairs, step_s)
        obj = {"value": tot_pairs * steps / max_elapsed, "unit": "alignments/s", "ms_per_step": step_s * 1e3,
               "gcups": tot_cells * steps / max_elapsed / 1e9,
               # cells actually computed (SURVEY 8d "band GCUPS"): 64 rows x block-advances of the dominant kernel, this rank x world
               "band_gcups": 64.0 * work_blocks * self.world * steps / max_elapsed / 1e9,
               "pairs_per_gpu": count, "total_pairs": tot_pairs, "scaling": scaling, "steps": steps,
               "runs_in_flight": r["sets"], "single_batch_latency_ms": r["latency_s"] * 1e3,
               "single_batch_value": count / r["latency_s"],
               "score_checksum": tot_checksum, "roofline": roof, "valu": valu}
        if r["flow"]:
            obj["quicked_flow"] = r["flow"]
        if r["cigar_bytes"] is not None:
            obj["cigar_bytes_per_step"] = int(r["cigar_bytes"])
        if with_e2e:
            obj["e2e"] = self.e2e(workload, r, checksum)
        if with_cpu and self.rank == 0 and self.world == 1:      # the CPU reference is timed on rank 0 at N = 1 only
            base, ref_scores = cpu_baseline(r["batch"], dict(self.kw(workload)), workload, budget_s=args.cpu_budget,
                                            anchored=(args.length == 10000 and abs(args.error - 0.05) < 1e-9 and args.indels_num == 0))
            n = len(ref_scores)
            base["gpu_scores_identical_on_sample"] = bool((r["scores"][:n].astype(np.int64) == ref_scores).all())
            obj["cpu_baseline"] = base
        return obj


def quicked_score_leg(B, args, expect_checksum):
    """QuickEd with only_score on the line's pairs (still cached from workloads.quicked): queued runs as the headline loop
    queues them, one batch alone; the scores must be workloads.quicked's"""
    capi = B.capi
    batch = B._cache[1]
    rb = capi.ResidentBatch(batch)
    try:
        params = capi.make_params(algo=capi.QUICKED, only_score=True, bandwidth=args.bandwidth)

        def run(sync):
            st = rb.run(params, sync=sync)
            if st < 0:
                raise RuntimeError(f"quicked_batch_run failed: {capi.lib().quicked_status_msg(st).decode().strip()}")
        for _ in range(2):
            run(True)
        checksum = int(rb.scores()[0].astype(np.int64).sum())
        lat = []
        for _ in range(3):
            t0 = time.perf_counter(); run(True); lat.append(time.perf_counter() - t0)
        for _ in range(6):
            run(False)
        rb.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run(False)
        rb.sync()
        dt = time.perf_counter() - t0
        rb.fetch()
        cnt = rb.counters()
        return {"value": len(batch) * args.steps / dt, "unit": "alignments/s", "ms_per_step": dt / args.steps * 1e3,
                "gcups": batch.cells() * args.steps / dt / 1e9, "pairs_per_gpu": len(batch), "steps": args.steps,
                "single_batch_latency_ms": min(lat) * 1e3, "score_checksum": checksum,
                "scores_equal_workloads_quicked": (checksum == expect_checksum) if expect_checksum is not None else None,
                "traceback_steps": int(cnt[3]), "pairs_finished_outside_the_fast_flow": int(rb.deferred_pairs()),
                "data": "QuickEd, only_score: WindowEd bound, then one score-only BandEd pass over the fill's cells at that cutoff"}
    finally:
        rb.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=100000, help="pairs per GPU per step (weak scaling: the headline value)")
    ap.add_argument("--length", type=int, default=10000)
    ap.add_argument("--error", type=float, default=0.05)
    ap.add_argument("--bandwidth", type=int, default=15)
    ap.add_argument("--indels-num", type=int, default=0, help="large indels per pair (generate_dataset's -I), with --indels-len")
    ap.add_argument("--indels-len", type=int, default=0)
    ap.add_argument("--workload", choices=["banded_score", "quicked"], default="banded_score",
                    help="the headline workload; with the default, QuickEd + CIGAR is also measured (workloads.quicked)")
    ap.add_argument("--no-workloads", action="store_true", help="headline workload only (no workloads.quicked object)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of wall clock for the CPU baseline's sample")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-batches", type=int, default=48,
                    help="batches per end-to-end leg (pipeline fill and drain are inside the clock: ~3 batches' worth)")
    ap.add_argument("--e2e-slots", type=int, default=0, help="resident batch objects in rotation (0: 4 for score-only BandEd, 6 for QuickEd)")
    ap.add_argument("--e2e-uploaders", type=int, default=0, help="uploader threads (0: 2 for score-only BandEd, 3 for QuickEd)")
    ap.add_argument("--e2e-inflight", type=int, default=0, help="runs queued and not yet fetched (0: 2 for score-only BandEd, 3 for QuickEd)")
    ap.add_argument("--no-strong", action="store_true", help="no strong (N > 1) / strong_share (N = 1) legs")
    ap.add_argument("--mixed-share", type=float, default=0.01,
                    help="share of large-indel pairs in the mixed QuickEd leg (workloads.quicked_mixed; 0: no such leg)")
    ap.add_argument("--indel-pairs", type=int, default=20000,
                    help="pairs of the indel-heavy QuickEd leg (4 x 800-base indels per 10 kb pair: stages 2 / 3 and band doubling, "
                         "SURVEY's own trigger set); 0: no such leg")
    ap.add_argument("--cfg4-pairs", type=int, default=10000,
                    help="pairs of the long-read leg (BASELINE.json configs[3]: 100 kb at 10 %% error, QuickEd + Hirschberg CIGAR; "
                         "workloads.cfg4, N = 1); 0: no such leg")
    ap.add_argument("--cfg5-pairs", type=int, default=None,
                    help="pairs IN TOTAL of BASELINE.json configs[4] (QuickEd score + CIGAR, 10 kb at 5 %%, sharded over the GPUs; "
                         "workloads.cfg5 at N > 1, the 1/8 shard as workloads.cfg5_shard at N = 1); 0: no such leg; default: 1 000 000 on "
                         "10 kb reads")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="profiling aid: no overlap between consecutive runs, so per-kernel durations are those of a kernel alone")
    ap.add_argument("--seed", type=int, default=0x51CED)
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves, as a child, before this process touches the GPU (sysfs only here)
        have = visible_gpus()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); refusing to run fewer ranks "
                     "than asked for")
        sys.exit(shard.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match the launcher's WORLD_SIZE {world}")
    dist = None
    torch = None
    device = None
    # test hooks (tests/test_gpu_parity.py runs the whole N = 2 path on a 1-GPU box): QE_BENCH_SHARE_GPU=1 puts every rank
    # on device 0, QE_BENCH_BACKEND=gloo reduces over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get("QE_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("QE_BENCH_BACKEND", "nccl")
    if share:
        local_rank = 0
    if world > 1 or os.environ.get("QE_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        if torch.cuda.device_count() <= local_rank:
            sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, this node exposes {torch.cuda.device_count()}")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)                    # reductions on host tensors (device = None)

    collective = {"backend": "none" if dist is None else backend, "tensors": "device" if device is not None else "host",
                  "world_size": world} if dist is not None else {"backend": "none"}
    B = Bench(args, rank, world, local_rank, dist, torch, device, share)
    ranks_seen = shard.count_ranks(dist, torch, device)

    # ---- the headline: weak scaling, `pairs` pairs per GPU, inputs resident in HBM
    head = B.workload_object(args.workload, args.pairs, args.steps, args.warmup, with_e2e=not args.no_e2e,
                             with_cpu=not args.no_cpu_baseline)
    default_shape = args.workload == "banded_score" and not args.no_workloads
    others = {}
    if default_shape:
        # configs[2] on the same pairs: QuickEd bound-and-align + CIGAR (the HBM-relevant workload of the path)
        args.cpu_budget = min(args.cpu_budget, 6.0)
        others["quicked"] = B.workload_object("quicked", args.pairs, args.steps, args.warmup, with_e2e=not args.no_e2e,
                                              with_cpu=not args.no_cpu_baseline)
    if default_shape and world == 1:
        # the same pairs once more with only_score: the reference aligns all the same and counts the CIGAR's edits
        # (quicked.c:283-294); here the scores come from one score-only pass over the fill's cells
        try:
            others["quicked_score"] = quicked_score_leg(B, args, others["quicked"].get("score_checksum"))
        except Exception as e:      # noqa: BLE001  (a leg of its own: the line survives it)
            others["quicked_score"] = {"error": repr(e)}
    if default_shape and args.indel_pairs > 0 and args.indels_num == 0 and world == 1:
        # QuickEd's stages 2 / 3 (quicked.c:204-280): pairs with large indels leave stage 1, go through WindowEd(L) forward
        # and reverse and, most of them, through score-only BandEd with band doubling before the align step.  Host-driven
        # regrouping here (qe_driver.hip, quicked_classic); timed so that the rate is on record next to the uniform one
        saved = (args.indels_num, args.indels_len)
        args.indels_num, args.indels_len = 4, 800
        B._cache = None
        budget = args.cpu_budget
        args.cpu_budget = min(budget, 3.0)
        o = B.workload_object("quicked", args.indel_pairs, min(args.steps, 10), 2, with_e2e=False, with_cpu=not args.no_cpu_baseline)
        args.cpu_budget = budget
        args.indels_num, args.indels_len = saved
        B._cache = None
        others["quicked_indels"] = {k: o[k] for k in ("value", "unit", "ms_per_step", "pairs_per_gpu", "steps", "runs_in_flight",
                                                       "single_batch_latency_ms", "quicked_flow", "score_checksum", "cpu_baseline") if k in o}
        others["quicked_indels"]["data"] = "4 x 800-base indels per pair on top of the 5 % edits (generate_dataset's -I 4 -L 800)"
    if default_shape and args.cfg4_pairs > 0 and world == 1 and args.length == 10000:
        # BASELINE.json configs[3]: QuickEd + Hirschberg CIGAR on long reads (100 kb at 10 % error).  Every pair splits
        # (bpm_hirschberg.c:63-65); the dominant launches are the split levels' score-only half passes.  From empty pools:
        # one pool set of this workload holds ~90 GB.
        saved = (args.length, args.error, args.cpu_budget)
        args.length, args.error, args.cpu_budget = 100000, 0.10, min(args.cpu_budget, 4.0)
        B._cache = None
        B.capi.pool_trim()
        try:
            o = B.workload_object("quicked", args.cfg4_pairs, 5, 2, with_e2e=False, with_cpu=not args.no_cpu_baseline, kind=2, solo_steps=1)
            others["cfg4"] = {k: o[k] for k in ("value", "unit", "ms_per_step", "gcups", "band_gcups", "pairs_per_gpu", "steps", "runs_in_flight",
                                                "single_batch_latency_ms", "quicked_flow", "score_checksum", "cigar_bytes_per_step",
                                                "roofline", "valu", "cpu_baseline") if k in o}
            others["cfg4"]["data"] = "BASELINE.json configs[3]: 100 kb ONT-like reads at 10 % error, QuickEd + Hirschberg CIGAR"
        except Exception as e:          # noqa: BLE001  (a leg of its own: the line survives it)
            others["cfg4"] = {"error": repr(e)}
        args.length, args.error, args.cpu_budget = saved
        B._cache = None
        B.capi.pool_trim()
    if args.cfg5_pairs is None:                       # the default: configs[4]'s own size on its own kind of data
        args.cfg5_pairs = 1000000 if args.length == 10000 else 0
    if default_shape and args.cfg5_pairs > 0 and args.indels_num == 0:
        # BASELINE.json configs[4]: QuickEd score + CIGAR, 1 M pairs of 10 kb at 5 % sharded over the GPUs of the node
        # (align_benchmark.c:246-284: disjoint pair ranges, one aligner each; here contiguous ranges of the seeded dataset per
        # rank, no data-path collective, the totals through one all-reduce).  N > 1: the whole job, `cfg5_pairs` in total
        # (125 k pairs per GPU at 8; at most CFG5_MAX_PER_GPU per GPU, so N = 2 / 3 run a smaller total and say so).  N = 1: the
        # shard one GPU of eight gets, as `cfg5_shard`.
        per_gpu, total = shard.config5_plan(args.cfg5_pairs, world, CFG5_MAX_PER_GPU)
        B._cache = None
        B.capi.pool_trim()
        o = B.workload_object("quicked", per_gpu, min(args.steps, 10), 2, with_e2e=False, with_cpu=False)
        name = "cfg5" if world > 1 else "cfg5_shard"
        others[name] = {k: o[k] for k in ("value", "unit", "ms_per_step", "gcups", "band_gcups", "pairs_per_gpu", "total_pairs", "steps",
                                          "runs_in_flight", "single_batch_latency_ms", "single_batch_value", "quicked_flow",
                                          "score_checksum", "cigar_bytes_per_step") if k in o}
        others[name]["data"] = (f"BASELINE.json configs[4]: QuickEd score + CIGAR, {total} pairs of {args.length} bp at {args.error:g} error in total over {world} GPU(s), "
                                f"{per_gpu} per GPU and step" + ("" if world > 1 else f" (the share of one GPU of 8 of {args.cfg5_pairs} pairs)"))
        B._cache = None
        B.capi.pool_trim()
    # ---- strong scaling: `pairs` pairs in total (BASELINE.json's "100 k pairs at 8 GPUs"); at N = 1 the per-GPU share of it
    strong, strong_share = None, None
    if not args.no_strong:
        wls = [args.workload] + [w for w in others if w == "quicked"]
        if world > 1:
            strong = {}
            for wl in wls:
                # per-rank batches are small here: a stream long enough to amortise filling / draining the rotation (see strong_share)
                o = B.workload_object(wl, args.pairs, 4 * max(args.steps, 40), 1, with_e2e=False, with_cpu=False, scaling="strong")
                strong[wl] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "total_pairs", "pairs_per_gpu", "score_checksum",
                                                 "runs_in_flight", "single_batch_latency_ms")}
            # the headline workload's figures at the top of the object, as in round 2's line
            strong.update({"scaling": "strong", **strong[args.workload]})
        elif args.pairs > STRONG_SHARE_PAIRS:
            strong_share = {"pairs_per_gpu": STRONG_SHARE_PAIRS, "of": "100 k pairs in total over 8 GPUs (BASELINE.json north_star)",
                            "note": "`value` = steady rate of a stream of such batches (runs_in_flight of them on the device at once); "
                                    "single_batch_value = one batch alone, synchronous (what a caller with exactly 100 k pairs sees)"}
            for wl in wls:
                # a stream: long enough that filling and draining the rotation (one launch's duration with runs_in_flight of
                # them on the device, ~18 ms against 2.2 ms per step) is a few per cent of the timed region, not a fifth
                o = B.workload_object(wl, STRONG_SHARE_PAIRS, 4 * max(args.steps, 40), 1, with_e2e=False, with_cpu=False)
                strong_share[wl] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "runs_in_flight", "single_batch_latency_ms",
                                                       "single_batch_value", "score_checksum")}
                strong_share[wl]["aggregate_block_columns_per_s"] = o["valu"]["aggregate_block_columns_per_s"]
            if "quicked" in wls and world == 1:
                # the share with only_score (the 12.5 k pairs are still cached): two chains per batch instead of three
                saved_steps = args.steps
                try:
                    args.steps = 4 * max(saved_steps, 40)
                    o = quicked_score_leg(B, args, strong_share["quicked"]["score_checksum"])
                    strong_share["quicked_score"] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "single_batch_latency_ms",
                                                                         "score_checksum", "scores_equal_workloads_quicked", "traceback_steps")}
                except Exception as e:      # noqa: BLE001
                    strong_share["quicked_score"] = {"error": repr(e)}
                finally:
                    args.steps = saved_steps
            if "quicked" in wls and args.mixed_share > 0 and args.indels_num == 0:
                # the same share with realistic data: 1 % of every 12.5 k-pair batch leaves the fast flow.  A flow for a hundred
                # pairs lasts as long as one for thousands (launch latency), so the early-finish threads serve the pairs of
                # every run that is over with ONE flow (quicked_early_finish_stats: merged flows)
                B._cache = None
                B.capi.pool_trim()
                try:
                    before = B.capi.early_finish_stats()
                    o = mixed_leg(B.capi, B.datagen, STRONG_SHARE_PAIRS, args.length, args.error, args.mixed_share, steps=96, slots=14)
                    after = B.capi.early_finish_stats()
                    strong_share["quicked_mixed"] = {k: o[k] for k in ("value", "unit", "ms_per_batch", "batches", "batch_objects", "hard_pairs",
                                                                         "pairs_finished_outside_the_fast_flow_per_run", "early_finish_threads",
                                                                         "score_checksum") if k in o} or o
                    strong_share["quicked_mixed"]["early_finish_flows"] = {k: after[k] - before[k] for k in after}
                except Exception as e:      # noqa: BLE001
                    strong_share["quicked_mixed"] = {"error": repr(e)}

    if default_shape and args.mixed_share > 0 and args.indels_num == 0 and world == 1:
        # ordinary pairs with a few large-indel ones among them: what the fast flow's overflow path and the early-finish
        # threads are for.  Last, and from empty pools: four more resident batches and buffers for twice the
        # usual bounds next to everything the legs above have grown would be a leg about memory pressure
        B._cache = None
        B.capi.pool_trim()
        try:
            others["quicked_mixed"] = mixed_leg(B.capi, B.datagen, args.pairs, args.length, args.error, args.mixed_share,
                                                cpu_budget=0.0 if args.no_cpu_baseline else min(args.cpu_budget, 3.0))
        except Exception as e:          # noqa: BLE001  (a leg of its own: the line survives it)
            others["quicked_mixed"] = {"error": repr(e)}
        if "quicked_indels" in others:
            # the indel-heavy pairs once more, as such a fetched stream (fast flow for the pairs that stay in stage 1, the
            # early-finish threads for the rest, several batches at a time) next to the host-driven resident loop above
            try:
                B.capi.pool_trim()
                o = mixed_leg(B.capi, B.datagen, args.indel_pairs, args.length, args.error, 1.0, steps=16, slots=6)
                others["quicked_indels"]["fetched_stream"] = {k: o[k] for k in ("value", "unit", "ms_per_batch", "batches", "batch_objects",
                                                                                "fetching_threads", "early_finish_threads",
                                                                                "pairs_finished_outside_the_fast_flow_per_run") if k in o} or o
            except Exception as e:      # noqa: BLE001
                others["quicked_indels"]["fetched_stream"] = {"error": repr(e)}
    line = None
    if rank == 0:
        label = f"{args.length / 1000:g}kb x {args.length / 1000:g}kb {args.error * 100:g}%-error pairs"
        line = {
            "metric": f"alignments/sec + GCUPS, {label}",
            "value": head["value"], "unit": "alignments/s", "gcups": head["gcups"], "band_gcups": head["band_gcups"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks_seen": ranks_seen, "collective": collective,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.workload}, DEVICE-RESIDENT inputs: {args.pairs} pairs/GPU x {args.length} bp @ {args.error:g} error, "
                                   f"bandwidth {args.bandwidth} %, seeded generator (SURVEY 8d), ASCII already in HBM when the clock "
                                   "starts, results left in HBM (end-to-end rates: e2e)",
                       "pairs_per_gpu": args.pairs, "length": args.length, "error": args.error,
                       "bandwidth": args.bandwidth, "parallelism": B.parallelism},
            "runs_in_flight": head["runs_in_flight"], "single_batch_latency_ms": head["single_batch_latency_ms"],
            "roofline": dict(head["roofline"], hbm_copy_measured_GBs=measured_copy_bandwidth()),
            "valu": head["valu"],
            "score_checksum": head["score_checksum"],
        }
        for k in ("quicked_flow", "cigar_bytes_per_step", "e2e", "cpu_baseline"):
            if k in head:
                line[k] = head[k]
        # The same story as flat top-level numbers (the driver's record keeps scalars, not nested objects).  `value` stays the
        # device-resident rate: the round's contract defines it with the inputs already in HBM and says the PCIe-inclusive
        # rate is never `value`; SURVEY 8(d)'s end-to-end definition (ASCII in -> results on the host) is `e2e_value`.
        line["value_definition"] = "whole-job alignments/s, inputs resident in HBM when the clock starts (= kernel_value)"
        line["kernel_value"] = head["value"]
        if head.get("single_batch_latency_ms"):
            line["single_batch_value"] = args.pairs / (head["single_batch_latency_ms"] * 1e-3)      # ONE batch alone on the chip, synchronous
        e2e = head.get("e2e") or {}
        if isinstance(e2e.get("ascii_hostpacked"), dict) and "value" in e2e["ascii_hostpacked"]:
            line["e2e_value"] = e2e["ascii_hostpacked"]["value"]          # ASCII (the C-ABI's input type) in, results on the host, packing inside the clock
        if isinstance(e2e.get("ascii_pinned"), dict) and "value" in e2e["ascii_pinned"]:
            line["e2e_ascii_link_value"] = e2e["ascii_pinned"]["value"]   # the same with raw ASCII over PCIe: the link's floor
        if isinstance(e2e.get("2bit_pinned"), dict) and "value" in e2e["2bit_pinned"]:
            line["e2e_2bit_value"] = e2e["2bit_pinned"]["value"]
        mc = line["roofline"].get("hbm_copy_measured_GBs")
        if mc:
            line["roofline"]["frac_of_measured_copy"] = line["roofline"]["achieved"] / mc
        if others:
            line["workloads"] = {wl: dict(o, config={"workload": f"{wl}, DEVICE-RESIDENT inputs, " +
                                                                 (f"the same {args.pairs}" if wl in ("quicked", "quicked_score") else str(o.get("pairs_per_gpu"))) +
                                                                 (" pairs/GPU x 100000 bp @ 0.1 error" if wl == "cfg4" else f" pairs/GPU x {args.length} bp @ {args.error:g} error") +
                                                                 ("; scores left in HBM" if wl == "quicked_score" else
                                                                  "; CIGAR strings left in HBM (end-to-end: e2e, strings on the host)")})
                                 for wl, o in others.items()}
        if strong is not None:
            line["strong"] = strong
        if strong_share is not None:
            line["strong_share"] = strong_share
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # the one JSON line is the last thing on stdout: RCCL writes a banner through C stdio, which is block-buffered
        # when stdout is a pipe and would otherwise surface at exit, after the line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:      # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
