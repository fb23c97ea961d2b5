#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic read pairs
whose ASCII bytes are already resident in HBM: pack -> BandEd score-only kernel
-> scores in HBM (configs[1] of BASELINE.json: 100 k pairs of 10 kb at 5 %
error, reference default bandwidth 15 %).  `--workload quicked` runs configs[2]
(QuickEd bound-and-align + CIGAR) instead.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (torch.distributed.run), each rank owns its own
shard of pairs (weak scaling, no data-path collective); RCCL is used only for
the final reduction of the timings and checksums.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer-VALU issue roofline of the block step.  gfx950 does not issue every VALU op at the same
# rate: measured with tools/valu_rate.hip (profiles/r01_*_valu_rates.txt), v_and/or/xor/add/lshr/mov and v_bitop3 hold
# a SIMD for ~2.2-2.6 cycles per wave64 op, v_bfe/v_alignbit/v_lshl_add_u64 for ~4.2-4.5.  The cheapest form of the
# walk, four band slots per pass (run64_multi<4>, ISA of k_banded<false>: 2 766 fast + 570 slow instructions per 32
# columns x 4 blocks), needs 9 393 SIMD cycles = 73.4 per block-column (two slots: 78.6, one: 89.5); 1024 SIMDs x
# 2.4 GHz x 64 lanes / 73.4 cycles is what the chip could issue if every slot went through the 4-slot form and nothing
# else ever stalled a SIMD.
ISSUE_CYCLES_PER_BLOCK_COLUMN = 73.4
VALU_PEAK_BLOCK_COLUMNS = 256 * 4 * 2.4e9 * 64 / ISSUE_CYCLES_PER_BLOCK_COLUMN
OPS_PER_BLOCK_COLUMN = 26.1      # VALU instructions per 64-row block per column in that loop (3 336 / 128)

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


def cpu_baseline(batch, params_kw, budget_s=15.0):
    """The compiled reference (oracle/_ref, kind "reference") or the oracle
    restatement (kind "port") on the host cores: oracle/cpu_bench.c, one aligner
    per OpenMP thread over disjoint pair ranges (the reference's own model,
    align_benchmark.c:246-284), on a bounded prefix of the same workload."""
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
    cores = os.cpu_count() or 1

    def run(n, threads):
        scores = np.zeros(n, dtype=np.int32)
        wall = lib.cpu_bench_run(ref_so, n, batch.pattern_pool.ctypes.data, batch.pattern_off.ctypes.data,
                                 batch.pattern_len.ctypes.data, batch.text_pool.ctypes.data, batch.text_off.ctypes.data,
                                 batch.text_len.ctypes.data, params_kw["algo"], 1 if params_kw.get("only_score") else 0,
                                 params_kw.get("bandwidth", 15), threads, scores.ctypes.data)
        assert wall > 0, "cpu_bench_run failed"
        return wall, scores

    run(min(len(batch), cores), cores)                   # warm the library, the arenas and the page cache
    n1 = min(len(batch), 64)
    w1, _ = run(n1, 1)                                   # single-thread calibration
    per = w1 / n1
    n = int(min(len(batch), max(cores * 8, budget_s / per * cores)))
    wall, scores = run(n, cores)
    return {"value": n / wall, "unit": "alignments/s", "cores": cores, "cpu_model": cpu_model(), "kind": kind,
            "sample": f"first {n} pairs of the same workload, {cores} OpenMP threads, one aligner per thread",
            "single_thread_value": 1.0 / per}, scores.astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=100000, help="pairs per GPU per step")
    ap.add_argument("--length", type=int, default=10000)
    ap.add_argument("--error", type=float, default=0.05)
    ap.add_argument("--bandwidth", type=int, default=15)
    ap.add_argument("--workload", choices=["banded_score", "quicked"], default="banded_score")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="profiling aid: no overlap between consecutive runs, so per-kernel durations are those of a kernel alone")
    ap.add_argument("--seed", type=int, default=0x51CED)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1 or os.environ.get("QE_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm

    from quicked_amd import capi, datagen
    capi.lib().quicked_set_device(local_rank)

    # each rank owns pairs [rank * pairs, (rank + 1) * pairs) of the seeded dataset: independent work units
    batch = datagen.generate(args.pairs, args.length, args.error, seed=args.seed, first=rank * args.pairs)
    cells = batch.cells()
    if args.workload == "banded_score":
        kw = dict(algo=capi.BANDED, only_score=True, bandwidth=args.bandwidth)
    else:
        kw = dict(algo=capi.QUICKED, only_score=False, bandwidth=args.bandwidth)
    params = capi.make_params(**kw)
    rb = capi.ResidentBatch(batch)           # H2D happens here, outside the timed region

    for _ in range(max(args.warmup, 0)):
        st = rb.run(params, sync=True)
        assert st >= 0, f"quicked_batch_run failed: {capi.lib().quicked_status_msg(st).decode().strip()}"
    if args.warmup <= 0:
        st = 0
    rb.kernel_time()                         # drop the warm-up launches

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        rb.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = rb.run(params, sync=args.sync_each_step)     # results stay resident in HBM; the driver still syncs where a stage needs host decisions
        assert st >= 0, f"quicked_batch_run failed: {capi.lib().quicked_status_msg(st).decode().strip()}"
    rb.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = rb.kernel_time()

    # one synchronous run to fetch results + work counters for the report
    st = rb.run(params, sync=True)
    assert st >= 0, f"quicked_batch_run failed: {capi.lib().quicked_status_msg(st).decode().strip()}"
    scores, status = rb.scores()
    assert (status >= 0).all(), "some pairs failed" 
    counters = rb.counters()
    rb.kernel_time()
    checksum = int(scores.astype(np.int64).sum())

    tot_pairs, tot_cells, max_elapsed, tot_checksum = args.pairs, cells, elapsed, checksum
    if dist is not None:
        t = torch.tensor([float(args.pairs), float(cells), float(checksum)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        e = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        tot_pairs, tot_cells, tot_checksum, max_elapsed = int(t[0].item()), int(t[1].item()), int(t[2].item()), e.item()

    if rank == 0:
        value = tot_pairs * args.steps / max_elapsed
        per_launch_bytes = float((batch.pattern_len.astype(np.int64) + batch.text_len.astype(np.int64) + 4).sum())
        kern_s = (kern_ms / 1e3 / kern_n) if kern_n else float("nan")
        if args.workload == "banded_score":
            kernel, alg_bytes, work_blocks = "k_banded<false> (BandEd score-only)", per_launch_bytes, int(counters[0])
        else:
            # ASCII in + 16 B per stored block-column + 16 B per traceback step + ops out.  NB the kernels
            # store a 16-byte checkpoint every 8th column and recompute the rest: the HBM traffic they
            # generate (roofline.traffic) is BELOW this figure, so frac can exceed the naive bound
            alg_bytes = per_launch_bytes + 16.0 * counters[1] + 16.0 * counters[3] + float(counters[4])
            kernel, work_blocks = "k_banded<true> (BandEd fill)", int(counters[1])
        traffic = None
        try:      # HBM bytes per launch from the committed PMC passes of this same command (profiles/)
            with open(os.path.join(ROOT, "profiles", "r01_i_pmc_traffic.json")) as f:
                pm = json.load(f)["banded_score" if args.workload == "banded_score" else "quicked"]
            key = "k_banded<false>" if args.workload == "banded_score" else "k_banded<true>"
            if args.pairs == 100000 and args.length == 10000:
                traffic = pm[key]["hbm_bytes"]
        except Exception:
            traffic = None
        achieved = alg_bytes / kern_s / 1e9
        valu_rate = work_blocks / kern_s
        # consecutive runs overlap on two streams (the next run's kernel takes the SIMD slots this one leaves empty), so
        # a launch lasts longer than its share of the wall clock: the aggregate figures divide the same per-launch
        # work by the step time instead of the launch duration
        step_s = max_elapsed / args.steps
        agg_valu_rate = work_blocks / step_s
        line = {
            "metric": "alignments/sec + GCUPS, 10kb x 10kb 5%-error pairs",
            "value": value, "unit": "alignments/s", "gcups": tot_cells * args.steps / max_elapsed / 1e9,
            # cells actually computed (SURVEY 8d "band GCUPS"): 64 rows x block-advances of the dominant kernel, this rank x world
            "band_gcups": 64.0 * work_blocks * world * args.steps / max_elapsed / 1e9,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": max_elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.pairs} pairs/GPU x {args.length} bp @ {args.error:g} error, "
                                   f"bandwidth {args.bandwidth} %, seeded generator (SURVEY 8d), ASCII resident in HBM",
                       "pairs_per_gpu": args.pairs, "length": args.length, "error": args.error,
                       "bandwidth": args.bandwidth, "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_s * 1e3,
                         "aggregate_achieved": alg_bytes / step_s / 1e9,
                         "hbm_copy_measured_GBs": measured_copy_bandwidth(),
                         "note": "score-only BandEd is integer-VALU-bound, not HBM-bound (SURVEY 8d); see valu"},
            "valu": {"bound": "integer VALU issue", "achieved": valu_rate, "peak": VALU_PEAK_BLOCK_COLUMNS,
                     "unit": "block-columns/s", "frac": valu_rate / VALU_PEAK_BLOCK_COLUMNS,
                     "aggregate_achieved": agg_valu_rate, "aggregate_frac": agg_valu_rate / VALU_PEAK_BLOCK_COLUMNS,
                     "block_columns_per_launch": work_blocks, "issue_cycles_per_block_column": ISSUE_CYCLES_PER_BLOCK_COLUMN,
                     "instructions_per_block_column": OPS_PER_BLOCK_COLUMN,
                     "note": "achieved uses the launch duration (launches of consecutive runs overlap); aggregate uses the step time"},
            "score_checksum": tot_checksum,
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU reference is timed on rank 0 at N = 1 only
            base, ref_scores = cpu_baseline(batch, {k: v for k, v in kw.items()})
            n = len(ref_scores)
            base["gpu_scores_identical_on_sample"] = bool((scores[:n].astype(np.int64) == ref_scores).all())
            line["cpu_baseline"] = base
    rb.close()
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
