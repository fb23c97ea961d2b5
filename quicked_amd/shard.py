"""Multi-GPU bookkeeping of the hot path: pairs are independent work units (the reference's only
parallel mode is an OpenMP loop over pairs, tools/align_benchmark/align_benchmark.c:269-284), so rank r of W simply
owns a contiguous index range of the seeded dataset and the only collective is the final reduction of a tiny vector.
bench.py and tests/test_dist_cpu.py (gloo, world_size 2) both go through these functions.
"""
import os
import subprocess
import sys


def shard_range(total_pairs, rank, world):
    """contiguous range [first, first + count) of rank `rank`: [g N / G, (g + 1) N / G) as SURVEY 8(e) has it"""
    lo = total_pairs * rank // world
    hi = total_pairs * (rank + 1) // world
    return lo, hi - lo


def plan(pairs, rank, world, scaling):
    """-> (first pair index, pairs of this rank, pairs of the whole job).  weak: `pairs` per GPU whatever the world
    size; strong: `pairs` in total, split over the ranks."""
    if scaling == "weak":
        return rank * pairs, pairs, pairs * world
    if scaling == "strong":
        first, count = shard_range(pairs, rank, world)
        return first, count, pairs
    raise ValueError(scaling)


def config5_plan(total_pairs, world, max_per_gpu=250000, node_gpus=8):
    """BASELINE.json configs[4] -- QuickEd score + CIGAR on `total_pairs` (1 M) pairs sharded over the node's GPUs -- for a
    run on `world` GPUs: -> (pairs per GPU, pairs of the whole job).  world > 1: the job split evenly, at most `max_per_gpu`
    per GPU (so 2 or 3 GPUs run a smaller total and say so); world == 1: the shard one GPU of `node_gpus` gets."""
    per_gpu = min(total_pairs // (world if world > 1 else node_gpus), max_per_gpu)
    return per_gpu, per_gpu * world


def reduce_totals(dist, torch, device, pairs, cells, checksum, elapsed, extra_sum=()):
    """SUM of (pairs, cells, checksum, *extra_sum) and MAX of elapsed over the ranks; identity without a process group.
    -> (pairs, cells, checksum, elapsed, [extra sums])"""
    if dist is None:
        return int(pairs), int(cells), int(checksum), float(elapsed), [float(x) for x in extra_sum]
    t = torch.tensor([float(pairs), float(cells), float(checksum)] + [float(x) for x in extra_sum],
                     dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    e = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    v = t.tolist()
    return int(v[0]), int(v[1]), int(v[2]), float(e.item()), v[3:]


def count_ranks(dist, torch, device):
    """all-reduce SUM of 1: how many ranks really took part (1 without a process group)"""
    if dist is None:
        return 1
    t = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(t.item()))


def launch_ranks(n_gpus, script, argv, port=None):
    """`python script --gpus N ...` without a launcher: start N ranks (one per GPU) with torch.distributed.run as a CHILD
    process -- before this process has touched the GPU; a process that has initialised HIP must never exec -- relay the
    child's output and return its exit code."""
    port = port or int(os.environ.get("MASTER_PORT", "29533"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = proc.stdout.splitlines()
    json_lines = [l for l in lines if l.startswith("{")]
    for l in lines:                         # everything but the JSON line first, the line itself last
        if not json_lines or l is not json_lines[-1]:
            print(l)
    if json_lines:
        sys.stdout.flush()
        print(json_lines[-1], flush=True)
    return proc.returncode if proc.returncode != 0 or json_lines else 1
