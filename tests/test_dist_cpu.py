"""The N > 1 path of bench.py on CPU: world_size 2 over gloo through the SAME functions bench.py calls
(quicked_amd/shard.py: plan / shard_range / reduce_totals), and the launcher behaviour of `bench.py --gpus N`."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O
    from quicked_amd import datagen, shard
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {}
    for scaling, pairs in (('weak', 6), ('strong', 9)):       # 9 over 2 ranks: uneven shards
        first, count, total = shard.plan(pairs, rank, world, scaling)       # bench.py's sharding rule
        mine = datagen.generate(count, 400, 0.05, seed=99, first=first)
        scores = [O.oracle_align(p, t, algo=2, only_score=True)[1] for p, t in mine.pairs()]
        tp, tc, ts, te, ext = shard.reduce_totals(dist, torch, None, count, mine.cells(), sum(scores), 1.0 + rank,
                                                  extra_sum=(10.0 * (rank + 1),))
        if rank == 0:
            whole = datagen.generate(total, 400, 0.05, seed=99)
            ref = [O.oracle_align(p, t, algo=2, only_score=True)[1] for p, t in whole.pairs()]
            out[scaling] = {'pairs': tp, 'cells': tc, 'checksum': ts, 'max_elapsed': te, 'extra': ext, 'total': total,
                            'ref_checksum': sum(ref), 'ref_cells': whole.cells()}
    seen = shard.count_ranks(dist, torch, None)                 # bench.py's ranks_seen
    if rank == 0:
        out['ranks_seen'] = seen
        print(json.dumps(out))
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def test_two_rank_sharding_and_reduction(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["ranks_seen"] == 2
    assert r["weak"]["pairs"] == 12 and r["weak"]["total"] == 12 and r["weak"]["max_elapsed"] == 2.0
    assert r["strong"]["pairs"] == 9 and r["strong"]["total"] == 9
    for k in ("weak", "strong"):
        assert r[k]["checksum"] == r[k]["ref_checksum"] and r[k]["cells"] == r[k]["ref_cells"]   # shards == whole
        assert r[k]["extra"] == [30.0]


def test_shard_ranges_partition_the_dataset():
    from quicked_amd import shard
    for total in (0, 1, 7, 100000, 1000003):
        for world in (1, 2, 3, 8):
            covered = 0
            for rank in range(world):
                lo, cnt = shard.shard_range(total, rank, world)
                assert lo == covered and cnt >= 0
                covered += cnt
            assert covered == total
    assert shard.plan(100, 3, 8, "weak") == (300, 100, 800)
    assert shard.plan(100, 7, 8, "strong") == (87, 13, 100)
    assert shard.reduce_totals(None, None, None, 5, 6, 7, 0.5) == (5, 6, 7, 0.5, [])
    assert shard.count_ranks(None, None, None) == 1


def test_launcher_counts_gpus_without_touching_hip(tmp_path, monkeypatch):
    """`bench.py --gpus N` without a launcher decides whether the node has N GPUs from sysfs (KFD topology / DRM render
    nodes) -- the parent of the ranks must not initialise the GPU -- and honours HIP_VISIBLE_DEVICES"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert "torch" not in bench.visible_gpus.__code__.co_names
    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024, 1024)):           # node 0 is the CPU
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\nmem_banks_count 1\n")
    real_glob = bench.glob.glob
    monkeypatch.setattr(bench.glob, "glob", lambda pat: real_glob(str(nodes / "*" / "properties")) if "kfd" in pat else [])
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`bench.py --gpus 2` without a launcher on a node without 2 GPUs: loud failure, no JSON line claiming n_gpus 1"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                  # a real multi-GPU node: the ranks would start; covered on the GPU side
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 2" in out.stderr and "GPU" in out.stderr
    # a launcher whose world size disagrees with --gpus is refused too
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                         env=env2, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_launch_ranks_relays_the_json_line(tmp_path):
    """shard.launch_ranks starts N ranks as a child and prints the child's JSON line last"""
    from quicked_amd import shard
    script = tmp_path / "r.py"
    script.write_text(textwrap.dedent("""
        import os, json, sys
        print("noise from rank", os.environ["RANK"])
        if os.environ["RANK"] == "0":
            print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "argv": sys.argv[1:]}), flush=True)
    """))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from quicked_amd import shard
        sys.exit(shard.launch_ranks(2, {str(script)!r}, ["--gpus", "2"], port=29591))
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert json.loads(last) == {"n_gpus": 2, "argv": ["--gpus", "2"]}


def test_config_5_plan():
    """BASELINE.json configs[4] (1 M pairs over 8 GPUs) per world size: 125 k pairs per GPU at 8 and as the N = 1 shard, an even
    split capped at 250 k per GPU below that, contiguous ranges that tile the job (what bench.py's cfg5 leg runs)"""
    from quicked_amd import shard
    assert shard.config5_plan(1000000, 8) == (125000, 1000000)
    assert shard.config5_plan(1000000, 1) == (125000, 125000)
    assert shard.config5_plan(1000000, 4) == (250000, 1000000)
    assert shard.config5_plan(1000000, 2) == (250000, 500000)
    assert shard.config5_plan(8000, 1) == (1000, 1000) and shard.config5_plan(5000, 2) == (2500, 5000)
    per_gpu, total = shard.config5_plan(1000000, 8)
    ranges = [shard.plan(per_gpu, r, 8, "weak")[:2] for r in range(8)]
    assert ranges[0] == (0, 125000) and all(ranges[r][0] == ranges[r - 1][0] + ranges[r - 1][1] for r in range(1, 8))
    assert ranges[-1][0] + ranges[-1][1] == total
