"""The N > 1 bookkeeping of bench.py on CPU: world_size 2, gloo.  Shards are
disjoint, reproducible per rank, and the final reduction is SUM / MAX."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O
    from quicked_amd import datagen
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    per = 6
    mine = datagen.generate(per, 400, 0.05, seed=99, first=rank * per)      # bench.py's sharding rule
    scores = [O.oracle_align(p, t, algo=2, only_score=True)[1] for p, t in mine.pairs()]
    t = torch.tensor([float(per), float(mine.cells()), float(sum(scores))], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    e = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    if rank == 0:
        whole = datagen.generate(per * world, 400, 0.05, seed=99)
        ref = [O.oracle_align(p, t, algo=2, only_score=True)[1] for p, t in whole.pairs()]
        print(json.dumps({'pairs': t[0].item(), 'cells': t[1].item(), 'checksum': t[2].item(), 'max_elapsed': e.item(),
                          'ref_checksum': float(sum(ref)), 'ref_cells': float(whole.cells())}))
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
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["pairs"] == 12 and r["max_elapsed"] == 2.0
    assert r["checksum"] == r["ref_checksum"] and r["cells"] == r["ref_cells"]   # shard-of-2 == whole
