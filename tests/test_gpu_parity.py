"""Parity tests proper: the HIP path, called through the C-ABI, against the
oracle on the same seeded inputs and against the golden vectors captured from
the compiled reference.  Bit-exact: scores, statuses and RLE CIGAR strings."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O
from quicked_amd import capi, datagen

pytestmark = pytest.mark.gpu


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest() if s is not None else None


_ORACLE_MEMO = {}


def oracle_cached(p, t, trace=False, **kw):
    """O.oracle_align, memoised per (pair, parameters): the forced-form tests run the same inputs once per kernel form"""
    kw = dict(kw, trace=trace)
    key = (bytes(p), bytes(t), tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items())))
    r = _ORACLE_MEMO.get(key)
    if r is None:
        r = _ORACLE_MEMO[key] = O.oracle_align(p, t, **kw)
    return r


_EXACT_MEMO = {}


def exact_cached(p, t):
    """the oracle's full-height DP distance, memoised (it is the slow part of the in-domain checks)"""
    key = (bytes(p), bytes(t))
    if key not in _EXACT_MEMO:
        _EXACT_MEMO[key] = O.oracle().qo_exact_distance(p, len(p), t, len(t))
    return _EXACT_MEMO[key]


def oracle_many(pairs, threads=None, **kw):
    """oracle_cached over many pairs on the host's cores (ctypes releases the GIL inside the C oracle): the strided
    samples of the full-size tests are >= 1 000 pairs of 10 kb"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, min(32, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda pt: oracle_cached(pt[0], pt[1], **kw), pairs))


def gpu_batch(batch, **kw):
    """-> (scores, statuses, cigars or None, counters) through quicked_batch_*"""
    rb = capi.ResidentBatch(batch)
    p = capi.make_params(**kw)
    st = rb.run(p, sync=True)
    assert st >= 0 or st == capi.QUICKED_EMPTY_SEQUENCE, st
    scores, status = rb.scores()
    cig = None if kw.get("only_score") else rb.cigars()
    cnt = rb.counters()
    rb.close()
    return scores, status, cig, cnt


def test_kats_single_pair_abi(golden):
    """every KAT of the reference's tests/examples through quicked_new/align/free"""
    import ctypes as C
    lib = capi.lib()
    for k in golden["kats"]:
        p = capi.make_params(**k["params"])
        a = capi.Aligner()
        assert lib.quicked_new(C.byref(a), C.byref(p)) == capi.QUICKED_WIP
        pat, txt = k["pattern"].encode(), k["text"].encode()
        st = lib.quicked_align(C.byref(a), pat, len(pat), txt, len(txt))
        assert st == k["status"], k
        if st >= 0:
            assert a.score == k["score"], k
            assert (a.cigar.decode() if a.cigar else None) == k["cigar"], k
        lib.quicked_free(C.byref(a))


def test_binding_mirror_example():
    """examples/bindings/basic.py of the reference: ACGT vs ACTT -> 1, 2M1X1M"""
    al = capi.QuickedAligner()
    al.align("ACGT", "ACTT")
    assert al.getScore() == 1 and al.getCigar() == "2M1X1M"
    al.setAlgorithm(capi.BANDED)
    al.setOnlyScore(True)
    al.align("ACGT", "ACTT")
    assert al.getScore() == 1
    with pytest.raises(capi.QuickedException):
        al.align("", "")


DATASETS = ["cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len50", "len63", "len64", "len65", "len128",
            "len130", "len1024", "err35_2kb"]


@pytest.mark.parametrize("name", DATASETS)
def test_golden_datasets(golden, name):
    entry = golden["datasets"][name]
    batch = datagen.generate(**entry["gen"])
    pairs = list(batch.pairs())
    for label, run in entry["runs"].items():
        scores, status, cig, _ = gpu_batch(batch, **run["params"])
        assert status.tolist() == run["status"], (name, label)
        assert scores.tolist() == run["score"], (name, label)
        if "cigar_sha256" in run:
            assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
            for (p, t), c in zip(pairs, cig):
                assert O.cigar_is_valid(p, t, c)


RUNS = [
    dict(algo=2, only_score=True, bandwidth=1), dict(algo=2, only_score=True, bandwidth=4),
    dict(algo=2, only_score=True, bandwidth=15), dict(algo=2, bandwidth=15), dict(algo=3, bandwidth=15),
    dict(algo=1, only_score=True), dict(algo=1), dict(algo=1, only_score=True, window_size=2),
    dict(algo=1, window_size=2), dict(algo=1, only_score=True, window_size=2, force_scalar=True),
    dict(algo=1, window_size=2, force_scalar=True), dict(algo=1, only_score=True, window_size=4, overlap_size=2),
    dict(algo=0), dict(algo=0, force_scalar=True), dict(algo=0, only_score=True),
]


@pytest.mark.parametrize("gen", [
    dict(count=130, length=1000, error=0.05, seed=301), dict(count=70, length=10000, error=0.05, seed=302),
    dict(count=100, length=200, error=0.15, seed=303), dict(count=100, length=70, error=0.2, seed=304),
    dict(count=40, length=3000, error=0.3, seed=305), dict(count=70, length=1, error=0, seed=306),
    dict(count=70, length=5, error=2, seed=307),
])
def test_oracle_parity_random(gen):
    batch = datagen.generate(**gen)
    pairs = list(batch.pairs())
    for kw in RUNS:
        scores, status, cig, _ = gpu_batch(batch, **kw)
        for i, (p, t) in enumerate(pairs):
            st, sc, cg = oracle_cached(p, t, **kw)
            assert status[i] == st, (gen, kw, i)
            assert scores[i] == sc, (gen, kw, i)
            if cig is not None:
                assert cig[i] == cg, (gen, kw, i)


def mixed_batch():
    """ragged lengths, empty sequences, N / lower-case / IUPAC symbols in one batch"""
    rng = np.random.default_rng(5)
    base = datagen.generate(count=48, length=700, error=0.08, seed=41)
    pairs = []
    for i, (p, t) in enumerate(base.pairs()):
        p, t = bytearray(p[: 20 + 14 * i]), bytearray(t[: 25 + 13 * i])
        if i % 4 == 1:
            for k in rng.integers(0, len(p), 3): p[k] = ord("N")
            for k in rng.integers(0, len(t), 3): t[k] = ord("N")
        if i % 4 == 2:
            p = bytearray(bytes(p).lower())
        if i % 4 == 3:
            for k in rng.integers(0, len(t), 2): t[k] = ord("R")
            for k in rng.integers(0, len(p), 2): p[k] = ord("n")
        pairs.append((bytes(p), bytes(t)))
    pairs[7] = (b"", pairs[7][1])
    pairs[9] = (pairs[9][0], b"")
    pairs[11] = (b"", b"")
    return pairs


@pytest.mark.parametrize("kw", [dict(algo=2, only_score=True), dict(algo=2), dict(algo=1, only_score=True),
                                dict(algo=1), dict(algo=1, window_size=2), dict(algo=0), dict(algo=3)])
def test_ragged_empty_and_non_acgt(kw):
    pairs = mixed_batch()
    al = capi.QuickedAligner()
    for k, v in kw.items():
        setattr(al._params, k, v)
    st, out = al.alignBatch(pairs)
    for i, (p, t) in enumerate(pairs):
        est, esc, ecg = oracle_cached(p, t, **kw)
        assert out[i][0] == est, (kw, i)
        if est >= 0:
            assert out[i][1] == esc, (kw, i, len(p), len(t))
            assert out[i][2] == ecg, (kw, i)


def test_counters_match_oracle_work():
    """block-advances / window steps / traceback steps are the SURVEY 8(d) work units"""
    batch = datagen.generate(count=64, length=2000, error=0.05, seed=77)
    pairs = list(batch.pairs())
    _, _, _, cnt = gpu_batch(batch, algo=2, only_score=True, bandwidth=15)
    exp = sum(oracle_cached(p, t, trace=True, algo=2, only_score=True, bandwidth=15)[3]["score_block_advances"] for p, t in pairs)
    assert cnt[0] == exp
    _, _, _, cnt = gpu_batch(batch, algo=0)
    tr = [oracle_cached(p, t, trace=True, algo=0)[3] for p, t in pairs]
    assert cnt[1] == sum(x["fill_block_advances"] for x in tr)
    assert cnt[2] == sum(x["window_block_steps"] for x in tr)
    assert cnt[3] == sum(x["traceback_steps"] for x in tr)


@pytest.mark.parametrize("name", ["cfg4_100kb_10pct", "cfg4_indel_100kb"])
def test_hirschberg_split_100kb_golden(golden, name):
    """configs[3] shape: 100 kb / 10 % pairs go through two Hirschberg split levels (bpm_hirschberg.c:63-243); six
    ordinary pairs and two with 3 x 2 kb indels (stages 2 / 3 before the split), bytes of the compiled reference"""
    entry = golden["datasets"][name]
    batch = datagen.generate(**entry["gen"])
    for label in ("quicked", "hirschberg_bw15", "banded_so_bw15", "windowed_so_2_1_sse", "windowed_so_2_1_scalar"):
        run = entry["runs"][label]
        scores, status, cig, cnt = gpu_batch(batch, **run["params"])
        assert status.tolist() == run["status"], label
        assert scores.tolist() == run["score"], label
        if "cigar_sha256" in run:
            assert [sha(c) for c in cig] == run["cigar_sha256"], label


def test_ont_real_data_fixture_on_the_gpu(golden):
    """the reference's one real-data test (tests/CMakeLists.txt:32, tests/test_data/ONT.MiniION.1.seq: 508 596 x 505 792
    bases, ~7.4 % error, all QuickEd stages + Hirschberg) through quicked_new / quicked_align / quicked_free: status, score
    39 743 (edlib-confirmed) and the CIGAR bytes of the compiled reference (length + SHA-256 in golden.json)"""
    import ctypes as C
    import os
    g = golden["ont_miniion_1"]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ont_miniion_1.seq"), "rb") as f:
        l1, l2 = f.read().split(b"\n")[:2]
    pat, txt = (l1[1:], l2[1:]) if l1[:1] == b">" else (l2[1:], l1[1:])
    assert (len(pat), len(txt)) == (g["plen"], g["tlen"])
    lib = capi.lib()
    p = capi.make_params(algo=0)
    a = capi.Aligner()
    assert lib.quicked_new(C.byref(a), C.byref(p)) == capi.QUICKED_WIP
    st = lib.quicked_align(C.byref(a), pat, len(pat), txt, len(txt))
    cg = a.cigar.decode() if a.cigar else None
    score = a.score
    lib.quicked_free(C.byref(a))
    assert (st, score) == (g["status"], g["score"]) and score == 39743
    assert cg is not None and len(cg) == g["cigar_len"] and sha(cg) == g["cigar_sha256"]


def test_reference_crash_fuzz_at_its_own_scale():
    """tests/CMakeLists.txt:15-21 + tests/random_test.sh:47-59: 10 000 pairs of 1 kb and 1 000 pairs of 10 kb with TEN edits
    each (generate_dataset's -e 10 is a count) through quicked_new / quicked_align / quicked_free PER PAIR, default
    parameters (QuickEd + CIGAR).  The reference's criterion is 'no crash'; here every status and score is checked
    (score <= 10: ten edits; == the oracle's on a stride) and every CIGAR's edit count is its score."""
    import ctypes as C
    import re
    lib = capi.lib()
    p = capi.make_params(algo=0)
    for count, length, stride in ((10000, 1000, 97), (1000, 10000, 53)):
        batch = datagen.generate(count, length, 10, seed=4242 + length)       # error >= 1: a number of edits
        for i, (pat, txt) in enumerate(batch.pairs()):
            a = capi.Aligner()
            assert lib.quicked_new(C.byref(a), C.byref(p)) == capi.QUICKED_WIP
            st = lib.quicked_align(C.byref(a), pat, len(pat), txt, len(txt))
            score, cg = a.score, (a.cigar.decode() if a.cigar else None)
            lib.quicked_free(C.byref(a))
            assert st == capi.QUICKED_WIP and 0 <= score <= 10 and cg is not None, (length, i, st, score)
            if i % stride == 0:
                assert (st, score, cg) == oracle_cached(pat, txt, algo=0), (length, i)
                assert sum(int(n) for n, op in re.findall(r"(\d+)([MXID])", cg) if op != "M") == score


def test_configs_2_and_3_at_full_size():
    """BASELINE configs[1] / [2] at their own size: 100 k pairs of 10 kb at 5 %.  BandEd score-only (bandwidth 15) and
    QuickEd + CIGAR on the same pairs: the two exact distances agree on all 100 k pairs, every CIGAR passes the device-side
    validator (cigar_check_alignment, cigar.c:363-434) and carries its score as edit count, a strided sample equals the
    oracle's bytes -- and the compiled reference's where oracle/_ref exists (the build container)."""
    N = 100000
    whole = datagen.generate(N, 10000, 0.05, seed=0x51CED)
    rb = capi.ResidentBatch(whole)
    try:
        assert rb.run(capi.make_params(algo=2, only_score=True, bandwidth=15), sync=True) == capi.QUICKED_WIP
        s_b, st_b = rb.scores()
        s_b = s_b.copy()
        assert (st_b == capi.QUICKED_WIP).all()
        rb.configure(check=True)
        assert rb.run(capi.make_params(algo=0), sync=True) == capi.QUICKED_WIP
        s_q, st_q = rb.scores()
        assert (st_q == capi.QUICKED_WIP).all() and (s_q == s_b).all()
        assert 440 < int(s_q.min()) and int(s_q.max()) < 520                 # 460-493 on this generator
        assert bool(rb.check_results().all())
        cig = rb.cigars()
        s_q = s_q.copy()
        # QuickEd with only_score (one score-only pass over the fill's cells instead of the alignment): the same 100 k distances,
        # a run the caller waits for and a queued one (the pass inside the fast flow, its cutoffs from the device)
        rb.configure(check=False)
        p_qs = capi.make_params(algo=0, only_score=True)
        assert rb.run(p_qs, sync=True) == capi.QUICKED_WIP
        s_s, st_s = rb.scores()
        assert (st_s == capi.QUICKED_WIP).all() and (s_s == s_q).all() and rb.counters()[3] == 0
        assert rb.run(p_qs, sync=False) >= 0
        rb.fetch()
        s_s, st_s = rb.scores()
        assert (st_s == capi.QUICKED_WIP).all() and (s_s == s_q).all() and rb.counters()[3] == 0
    finally:
        rb.close()
        capi.pool_trim()                  # 100 k-pair QuickEd pools (five sets): not this test's to leave to the ones after it
    idx = list(range(0, N, 97))                                              # 1 031 pairs: status, score and CIGAR bytes
    sample = [(whole.pattern(i), whole.text(i)) for i in idx]
    want_q = oracle_many(sample, algo=0)
    want_b = oracle_many(sample, algo=2, only_score=True, bandwidth=15)
    for i, wq, wb in zip(idx, want_q, want_b):
        assert (capi.QUICKED_WIP, int(s_q[i]), sha(cig[i])) == (wq[0], wq[1], sha(wq[2])), i
        assert wb[1] == s_b[i], i
    if O.have_ref():
        for i, (p, t), wq in list(zip(idx, sample, want_q))[::20]:
            assert O.ref_align(p, t, algo=0) == wq, i


def test_config_5_shard_at_full_size():
    """BASELINE configs[4] (1 M pairs of 10 kb at 5 %, QuickEd score + CIGAR over 8 GPUs) at the size ONE GPU of the eight
    sees: shard 3 of 8 = pairs [375 000, 500 000) of the seeded dataset (quicked_amd/shard.py; align_benchmark.c:246-284 hands
    its threads disjoint pair ranges the same way).  Every CIGAR passes the device-side validator and carries its score as
    edit count; status, score and CIGAR bytes equal the oracle's on every 250th pair; and a shard generated on its own is
    the same data as that range of a larger one (what makes the ranks independent)."""
    from quicked_amd import shard
    first, count, total = shard.plan(1000000, 3, 8, "strong")
    assert (first, count, total) == (375000, 125000, 1000000)
    part = datagen.generate(count, 10000, 0.05, seed=0x51CED, first=first)
    probe = datagen.generate(3, 10000, 0.05, seed=0x51CED, first=first + 1000)
    assert [part.pattern(1000 + k) for k in range(3)] == [probe.pattern(k) for k in range(3)]
    rb = capi.ResidentBatch(part)
    try:
        rb.configure(check=True)
        assert rb.run(capi.make_params(algo=0), sync=True) == capi.QUICKED_WIP
        s_q, st_q = rb.scores()
        assert (st_q == capi.QUICKED_WIP).all()
        assert 430 < int(s_q.min()) and int(s_q.max()) < 530
        assert bool(rb.check_results().all())                    # validity + edit count == score, on the device, all 125 k
        cig = rb.cigars()
        # the stream form the bench times (queued runs, nothing fetched in between) leaves the same results
        for _ in range(3):
            assert rb.run(capi.make_params(algo=0), sync=False) >= 0
        assert rb.fetch() >= 0
        s2, st2 = rb.scores()
        assert (s2 == s_q).all() and (st2 == st_q).all() and rb.deferred_pairs() == 0
    finally:
        rb.close()
        capi.pool_trim()
    idx = list(range(0, count, 250))
    want = oracle_many([(part.pattern(i), part.text(i)) for i in idx], algo=0)
    for i, w in zip(idx, want):
        assert (capi.QUICKED_WIP, int(s_q[i]), sha(cig[i])) == (w[0], w[1], sha(w[2])), i


def test_config_4_at_full_size():
    """BASELINE configs[3] at its own size: 10 k pairs of 100 kb at 10 %, QuickEd + Hirschberg CIGAR (every pair splits,
    bpm_hirschberg.c:63-65).  The device-side validator passes every CIGAR (valid + edit count == score), the score
    equals BandEd score-only's at bandwidth 15 on every pair (two different kernels, the same exact distance), and status,
    score and CIGAR bytes equal the oracle's on every 400th pair."""
    N = 10000
    whole = datagen.generate(N, 100000, 0.10, seed=0x51CED)
    rb = capi.ResidentBatch(whole)
    try:
        rb.configure(check=True)
        assert rb.run(capi.make_params(algo=0), sync=True) == capi.QUICKED_WIP
        s_q, st_q = rb.scores()
        s_q = s_q.copy()
        assert (st_q == capi.QUICKED_WIP).all()
        assert bool(rb.check_results().all())
        cig = rb.cigars()
        capi.pool_trim()
        assert rb.run(capi.make_params(algo=2, only_score=True, bandwidth=15), sync=True) == capi.QUICKED_WIP
        s_b, st_b = rb.scores()
        assert (st_b == capi.QUICKED_WIP).all() and (s_b == s_q).all()
        assert 8500 < int(s_q.min()) and int(s_q.max()) < 9900            # ~9.2 k on this generator
    finally:
        rb.close()
        capi.pool_trim()
    idx = list(range(0, N, 400))
    want = oracle_many([(whole.pattern(i), whole.text(i)) for i in idx], algo=0)
    for i, w in zip(idx, want):
        assert (capi.QUICKED_WIP, int(s_q[i]), sha(cig[i])) == (w[0], w[1], sha(w[2])), i


def test_hirschberg_forced_deep_splits(monkeypatch):
    """QE_SPLIT_BYTES forces many levels on small inputs; the oracle runs with the same threshold"""
    import ctypes as C
    monkeypatch.setenv("QE_SPLIT_BYTES", str(1 << 15))
    lib = O.oracle()
    for gen in (dict(count=70, length=3000, error=0.08, seed=401), dict(count=40, length=1500, error=0.2, seed=402)):
        batch = datagen.generate(**gen)
        pairs = list(batch.pairs())
        scores, status, cig, cnt = gpu_batch(batch, algo=0)
        for i, (p, t) in enumerate(pairs):
            st, sc, cg, tr = oracle_cached(p, t, trace=True, algo=0)      # reference threshold: bound + score
            ops = C.create_string_buffer(len(p) + len(t) + 1)
            n = C.c_int64()
            lib.qo_hirschberg(p, len(p), t, len(t), tr["bound"], 1 << 15, ops, C.byref(n), None)
            buf = C.create_string_buffer(2 * n.value + 16)
            lib.qo_cigar_rle(ops, n.value, buf)
            assert scores[i] == sc == lib.qo_exact_distance(p, len(p), t, len(t)), (gen, i)
            assert cig[i] == buf.value.decode(), (gen, i)
            assert O.cigar_is_valid(p, t, cig[i])


def test_align_benchmark_harness(tmp_path):
    """tools/align_benchmark: the reference CLI's input / output formats and --check score"""
    import subprocess
    from quicked_amd import build
    exe = build.build_harness()
    batch = datagen.generate(count=100, length=1500, error=0.06, seed=77)
    pairs = list(batch.pairs())
    seq = tmp_path / "in.seq"
    with open(seq, "wb") as f:
        for p, t in pairs:
            f.write(b">" + p + b"\n<" + t + b"\n")
    for algo, kw in (("quicked", dict(algo=0)), ("edit-banded", dict(algo=2)), ("edit-windowed", dict(algo=1)),
                     ("edit-banded-hirschberg", dict(algo=3))):
        out = tmp_path / f"{algo}.out"
        r = subprocess.run([exe, "-a", algo, "-i", str(seq), "-o", str(out), "-c", "score", "--batch-size", "64"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = out.read_text().splitlines()
        assert len(lines) == len(pairs)
        for (p, t), line in zip(pairs, lines):
            st, sc, cg = oracle_cached(p, t, **kw)
            assert line == f"{sc}\t{cg}", algo
        assert "INACCURATE SCORE" not in r.stderr or algo == "edit-windowed"      # WindowEd is a bound, not exact
        assert "Alignments.Correct     100/100" in r.stderr
    # --verbose: the stage-timer report of align_benchmark.c:116-128 (two batches of 64 / 36 pairs -> two laps)
    # the long option takes the level, -v is level 1 (align_benchmark_params.c:126-127, 241-250)
    for flags in (["--verbose", "2"], ["-v"], ["--verbose=1"]):
        r = subprocess.run([exe, "-a", "quicked", "-i", str(seq), "--batch-size", "64"] + flags, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        for name in ("Windowed Small", "Windowed Large", "Banded", "Align"):
            assert f"=> Time.{name}" in r.stderr
        assert [l for l in r.stderr.splitlines() if "Time.Windowed Small" in l][0].rstrip().endswith("(2 calls)")
        assert [l for l in r.stderr.splitlines() if "Time.Align " in l][0].rstrip().endswith("(2 calls)")
        assert f"Total.reads              {len(pairs)}" in r.stderr
    r = subprocess.run([exe, "-a", "quicked", "-i", str(seq), "--verbose", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must be in {0,1,2,3,4}" in r.stderr


def test_align_benchmark_totals_through_rccl_on_one_device(tmp_path):
    """tools/align_benchmark --force-rccl: the multi-device totals path (dlopen librccl, ncclCommInitAll, a grouped
    ncclAllReduce on device buffers, ncclCommDestroy: align_benchmark.c:246-284's reduction mapped to devices) executed with the
    ONE device this box has -- a one-rank communicator -- and checked by the tool itself against the host-summed totals; the
    printed figures must equal the plain run's."""
    import re
    import subprocess
    from quicked_amd import build
    exe = build.build_harness()
    batch = datagen.generate(count=300, length=1200, error=0.06, seed=81)
    seq = tmp_path / "in.seq"
    with open(seq, "wb") as f:
        for p, t in batch.pairs():
            f.write(b">" + p + b"\n<" + t + b"\n")
    got = {}
    for tag, extra in (("plain", []), ("rccl", ["--force-rccl"]), ("rccl_t3", ["--force-rccl", "-t", "3", "--devices", "1"])):
        r = subprocess.run([exe, "-a", "quicked", "-i", str(seq), "-c", "correct", "--batch-size", "128"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        got[tag] = (re.search(r"Total\.reads\s+(\d+)", r.stderr).group(1), re.search(r"Score\.sum (-?\d+)", r.stderr).group(1),
                    re.search(r"Alignments\.Correct\s+(\d+/\d+)", r.stderr).group(1))
        assert ("Totals.by ncclAllReduce" in r.stderr) == (tag != "plain"), r.stderr
    assert got["plain"] == got["rccl"] == got["rccl_t3"] and got["plain"][0] == "300", got


def test_align_benchmark_worker_threads_write_the_same_file(tmp_path):
    """tools/align_benchmark -t N: the reference's parallel mode (N aligners over disjoint pairs of every block,
    align_benchmark.c:246-284) as N host threads with an aligner each, worker g on device g % devices, reading / aligning /
    writing overlapped.  On this one-GPU box both workers share device 0 (the totals are then summed on the host; with
    several devices they go through ncclAllReduce): the output file must be byte-identical to -t 1's, whatever the job size,
    and the totals the same."""
    import re
    import subprocess
    from quicked_amd import build
    exe = build.build_harness()
    batch = datagen.generate(count=600, length=1200, error=0.06, seed=78)
    hard = datagen.generate(count=40, length=1200, error=0.06, seed=79, indels_num=2, indels_len=150)
    pairs = list(batch.pairs()) + list(hard.pairs()) + [(b"ACGT", b""), (b"ACGTACGT", b"ACGAACGT")]
    seq = tmp_path / "in.seq"
    with open(seq, "wb") as f:
        for p, t in pairs:
            f.write(b">" + p + b"\n<" + t + b"\n")
    outs, sums = {}, {}
    for tag, extra in (("t1", []), ("t2", ["-t", "2"]), ("t3", ["-t", "3", "--batch-size", "100"]), ("t4", ["-t", "4", "--devices", "1"]),
                       ("t8", ["-t", "8", "--devices", "1", "--batch-size", "48"])):      # config 5's worker count on the one device
        out = tmp_path / f"{tag}.out"
        r = subprocess.run([exe, "-a", "quicked", "-i", str(seq), "-o", str(out), "-c", "correct", "--batch-size", "128"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs[tag] = out.read_bytes()
        sums[tag] = re.search(r"Score\.sum (-?\d+)", r.stderr).group(1)
        assert f"Total.reads              {len(pairs)}" in r.stderr, r.stderr
        assert f"Alignments.Correct     {len(pairs) - 1}/{len(pairs) - 1}" in r.stderr, r.stderr      # the empty-text pair has no alignment
    lines = outs["t1"].decode().splitlines()
    assert len(lines) == len(pairs)
    for i in list(range(0, len(pairs), 37)) + [len(pairs) - 1]:
        st, sc, cg = oracle_cached(*pairs[i], algo=0)
        assert lines[i] == f"{sc}\t{cg}", i
    assert lines[len(pairs) - 2] == "-\t-"
    for tag in ("t2", "t3", "t4", "t8"):
        assert outs[tag] == outs["t1"], tag
        assert sums[tag] == sums["t1"], (tag, sums)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_randomised_shapes_and_params(seed, monkeypatch):
    """fuzz: ragged lengths 1..4000, mixed error rates, random algorithm parameters -- every score, status
    and CIGAR equal to the oracle's"""
    rng = np.random.default_rng(seed)
    pairs = []
    for i in range(96):
        L = int(rng.choice([1, 2, 7, 63, 64, 65, 127, 128, 129, 500, 1000, 2500, 4000]))
        e = float(rng.choice([0.0, 0.02, 0.1, 0.3]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=seed * 1000 + i)
        p, t = next(b.pairs())
        if rng.random() < 0.15:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 3))))]      # length mismatch
        if rng.random() < 0.1:
            p = p.replace(b"A", b"N", 2)
        pairs.append((p, t))
    for _ in range(6):
        algo = int(rng.integers(0, 4))
        kw = dict(algo=algo, bandwidth=int(rng.choice([1, 3, 10, 15, 40])), only_score=bool(rng.integers(0, 2)),
                  force_scalar=bool(rng.integers(0, 2)))
        if algo in (0, 1):
            W = int(rng.choice([2, 3, 5, 9]))
            kw.update(window_size=W, overlap_size=int(rng.integers(0, W)))
        if algo == 0:
            kw.update(hew_threshold=(int(rng.choice([10, 40])),) * 2, hew_percentage=(int(rng.choice([1, 15])),) * 2)
        al = capi.QuickedAligner()
        for k, v in kw.items():
            if k in ("hew_threshold", "hew_percentage"):
                getattr(al._params, k)[0], getattr(al._params, k)[1] = v
            else:
                setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            est, esc, ecg = oracle_cached(p, t, **kw)
            # CIGAR-producing BandEd / Hirschberg below the true distance is outside the parity domain
            in_domain = not (algo in (2, 3) and not kw["only_score"]) or \
                exact_cached(p, t) <= max(len(p), len(t)) * kw["bandwidth"] // 100
            assert out[i][0] == est or not in_domain, (kw, i)
            if est >= 0 and in_domain:
                assert out[i][1] == esc, (kw, i, len(p), len(t))
                assert out[i][2] == ecg, (kw, i)
        if algo == 0:
            # the same through a resident batch: its first run is a host-driven one (and sets the bound estimate), the second
            # -- synchronous -- and the third -- queued, then fetched -- go through the fast flow (the aligner's calls above
            # never do: its stage timers bracket host-synchronous stages)
            rb = capi.ResidentBatch(datagen.PairBatch(*_pools(pairs)))
            pp = capi.make_params(**kw)
            for rep in range(3):
                st = rb.run(pp, sync=rep < 2)
                assert st >= 0 or st == capi.QUICKED_EMPTY_SEQUENCE, st
                if rep == 2:
                    rb.fetch()
                sc, stt = rb.scores()
                cg = None if kw["only_score"] else rb.cigars()
                for i, (p, t) in enumerate(pairs):
                    est, esc, ecg = oracle_cached(p, t, **kw)
                    assert stt[i] == est, (kw, i, rep)
                    if est >= 0:
                        assert sc[i] == esc and (cg is None or cg[i] == ecg), (kw, i, rep, len(p), len(t))
            rb.close()


def test_quicked_only_score_pass(monkeypatch):
    """QuickEd with only_score: one score-only pass over the FILL's cells (run_fill_score, BandedArgs::fill_geom; one lane, 16
    lanes or a wave per alignment by the launch's size) instead of fill + traceback + edit count.  On (the default) and
    switched off: the oracle's scores and statuses on 10 kb reads, indel-heavy pairs (stages 2 / 3), ragged lengths with N; a batch
    with lower-case / IUPAC symbols (the reference's traceback compares raw bytes, bpm_banded.c:1012: the library must keep
    the align step there); queued runs, where the pass sits inside the fast flow with its cutoffs still on the device and
    the pairs that leave the flow are finished afterwards; and the same array as the align step's with the same block-advance
    count and no traceback step."""
    sets = [("10 kb", list(datagen.generate(192, 10000, 0.05, seed=611).pairs())),
            ("indels", list(datagen.generate(160, 3000, 0.1, seed=612, indels_num=2, indels_len=300).pairs())),
            ("mixed", mixed_batch())]
    rng = np.random.default_rng(613)
    ragged = []
    for i in range(128):
        L = int(rng.choice([1, 2, 63, 64, 65, 129, 500, 1000, 2500, 4000]))
        e = float(rng.choice([0.0, 0.02, 0.1, 0.3]))
        pt, tx = next(datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=61300 + i).pairs())
        if rng.random() < 0.2:
            tx = tx[: max(1, len(tx) - int(rng.integers(0, max(1, len(tx) // 3))))]
        if rng.random() < 0.15:
            pt = pt.replace(b"A", b"N", 2)
        ragged.append((pt, tx))
    sets.append(("ragged", ragged))
    kws = (dict(algo=0, only_score=True), dict(algo=0, only_score=True, force_scalar=True),
           dict(algo=0, only_score=True, window_size=5, overlap_size=2, hew_threshold=(10, 40), hew_percentage=(1, 15)))
    for mode in ("1", "0"):
        monkeypatch.setenv("QE_QUICKED_SCORE_PASS", mode)
        for name, pairs in sets:
            for kw in kws:
                al = capi.QuickedAligner()
                for k, v in kw.items():
                    if k in ("hew_threshold", "hew_percentage"):
                        getattr(al._params, k)[0], getattr(al._params, k)[1] = v
                    else:
                        setattr(al._params, k, v)
                want = oracle_many(pairs, **kw)
                st, out = al.alignBatch(pairs)                # (the host-driven flow: an aligner's stage timers keep it there)
                assert out == want, (mode, name, kw, [i for i in range(len(pairs)) if out[i] != want[i]][:5])
            # queued runs (the fast flow from the second on; pairs that leave it -- stage 2, bounds above the estimate, raw
            # symbols -- finished by the library's threads or the fetch), a small forced estimate once
            kw = kws[0]
            want = oracle_many(pairs, **kw)
            rb = capi.ResidentBatch(datagen.PairBatch(*_pools(pairs)))
            p = capi.make_params(**kw)
            for rep in range(4):
                if rep == 3:
                    monkeypatch.setenv("QE_QUICKED_EST", "40")
                assert rb.run(p, sync=False) >= 0
                rb.fetch()
                sc, stt = rb.scores()
                out = [(int(stt[i]), int(sc[i])) for i in range(len(pairs))]
                assert out == [w[:2] for w in want], (mode, name, rep, [i for i in range(len(pairs)) if out[i] != want[i][:2]][:5])
            monkeypatch.delenv("QE_QUICKED_EST", raising=False)
            capi.reload_env()
            rb.close()
    # reads whose align step splits (bpm_hirschberg.c:63-65) keep it, forced or not: the levels' half passes beat one pass
    b = datagen.generate(48, 30000, 0.08, seed=615)
    want = oracle_many(list(b.pairs()), algo=0, only_score=True)
    for mode in ("1", None):
        if mode is None:
            monkeypatch.delenv("QE_QUICKED_SCORE_PASS", raising=False)
            capi.reload_env()
        else:
            monkeypatch.setenv("QE_QUICKED_SCORE_PASS", mode)
        sc, stt, _, cnt = gpu_batch(b, algo=0, only_score=True)
        assert [(int(stt[i]), int(sc[i]), None) for i in range(len(b))] == want, mode
        assert cnt[3] > 0, (mode, cnt)                         # traceback steps: the align step ran
    # the library's own choice (the pass: 20 000 pairs are past the size below which a waited-for run of tall bands keeps
    # the align step), one lane per alignment at this size
    monkeypatch.delenv("QE_QUICKED_SCORE_PASS", raising=False)
    capi.reload_env()
    b = datagen.generate(20000, 1000, 0.05, seed=614)
    rb = capi.ResidentBatch(b)
    p = capi.make_params(algo=0, only_score=True)
    got = []
    for rep in range(2):
        assert rb.run(p, sync=True) >= 0
        got.append((rb.scores()[0].copy(), rb.scores()[1].copy(), rb.counters().copy()))
    assert rb.run(p, sync=False) >= 0                          # queued: the pass inside the fast flow, cutoffs from the device
    rb.fetch()
    queued = (rb.scores()[0].copy(), rb.scores()[1].copy(), rb.counters().copy())
    monkeypatch.setenv("QE_QUICKED_SCORE_PASS", "0")
    assert rb.run(p, sync=True) >= 0
    off = (rb.scores()[0].copy(), rb.scores()[1].copy(), rb.counters().copy())
    rb.close()
    for sc, stt, cnt in got + [queued]:
        assert np.array_equal(sc, off[0]) and np.array_equal(stt, off[1])
    for sc, stt, cnt in got + [queued]:
        assert cnt[3] == 0 and cnt[4] == 0, cnt                # no traceback step: the pass ran
        assert cnt[1] == off[2][1] and cnt[0] == off[2][0] and cnt[2] == off[2][2], (cnt, off[2])
    assert off[2][3] > 0
    idx = list(range(0, len(b), 61))
    pl = list(b.pairs())
    want = oracle_many([pl[i] for i in idx], algo=0, only_score=True)
    assert [(int(off[1][i]), int(off[0][i]), None) for i in idx] == want


@pytest.mark.parametrize("cp", ["1", "0"])
def test_windowed_checkpoint_and_history_paths(cp, monkeypatch):
    """WindowEd for any window shape but (2, 1): k_windowed_cp keeps checkpoints + carry words and recomputes the tile the
    in-window traceback is in (QE_WINDOWED_CP = 1, the default); QE_WINDOWED_CP = 0 is the path that stores every column's
    history (what waves with N / non-ACGT input still use).  Both equal the oracle: scores, HEW-driven QuickEd stages,
    CIGARs, work counters -- on ragged lengths, partial windows and pairs with large indels."""
    monkeypatch.setenv("QE_WINDOWED_CP", cp)
    rng = np.random.default_rng(77)
    pairs = []
    for i in range(80):
        L = int(rng.choice([1, 63, 64, 65, 300, 575, 576, 577, 1200, 3000, 5000]))
        e = float(rng.choice([0.0, 0.03, 0.1, 0.25]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=7700 + i,
                             indels_num=int(rng.integers(0, 3)) if L >= 3000 else 0, indels_len=200)
        p, t = next(b.pairs())
        if rng.random() < 0.2:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 3))))]
        pairs.append((p, t))
    for kw in (dict(algo=1, only_score=True), dict(algo=1), dict(algo=1, window_size=3, overlap_size=1),
               dict(algo=1, only_score=True, window_size=4, overlap_size=2), dict(algo=1, window_size=7, overlap_size=5),
               dict(algo=1, window_size=2, overlap_size=1, force_scalar=True),
               dict(algo=0, hew_threshold=(10, 10), hew_percentage=(1, 1)), dict(algo=0, window_size=5, overlap_size=2, hew_threshold=(10, 40), hew_percentage=(1, 15))):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            if k in ("hew_threshold", "hew_percentage"):
                getattr(al._params, k)[0], getattr(al._params, k)[1] = v
            else:
                setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            assert out[i] == oracle_cached(p, t, **kw), (cp, kw, i, len(p), len(t))
    # work counters: block-advances of the windows are the oracle's whichever path ran
    b = datagen.generate(64, 3000, 0.1, seed=78, indels_num=1, indels_len=300)
    _, _, _, cnt = gpu_batch(b, algo=1, only_score=True)
    assert cnt[2] == sum(oracle_cached(p, t, trace=True, algo=1, only_score=True)[3]["window_block_steps"] for p, t in b.pairs())


@pytest.mark.parametrize("quad", ["1", "0"])
def test_windowed_quad_forced(quad, golden, monkeypatch):
    """WindowEd(2, 1) score-only with four lanes per alignment (k_windowed_quad: 32-bit blocks in a systolic array, the
    chain of full windows; k_windowed finishes every task's clamped windows) against one lane per alignment
    (QE_WINDOWED_QUAD = 0): the goldens captured from the compiled reference (scalar and x86-SSE window semantics, the
    QuickEd stages the HEW counts drive), the oracle on ragged / N / lower-case input, indel-heavy pairs whose windows
    leave the diagonal, windows that start at the text's first bases, and the work counters."""
    monkeypatch.setenv("QE_WINDOWED_QUAD", quad)
    for name in ("cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len128", "len130", "len1024"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            if run["params"].get("algo") not in (0, 1):
                continue
            scores, status, cig, _ = gpu_batch(batch, **run["params"])
            assert status.tolist() == run["status"], (name, label)
            assert scores.tolist() == run["score"], (name, label)
            if "cigar_sha256" in run:
                assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
    rng = np.random.default_rng(91)
    pairs = mixed_batch()
    for i in range(60):
        L = int(rng.choice([127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 700, 2500, 6000]))
        e = float(rng.choice([0.0, 0.03, 0.1, 0.3]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=9100 + i,
                             indels_num=int(rng.integers(0, 3)) if L >= 2500 else 0, indels_len=150)
        p, t = next(b.pairs())
        if rng.random() < 0.3:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 3))))]
        pairs.append((p, t))
    for kw in (dict(algo=1, only_score=True, window_size=2, overlap_size=1),
               dict(algo=1, only_score=True, window_size=2, overlap_size=1, force_scalar=True),
               dict(algo=0), dict(algo=0, force_scalar=True), dict(algo=0, only_score=True),
               dict(algo=0, hew_threshold=(10, 10), hew_percentage=(1, 1))):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            if k in ("hew_threshold", "hew_percentage"):
                getattr(al._params, k)[0], getattr(al._params, k)[1] = v
            else:
                setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            assert out[i] == oracle_cached(p, t, **kw), (quad, kw, i, len(p), len(t))
    b = datagen.generate(80, 3000, 0.1, seed=92, indels_num=1, indels_len=300)
    for kw in (dict(algo=1, only_score=True, window_size=2, overlap_size=1), dict(algo=0)):
        _, _, _, cnt = gpu_batch(b, **kw)
        assert cnt[2] == sum(oracle_cached(p, t, trace=True, **kw)[3]["window_block_steps"] for p, t in b.pairs()), kw


@pytest.mark.parametrize("wsys", ["1", "0"])
def test_windowed_systolic_forced(wsys, golden, monkeypatch):
    """WindowEd score-only for any window shape of up to 15 blocks with sixteen lanes per alignment (k_windowed_sys: the
    window's block rows as a systolic array, sixteen traceback tiles rebuilt at a time; QE_WINDOWED_SYS = 1) against
    k_windowed_cp (0): the goldens (WindowEd(9, 1) scores, the QuickEd stages the HEW counts of stage 2 drive -- forward and
    reversed), the oracle on ragged lengths, clamped windows, pairs with large indels, N / lower-case input (flagged tasks
    fall back), several window shapes, and the window-step counter."""
    monkeypatch.setenv("QE_WINDOWED_SYS", wsys)
    for name in ("cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len128", "len130", "len1024"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            prm = run["params"]
            if not ((prm.get("algo") == 1 and prm.get("only_score") and prm.get("window_size", 9) != 2) or (prm.get("algo") == 0 and name == "indel_10kb")):
                continue
            scores, status, cig, _ = gpu_batch(batch, **prm)
            assert status.tolist() == run["status"], (name, label)
            assert scores.tolist() == run["score"], (name, label)
            if "cigar_sha256" in run:
                assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
    rng = np.random.default_rng(37)
    pairs = mixed_batch()
    for i in range(70):
        L = int(rng.choice([1, 63, 64, 65, 300, 575, 576, 577, 1200, 3000, 5000, 9000]))
        e = float(rng.choice([0.0, 0.03, 0.1, 0.25]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=3700 + i,
                             indels_num=int(rng.integers(0, 3)) if L >= 3000 else 0, indels_len=int(rng.choice([200, 700])))
        p, t = next(b.pairs())
        if rng.random() < 0.2:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 3))))]
        pairs.append((p, t))
    for kw in (dict(algo=1, only_score=True), dict(algo=1, only_score=True, window_size=3, overlap_size=1),
               dict(algo=1, only_score=True, window_size=4, overlap_size=2), dict(algo=1, only_score=True, window_size=7, overlap_size=5),
               dict(algo=1, only_score=True, window_size=15, overlap_size=1),
               # W == 2 away from the on-chip (2, 1) shape: the reference runs its SSE window kernel there whenever force_scalar
               # is off, whatever the overlap (bpm_windowed.c:577) -- scores differ from the scalar kernel's on most pairs
               dict(algo=1, only_score=True, window_size=2, overlap_size=0),
               dict(algo=1, only_score=True, window_size=2, overlap_size=0, force_scalar=True),
               dict(algo=0, window_size=2, overlap_size=0, hew_threshold=(10, 10), hew_percentage=(1, 1)),
               dict(algo=0, hew_threshold=(10, 10), hew_percentage=(1, 1)),
               dict(algo=0, window_size=5, overlap_size=2, hew_threshold=(10, 40), hew_percentage=(1, 15))):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            if k in ("hew_threshold", "hew_percentage"):
                getattr(al._params, k)[0], getattr(al._params, k)[1] = v
            else:
                setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            assert out[i] == oracle_cached(p, t, **kw), (wsys, kw, i, len(p), len(t))
    b = datagen.generate(64, 3000, 0.1, seed=38, indels_num=1, indels_len=300)
    for kw in (dict(algo=1, only_score=True), dict(algo=1, only_score=True, window_size=4, overlap_size=2), dict(algo=0, hew_threshold=(10, 10), hew_percentage=(1, 1))):
        _, _, _, cnt = gpu_batch(b, **kw)
        assert cnt[2] == sum(oracle_cached(p, t, trace=True, **kw)[3]["window_block_steps"] for p, t in b.pairs()), (wsys, kw)


@pytest.mark.parametrize("multi", ["1", "0"])
def test_fill_multi_slot_passes_forced(multi, monkeypatch):
    """the BandEd fill runs K = 3 band slots per skewed pass, every lane masked to its own band (QE_FILL_MULTI = 1, default),
    or one slot per pass (0): same checkpoints, same CIGARs -- tight QuickEd bands, loose BandEd bands, Hirschberg leaves"""
    monkeypatch.setenv("QE_FILL_MULTI", multi)
    rng = np.random.default_rng(5)
    pairs = []
    for i in range(72):
        L = int(rng.choice([64, 200, 1000, 2500, 6000]))
        e = float(rng.choice([0.01, 0.05, 0.2]))
        p, t = next(datagen.generate(1, L, e, seed=9100 + i, indels_num=int(rng.integers(0, 2)) if L >= 2500 else 0, indels_len=150).pairs())
        pairs.append((p, t))
    for kw in (dict(algo=0), dict(algo=2, bandwidth=30), dict(algo=3, bandwidth=30), dict(algo=2, bandwidth=60)):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            want = oracle_cached(p, t, **kw)
            in_domain = kw["algo"] == 0 or exact_cached(p, t) <= max(len(p), len(t)) * kw["bandwidth"] // 100
            if in_domain:
                assert out[i] == want, (multi, kw, i, len(p), len(t))


@pytest.mark.parametrize("sys_", ["1", "0"])
def test_fill_systolic_forced(sys_, golden, monkeypatch):
    """the fill of TIGHT bands (QuickEd's bound, the exact child distances of a Hirschberg split) with sixteen lanes per
    leaf -- the band's rows as a systolic array, the reference's 64-column bookkeeping at every chunk's end
    (k_banded_sys; QE_FILL_SYS = 1) -- against one lane per leaf (0): the goldens captured from the compiled reference
    (CIGAR bytes), the oracle on ragged / N / lower-case input (flagged leaves fall back), forced deep splits whose
    children start anywhere in their pair's pattern, pairs with large indels (bands of more than 15 slots fall back),
    and the fill's block-advance counter."""
    monkeypatch.setenv("QE_FILL_SYS", sys_)
    for name in ("cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len63", "len64", "len65", "len128", "len130", "len1024"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            if run["params"].get("algo") != 0:
                continue
            scores, status, cig, _ = gpu_batch(batch, **run["params"])
            assert status.tolist() == run["status"], (name, label)
            assert scores.tolist() == run["score"], (name, label)
            if "cigar_sha256" in run:
                assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
    rng = np.random.default_rng(17)
    pairs = mixed_batch()
    for i in range(72):
        L = int(rng.choice([1, 63, 64, 65, 200, 1000, 1023, 1024, 1025, 2500, 6000]))
        e = float(rng.choice([0.0, 0.01, 0.05, 0.2]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=1700 + i,
                             indels_num=int(rng.integers(0, 2)) if L >= 2500 else 0, indels_len=150)
        p, t = next(b.pairs())
        if rng.random() < 0.25:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 4))))]
        pairs.append((p, t))
    for kw in (dict(algo=0), dict(algo=0, force_scalar=True), dict(algo=0, only_score=True)):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            assert out[i] == oracle_cached(p, t, **kw), (sys_, kw, i, len(p), len(t))
    # forced deep splits (the oracle's Hirschberg with the same threshold, cf. test_hirschberg_forced_deep_splits)
    import ctypes as C
    monkeypatch.setenv("QE_SPLIT_BYTES", str(1 << 15))
    lib = O.oracle()
    batch = datagen.generate(count=70, length=3000, error=0.08, seed=1801)
    scores, status, cig, _ = gpu_batch(batch, algo=0)
    for i, (p, t) in enumerate(batch.pairs()):
        st, sc, cg, tr = oracle_cached(p, t, trace=True, algo=0)
        ops = C.create_string_buffer(len(p) + len(t) + 1)
        nn = C.c_int64()
        lib.qo_hirschberg(p, len(p), t, len(t), tr["bound"], 1 << 15, ops, C.byref(nn), None)
        buf = C.create_string_buffer(2 * nn.value + 16)
        lib.qo_cigar_rle(ops, nn.value, buf)
        assert scores[i] == sc and cig[i] == buf.value.decode(), (sys_, i)
    monkeypatch.delenv("QE_SPLIT_BYTES")
    b = datagen.generate(96, 3000, 0.05, seed=18)
    _, _, _, cnt = gpu_batch(b, algo=0)
    tr = [oracle_cached(p, t, trace=True, algo=0)[3] for p, t in b.pairs()]
    assert cnt[1] == sum(x["fill_block_advances"] for x in tr)
    assert cnt[3] == sum(x["traceback_steps"] for x in tr)
    # bands of 64 .. 127 slots (bounds of 4 000 - 8 000: several large indels): two rows per lane, two sweeps (k_banded_sys2)
    tall = datagen.generate(24, 12000, 0.05, seed=19, indels_num=6, indels_len=900)
    s_, st_, cg_, cnt = gpu_batch(tall, algo=0)
    work = 0
    for i, (p, t) in enumerate(tall.pairs()):
        est, esc, ecg, trc = oracle_cached(p, t, trace=True, algo=0)
        work += trc["fill_block_advances"]
        assert (st_[i], s_[i], cg_[i]) == (est, esc, ecg), (sys_, "tall", i)
    assert cnt[1] == work


@pytest.mark.parametrize("tsys", ["1", "8", "4", "0"])
def test_traceback_systolic_forced(tsys, golden, monkeypatch):
    """the BandEd traceback with sixteen lanes per leaf (k_traceback_sys: the tiles along the path's diagonal rebuilt
    together, the walk handed from tile to tile; QE_TRACE_SYS = 1) against one lane per leaf (0): CIGAR bytes of the
    goldens, the oracle on tight QuickEd bands, loose BandEd / Hirschberg bands (a user bandwidth), ragged / N / lower-case
    input (flagged leaves fall back), pairs with large indels (long runs that leave the predicted tiles), forced deep
    splits, and the traceback's step counter."""
    monkeypatch.setenv("QE_TRACE_SYS", tsys)
    for name in ("cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len63", "len64", "len65", "len128", "len130", "len1024", "err35_2kb"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            if "cigar_sha256" not in run or run["params"].get("algo") == 1:
                continue
            scores, status, cig, _ = gpu_batch(batch, **run["params"])
            assert status.tolist() == run["status"], (name, label)
            assert scores.tolist() == run["score"], (name, label)
            assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
    rng = np.random.default_rng(23)
    pairs = mixed_batch()
    for i in range(72):
        L = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 200, 1000, 2500, 6000]))
        e = float(rng.choice([0.0, 0.01, 0.05, 0.2]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=2300 + i,
                             indels_num=int(rng.integers(0, 3)) if L >= 2500 else 0, indels_len=int(rng.choice([40, 150, 400])))
        p, t = next(b.pairs())
        if rng.random() < 0.25:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 4))))]
        pairs.append((p, t))
    for kw in (dict(algo=0), dict(algo=2, bandwidth=30), dict(algo=3, bandwidth=30), dict(algo=2, bandwidth=60)):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            want = oracle_cached(p, t, **kw)
            in_domain = kw["algo"] == 0 or len(p) == 0 or len(t) == 0 or \
                exact_cached(p, t) <= max(len(p), len(t)) * kw["bandwidth"] // 100
            if in_domain:
                assert out[i] == want, (tsys, kw, i, len(p), len(t))
    import ctypes as C
    monkeypatch.setenv("QE_SPLIT_BYTES", str(1 << 15))
    lib = O.oracle()
    batch = datagen.generate(count=70, length=3000, error=0.08, seed=2401)
    scores, status, cig, _ = gpu_batch(batch, algo=0)
    for i, (p, t) in enumerate(batch.pairs()):
        st, sc, cg, tr = oracle_cached(p, t, trace=True, algo=0)
        ops = C.create_string_buffer(len(p) + len(t) + 1)
        nn = C.c_int64()
        lib.qo_hirschberg(p, len(p), t, len(t), tr["bound"], 1 << 15, ops, C.byref(nn), None)
        buf = C.create_string_buffer(2 * nn.value + 16)
        lib.qo_cigar_rle(ops, nn.value, buf)
        assert scores[i] == sc and cig[i] == buf.value.decode(), (tsys, i)
    monkeypatch.delenv("QE_SPLIT_BYTES")
    b = datagen.generate(96, 3000, 0.05, seed=24, indels_num=1, indels_len=100)
    _, _, _, cnt = gpu_batch(b, algo=0)
    tr = [oracle_cached(p, t, trace=True, algo=0)[3] for p, t in b.pairs()]
    assert cnt[3] == sum(x["traceback_steps"] for x in tr)
    assert cnt[1] == sum(x["fill_block_advances"] for x in tr)


@pytest.mark.parametrize("rel", ["1", "2"])
def test_systolic_steady_blocks_switch(rel, monkeypatch):
    """k_banded_sys runs the sixteen-step units in which every row of every band is at work under ONE mask, the top row's
    boundary carry coming from a DPP move that leaves a masked-out source's destination as it was (QE_LANE_REL = 1, the
    default) or asks every step (2): same scores, CIGAR bytes and block-advance counts as the oracle either way, the
    systolic fill and score-only forms forced, tight QuickEd bands, a user bandwidth, ragged and N input."""
    monkeypatch.setenv("QE_LANE_REL", rel)
    monkeypatch.setenv("QE_FILL_SYS", "1")
    monkeypatch.setenv("QE_SCORE_SYS", "1")
    pairs = mixed_batch()
    rng = np.random.default_rng(41)
    for i in range(40):
        L = int(rng.choice([64, 65, 128, 700, 1000, 2500, 6000, 10000]))
        e = float(rng.choice([0.0, 0.02, 0.05, 0.2]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=4100 + i,
                             indels_num=int(rng.integers(0, 3)) if L >= 2500 else 0, indels_len=150)
        p, t = next(b.pairs())
        if rng.random() < 0.3:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 3))))]
        pairs.append((p, t))
    for kw in (dict(algo=0), dict(algo=2, only_score=True, bandwidth=15), dict(algo=2, bandwidth=20), dict(algo=2, only_score=True, bandwidth=40)):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            want = oracle_cached(p, t, **kw)
            in_domain = kw["algo"] == 0 or len(p) == 0 or len(t) == 0 or exact_cached(p, t) <= max(len(p), len(t)) * kw["bandwidth"] // 100
            if in_domain:
                assert out[i] == want, (rel, kw, i, len(p), len(t))


@pytest.mark.parametrize("ssys", ["1", "0"])
def test_score_systolic_forced(ssys, golden, monkeypatch):
    """BandEd score-only over whole texts with the band's rows as a systolic array (k_banded_sys<.., false>: sixteen lanes
    per task for bands of <= 15 slots, a wave per task up to 63; QE_SCORE_SYS = 1) against the other forms (0): the
    goldens incl. the geometry-dependent low bandwidths, QuickEd's stage-3 doubling rounds (indel_10kb), the oracle on
    ragged / N / lower-case input (flagged tasks fall back) and the block-advance counter."""
    monkeypatch.setenv("QE_SCORE_SYS", ssys)
    for name in ("cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len50", "len63", "len64", "len65", "len128", "len130", "len1024", "err35_2kb"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            prm = run["params"]
            if not ((prm.get("algo") == 2 and prm.get("only_score")) or (prm.get("algo") == 0 and name == "indel_10kb")):
                continue
            scores, status, cig, _ = gpu_batch(batch, **prm)
            assert status.tolist() == run["status"], (name, label)
            assert scores.tolist() == run["score"], (name, label)
            if "cigar_sha256" in run:
                assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
    pairs = mixed_batch()
    rng = np.random.default_rng(29)
    for i in range(60):
        L = int(rng.choice([1, 63, 64, 65, 127, 128, 129, 700, 2500, 6000]))
        e = float(rng.choice([0.0, 0.02, 0.1, 0.3]))
        b = datagen.generate(1, L, e if e * L >= 1 or e == 0 else 1, seed=2900 + i,
                             indels_num=int(rng.integers(0, 3)) if L >= 2500 else 0, indels_len=200)
        p, t = next(b.pairs())
        if rng.random() < 0.3:
            t = t[: max(1, len(t) - int(rng.integers(0, max(1, len(t) // 3))))]
        pairs.append((p, t))
    for kw in (dict(algo=2, only_score=True, bandwidth=1), dict(algo=2, only_score=True, bandwidth=4),
               dict(algo=2, only_score=True, bandwidth=15), dict(algo=2, only_score=True, bandwidth=40),
               dict(algo=0, hew_threshold=(10, 10), hew_percentage=(1, 1))):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            if k in ("hew_threshold", "hew_percentage"):
                getattr(al._params, k)[0], getattr(al._params, k)[1] = v
            else:
                setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            assert out[i] == oracle_cached(p, t, **kw), (ssys, kw, i, len(p), len(t))
    # bands of 64 .. 127 slots: k_banded_sys2<false>
    tall = datagen.generate(24, 10000, 0.05, seed=33, indels_num=2, indels_len=700)
    for bw in (45, 70):
        sc_, st_, _, cnt = gpu_batch(tall, algo=2, only_score=True, bandwidth=bw)
        tr = [oracle_cached(p, t, trace=True, algo=2, only_score=True, bandwidth=bw) for p, t in tall.pairs()]
        assert sc_.tolist() == [x[1] for x in tr] and st_.tolist() == [x[0] for x in tr], (ssys, bw)
        # (the cooperative LDS form that runs instead when the systolic form is off counts the slot a lane walks on a "no
        # cut yet" guess that the bookkeeping later cuts: its block-advance counter may exceed the oracle's by a few slots)
        if ssys == "1":
            assert cnt[0] == sum(x[3]["score_block_advances"] for x in tr), (ssys, bw)
    b = datagen.generate(64, 2000, 0.05, seed=30)
    for bw in (3, 15):
        _, _, _, cnt = gpu_batch(b, algo=2, only_score=True, bandwidth=bw)
        assert cnt[0] == sum(oracle_cached(p, t, trace=True, algo=2, only_score=True, bandwidth=bw)[3]["score_block_advances"] for p, t in b.pairs())


@pytest.mark.parametrize("dev", ["1", "0"])
def test_stage3_band_doubling_on_the_device(dev, golden, monkeypatch):
    """QuickEd's stage 3 (quicked.c:248-278: score-only BandEd, the cutoff doubled while the score says it was too small) as
    ONE launch with the convergence test on the device (k_banded_sys<6, false> with `doubling`; QE_STAGE3_DEVICE = 1) against
    the host-driven rounds (0): statuses, scores and CIGAR bytes of the compiled reference on the indel goldens, the oracle
    on pairs whose bounds need several doublings (HEW parameters that send everything to stage 3, small bandwidths so that
    the first cutoffs are far too small), ragged / N input (flagged tasks continue on the host), and the work counters --
    block advances summed over every round, pairs that reached stage 3."""
    monkeypatch.setenv("QE_STAGE3_DEVICE", dev)
    for name in ("indel_10kb", "cfg1_1kb_5pct"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            if run["params"].get("algo") != 0:
                continue
            scores, status, cig, _ = gpu_batch(batch, **run["params"])
            assert status.tolist() == run["status"], (name, label)
            assert scores.tolist() == run["score"], (name, label)
            if "cigar_sha256" in run:
                assert [sha(c) for c in cig] == run["cigar_sha256"], (name, label)
    rng = np.random.default_rng(31)
    pairs = mixed_batch()
    for i in range(56):
        L = int(rng.choice([300, 1000, 2500, 6000, 9000]))
        e = float(rng.choice([0.02, 0.1, 0.3]))
        b = datagen.generate(1, L, e, seed=3100 + i, indels_num=int(rng.integers(0, 4)) if L >= 2500 else 0, indels_len=int(rng.choice([100, 400, 900])))
        pairs.append(next(b.pairs()))
    for kw in (dict(algo=0, hew_threshold=(1, 1), hew_percentage=(1, 1), bandwidth=1),
               dict(algo=0, hew_threshold=(1, 1), hew_percentage=(1, 1), bandwidth=5),
               dict(algo=0, hew_threshold=(10, 10), hew_percentage=(1, 1)), dict(algo=0)):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            if k in ("hew_threshold", "hew_percentage"):
                getattr(al._params, k)[0], getattr(al._params, k)[1] = v
            else:
                setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            assert out[i] == oracle_cached(p, t, **kw), (dev, kw, i, len(p), len(t))
    b = datagen.generate(48, 5000, 0.05, seed=32, indels_num=3, indels_len=500)
    kw = dict(algo=0, bandwidth=2)
    _, _, _, cnt = gpu_batch(b, **kw)
    tr = [oracle_cached(p, t, trace=True, **kw)[3] for p, t in b.pairs()]
    assert cnt[0] == sum(x["score_block_advances"] for x in tr), (dev, int(cnt[0]))
    assert cnt[7] == sum(1 for x in tr if x["stage"] >= 3) and cnt[7] > 0


@pytest.mark.parametrize("tall", ["1", "0"])
def test_tall_band_cooperative_fill(tall, monkeypatch):
    """QuickEd's align step on pairs with LARGE bounds (large indels: bands of 30-50 slots) in a launch of few waves: the
    leaves whose bands are tall enough fill with G lanes each (k_banded_coop_lds<true>), the others -- ordinary pairs in
    the same list -- arrive flagged and stay with the one-lane kernel (QE_COOP_TALL_FILL = 1, default; 0: one lane for all).
    Same cells either way: statuses, scores, CIGARs and the fill's block-advance counter equal the oracle's."""
    monkeypatch.setenv("QE_COOP_TALL_FILL", tall)
    rng = np.random.default_rng(23)
    pairs = []
    for i in range(160):
        L = int(rng.choice([5000, 8000]))
        hard = rng.random() < 0.6
        p, t = next(datagen.generate(1, L, 0.05, seed=9500 + i, indels_num=3 if hard else 0, indels_len=int(rng.choice([400, 700]))).pairs())
        pairs.append((p, t))
    batch = datagen.PairBatch(*_pools(pairs))
    s, st, cg, cnt = gpu_batch(batch, algo=0)
    work = 0
    for i, (p, t) in enumerate(pairs):
        est, esc, ecg, tr = oracle_cached(p, t, trace=True, algo=0)
        work += tr["fill_block_advances"]
        assert (st[i], s[i], cg[i]) == (est, esc, ecg), (tall, i)
    assert cnt[1] == work, (tall, int(cnt[1]), work)


@pytest.mark.parametrize("rel", ["1", "0"])
def test_lane_relative_band_walk_forced(rel, monkeypatch):
    """k_banded walks, in every chunk, either the union of its 64 lanes' bands (QE_LANE_REL = 0) or every lane's own band
    from its own top slot (1, default; what pairs whose paths drift apart need: large indels scatter the tight QuickEd
    bands of a wave over dozens of slots).  Same cells either way: scores, CIGARs AND the block-advance counters equal the
    oracle's, for the fill (QuickEd, BandEd, Hirschberg leaves) and for the score-only kernel, on a batch that mixes
    ordinary pairs with large-indel ones of several lengths."""
    monkeypatch.setenv("QE_LANE_REL", rel)
    rng = np.random.default_rng(17)
    pairs = []
    for i in range(160):
        L = int(rng.choice([900, 2000, 4000, 7000]))
        ind = int(rng.integers(0, 4)) if L >= 2000 else 0
        p, t = next(datagen.generate(1, L, 0.05, seed=9300 + i, indels_num=ind, indels_len=int(rng.choice([100, 300, 600]))).pairs())
        pairs.append((p, t))
    batch = datagen.PairBatch(*_pools(pairs))
    for kw, slot, key in ((dict(algo=0), 1, "fill_block_advances"), (dict(algo=2, bandwidth=40), 1, "fill_block_advances"),
                          (dict(algo=3, bandwidth=40), 1, "fill_block_advances"), (dict(algo=2, bandwidth=40, only_score=True), 0, "score_block_advances")):
        s, st, cg, cnt = gpu_batch(batch, **kw)
        work = 0
        for i, (p, t) in enumerate(pairs):
            est, esc, ecg, tr = oracle_cached(p, t, trace=True, **kw)
            work += tr[key]
            in_domain = kw["algo"] == 0 or exact_cached(p, t) <= max(len(p), len(t)) * kw["bandwidth"] // 100
            if in_domain:
                assert (st[i], s[i]) == (est, esc), (rel, kw, i)
                if cg is not None:
                    assert cg[i] == ecg, (rel, kw, i)
        if kw["algo"] != 3:                    # Hirschberg's half passes count into the score-only slot
            assert cnt[slot] == work, (rel, kw, int(cnt[slot]), work)


@pytest.mark.parametrize("lds", ["1", "0"])
def test_cooperative_kernel_forced(lds, monkeypatch):
    """QE_COOP_G forces the G-lanes-per-alignment kernel (and its fallback pass) on shapes the host would not pick it for:
    the on-chip form (k_banded_coop_lds: uniform masked multi-slot passes, band state in LDS) and the global-memory form
    (QE_COOP_LDS = 0), every lane count, scores AND block-advance counts equal to the oracle's; ragged lengths with N
    symbols; Hirschberg half passes (stopped bands exported for the join) with forced splits"""
    monkeypatch.setenv("QE_COOP_LDS", lds)
    batch = datagen.generate(count=300, length=6000, error=0.07, seed=91)
    pairs = list(batch.pairs())
    ref = {}
    for bw in (8, 15, 40):
        tr = [oracle_cached(p, t, trace=True, algo=2, only_score=True, bandwidth=bw) for p, t in pairs]
        ref[bw] = ([x[1] for x in tr], sum(x[3]["score_block_advances"] for x in tr))
    for G in ("2", "4", "8", "16", "32"):
        monkeypatch.setenv("QE_COOP_G", G)
        for bw in (8, 15, 40):
            scores, status, _, cnt = gpu_batch(batch, algo=2, only_score=True, bandwidth=bw)
            assert scores.tolist() == ref[bw][0], (G, bw)
            assert cnt[0] == ref[bw][1], (G, bw, int(cnt[6]))
    # ragged lengths (64 / G tasks of different heights per wave), N symbols, short reads whose band is the whole matrix
    rng = np.random.default_rng(17)
    rag = []
    for i in range(200):
        L = int(rng.choice([700, 1500, 2500, 4000, 6000]))
        b = datagen.generate(1, L, 0.06, seed=9100 + i)
        p, t = next(b.pairs())
        if i % 5 == 0:
            p = bytearray(p)
            for k in rng.integers(0, len(p), 4): p[k] = ord("N")
            p = bytes(p)
        rag.append((p, t))
    al = capi.QuickedAligner()
    al.setAlgorithm(capi.BANDED); al.setOnlyScore(True); al.setBandwidth(30)
    exp = [oracle_cached(p, t, algo=2, only_score=True, bandwidth=30)[:2] for p, t in rag]
    for G in ("2", "4", "8"):
        monkeypatch.setenv("QE_COOP_G", G)
        st, out = al.alignBatch(rag)
        assert [(o[0], o[1]) for o in out] == exp, G
    # Hirschberg: the half passes stop mid-text and the join reads their bands from the workspace the kernel exports
    monkeypatch.setenv("QE_SPLIT_BYTES", str(1 << 17))
    hb = datagen.generate(count=40, length=6000, error=0.07, seed=92)
    hexp = None
    for G in ("1", "4", "16"):
        monkeypatch.setenv("QE_COOP_G", G)
        scores, status, cig, cnt = gpu_batch(hb, algo=3, bandwidth=20)
        got = (scores.tolist(), status.tolist(), cig, int(cnt[0]))
        if hexp is None:
            hexp = got
            assert all(s == exact_cached(p, t) for s, (p, t) in zip(got[0], hb.pairs()))
        assert got == hexp, G


def test_cooperative_fill_forced(monkeypatch):
    """QE_COOP_FILL_G forces the G-lanes-per-leaf fill (k_banded_coop_lds<true>: band state on chip, checkpoints / carry words /
    band edges written in the one-lane fill's layout for the same k_traceback) where the host would fill with one lane per
    leaf: CIGARs byte-identical to the oracle's and block-advance counts equal, BandEd + CIGAR at three bandwidths, QuickEd
    through both of its flows (bounds on the host; bounds still on the device), Hirschberg leaves after forced splits,
    ragged / N-bearing pairs"""
    batch = datagen.generate(count=200, length=5000, error=0.06, seed=191)
    pairs = list(batch.pairs())
    for G in ("2", "4", "8"):
        monkeypatch.setenv("QE_COOP_FILL_G", G)
        for bw in (10, 15, 40):
            scores, status, cig, cnt = gpu_batch(batch, algo=2, bandwidth=bw)
            tr = [oracle_cached(p, t, trace=True, algo=2, bandwidth=bw) for p, t in pairs]
            assert [(int(a), int(b), c) for a, b, c in zip(status, scores, cig)] == [(x[0], x[1], x[2]) for x in tr], (G, bw)
            assert cnt[1] == sum(x[3]["fill_block_advances"] for x in tr), (G, bw)
        rb = capi.ResidentBatch(batch)
        p = capi.make_params(algo=capi.QUICKED)
        exp = [oracle_cached(pt, tt, algo=0) for pt, tt in pairs]
        for rep in range(3):                          # first run: classic flow; then the stage-1 rule on the device
            assert rb.run(p, sync=True) >= 0
            s, st = rb.scores()
            assert [(int(a), int(b), c) for a, b, c in zip(st, s, rb.cigars())] == exp, (G, rep)
        rb.close()
    monkeypatch.setenv("QE_COOP_FILL_G", "4")
    monkeypatch.setenv("QE_SPLIT_BYTES", str(1 << 17))
    hb = datagen.generate(count=40, length=6000, error=0.07, seed=92)
    scores, status, cig, cnt = gpu_batch(hb, algo=3, bandwidth=20)
    monkeypatch.setenv("QE_COOP_FILL_G", "1")
    s1, st1, cig1, cnt1 = gpu_batch(hb, algo=3, bandwidth=20)
    assert (scores.tolist(), status.tolist(), cig, int(cnt[1])) == (s1.tolist(), st1.tolist(), cig1, int(cnt1[1]))
    monkeypatch.delenv("QE_SPLIT_BYTES")
    monkeypatch.setenv("QE_COOP_FILL_G", "2")
    mixed = mixed_batch()
    al = capi.QuickedAligner()
    al.setAlgorithm(capi.BANDED); al.setBandwidth(30)
    st, out = al.alignBatch(mixed)
    for i, (pp, tt) in enumerate(mixed):
        est, esc, ecg = oracle_cached(pp, tt, algo=2, bandwidth=30)
        in_domain = est < 0 or O.oracle().qo_exact_distance(pp, len(pp), tt, len(tt)) <= max(len(pp), len(tt)) * 30 // 100
        if in_domain:
            assert out[i] == (est, esc if est >= 0 else out[i][1], ecg if est >= 0 else out[i][2]), i


def test_reference_callers_link_and_run():
    """the reference's own harness, examples and C++ binding, compiled unmodified against the reference's
    headers and linked with libquicked_hip.so (quicked_amd/build.py: build_ref_callers): the drop-in claim.
    Mirrors tests/CMakeLists.txt:10-13 and examples/CMakeLists.txt:14-41 of the reference."""
    import os
    import subprocess
    from quicked_amd import build
    d = build.REF_CALLERS
    if not os.path.isdir(d) or not os.listdir(d):
        pytest.skip("reference callers were not built (no /root/reference in the build container)")
    run = lambda *a: subprocess.run([os.path.join(d, a[0]), *a[1:]], capture_output=True, text=True, timeout=120)
    r = run("quicked_harness", "", "")                                  # test_empty
    assert "ERROR: Tried to align an empty sequence" in r.stderr and r.returncode != 0
    r = run("quicked_harness", "GATC", "GATO", "1")                     # test_nonDNA
    assert r.returncode == 0 and "Got score: 1" in r.stdout
    r = run("quicked_harness", "ACGTACGTAC", "ACGT", "6")
    assert r.returncode == 0
    for exe in sorted(os.listdir(d)):
        if exe.startswith(("example_", "binding_")):
            r = run(exe)
            assert r.returncode == 0, (exe, r.stderr)
            assert "Score: 1" in r.stdout or "score: 1" in r.stdout.lower(), (exe, r.stdout)


def test_pyquicked_module_name():
    """`from pyquicked import ...` as in the reference's examples/bindings/basic.py"""
    import pyquicked
    al = pyquicked.QuickedAligner()
    al.setAlgorithm(pyquicked.QuickedAlgo.HIRSCHBERG)
    al.align("ACGT", "ACTT")
    assert (al.getScore(), al.getCigar()) == (1, "2M1X1M")
    assert pyquicked.BANDED == 2 and pyquicked.QUICKED_WIP == 1


def test_config_shape_properties_and_shard_invariance():
    """configs[1]/[2] shape (10 kb, 5 %) at a size the oracle cannot sweep in full: size-independent properties.
    * QuickEd's CIGAR is a valid alignment whose edit count is its score, and that score is the exact distance;
    * BandEd score-only (bandwidth 15) returns the same exact distance;
    * sharding the pairs over 'ranks' (the multi-GPU rule of bench.py) changes nothing."""
    lib = O.oracle()
    N = 3072
    whole = datagen.generate(N, 10000, 0.05, seed=0x51CED)
    s_q, st_q, cig, _ = gpu_batch(whole, algo=0)
    s_b, st_b, _, _ = gpu_batch(whole, algo=2, only_score=True, bandwidth=15)
    assert (st_q == capi.QUICKED_WIP).all() and (st_b == capi.QUICKED_WIP).all()
    assert (s_q == s_b).all()
    pairs = list(whole.pairs())
    for i in range(0, N, 7):                       # exact distance on a stride (full-height DP is the slow part)
        p, t = pairs[i]
        assert lib.qo_exact_distance(p, len(p), t, len(t)) == s_q[i]
    for i, (p, t) in enumerate(pairs):
        ops = O.rle_to_ops(cig[i])
        assert lib.qo_cigar_check(p, len(p), t, len(t), ops, len(ops)), i
        assert lib.qo_cigar_score(ops, len(ops)) == s_q[i], i
    # shard invariance: 4 'ranks' of N/4 pairs each, generated independently from (seed, first)
    per = N // 4
    for r in range(4):
        shard = datagen.generate(per, 10000, 0.05, seed=0x51CED, first=r * per)
        s, st, cg, _ = gpu_batch(shard, algo=0)
        assert (s == s_q[r * per:(r + 1) * per]).all()
        assert cg == cig[r * per:(r + 1) * per]


def test_upload_paths_agree():
    """dense pools go up as one span (pinned: straight DMA; pageable: pipelined staging), sparse pools are compacted first"""
    batch = datagen.generate(count=500, length=3000, error=0.05, seed=5)
    ref, _, _, _ = gpu_batch(batch, algo=2, only_score=True)
    pinned = capi.pinned_copy(batch)
    got, _, _, _ = gpu_batch(pinned, algo=2, only_score=True)
    capi.pinned_free(pinned)
    assert (got == ref).all()
    # a sparse layout: every pair in its own 64 KiB stride
    stride = 1 << 16
    pp = np.zeros(len(batch) * stride, dtype=np.uint8); tp = np.zeros(len(batch) * stride, dtype=np.uint8)
    poff = np.arange(len(batch), dtype=np.int64) * stride + 17
    toff = np.arange(len(batch), dtype=np.int64) * stride + 3
    for i, (p, t) in enumerate(batch.pairs()):
        pp[poff[i]:poff[i] + len(p)] = np.frombuffer(p, dtype=np.uint8)
        tp[toff[i]:toff[i] + len(t)] = np.frombuffer(t, dtype=np.uint8)
    sparse = datagen.PairBatch(pp, poff, batch.pattern_len, tp, toff, batch.text_len)
    got, _, _, _ = gpu_batch(sparse, algo=2, only_score=True)
    assert (got == ref).all()


def test_long_reads_like_the_reference_suite():
    """the reference's own long-sequence tests (tests/CMakeLists.txt:23-29): 1 Mb pairs with 10 edits, and a
    500 kb pair at ONT-like 7 % error that goes through all bound stages and five Hirschberg levels"""
    long_few = datagen.generate(2, 1000000, 10, seed=61)
    ont_like = datagen.generate(1, 500000, 0.07, seed=62, indels_num=3, indels_len=3000)
    for batch in (long_few, ont_like):
        scores, status, cig, _ = gpu_batch(batch, algo=0)
        for i, (p, t) in enumerate(batch.pairs()):
            st, sc, cg = oracle_cached(p, t, algo=0)
            assert (status[i], scores[i]) == (st, sc)
            assert cig[i] == cg


def test_many_short_pairs():
    """100 bp reads (the only workload the reference documents numbers for, tools/README.md:76): 50 k pairs"""
    batch = datagen.generate(50000, 100, 0.05, seed=63)
    scores, status, cig, _ = gpu_batch(batch, algo=0)
    pairs = list(batch.pairs())
    for i in range(0, len(pairs), 97):
        st, sc, cg = oracle_cached(*pairs[i], algo=0)
        assert (status[i], scores[i], cig[i]) == (st, sc, cg)
    lib = O.oracle()
    for c, sc in zip(cig[:2000], scores[:2000]):
        ops = O.rle_to_ops(c)
        assert lib.qo_cigar_score(ops, len(ops)) == sc


def test_pack_boundaries_forward_and_reversed(monkeypatch):
    """k_pack takes 16 bases per lane and 1 KB per wave load: sweep lengths across every boundary of that
    scheme (16-byte lane spans, 64-base rows, 1 KB spans, the ragged last lane), with non-ACGT symbols at
    the edges, forward (BandEd / WindowEd) and reversed (forced Hirschberg splits pack the reversed strings)"""
    monkeypatch.setenv("QE_SPLIT_BYTES", "4096")
    rng = np.random.default_rng(99)
    lens = sorted(set(list(range(1, 36)) + [47, 48, 49, 63, 64, 65, 79, 80, 81, 127, 128, 129, 1007, 1008, 1009,
                                            1023, 1024, 1025, 1039, 1040, 1041, 2047, 2048, 2049, 3071, 3089]))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    pairs = []
    for k, n in enumerate(lens):
        p = alpha[rng.integers(0, 4, n)].copy()
        t = p.copy()
        for pos in rng.integers(0, n, max(1, n // 25)):           # a few substitutions
            t[pos] = alpha[rng.integers(0, 4)]
        if n > 8 and k % 3 == 0:                                  # an indel: lengths differ by one
            t = np.delete(t, int(rng.integers(0, n)))
        if k % 4 == 1:
            p[0] = ord("N"); t[-1] = ord("N")
        if k % 4 == 2:
            p[-1] = ord("a"); t[0] = ord("n")
        if k % 4 == 3 and n > 17:
            p[15] = ord("R"); p[16] = ord("N"); t[n // 2] = ord("c")
        pairs.append((p.tobytes(), t.tobytes()))
    for kw in (dict(algo=2, only_score=True), dict(algo=2), dict(algo=1), dict(algo=0), dict(algo=3)):
        al = capi.QuickedAligner()
        for key, v in kw.items():
            setattr(al._params, key, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            est, esc, ecg = oracle_cached(p, t, **kw)
            assert out[i][0] == est, (kw, len(p), len(t))
            if est >= 0:
                assert out[i][1] == esc, (kw, len(p), len(t))
                assert out[i][2] == ecg, (kw, len(p), len(t))


def test_interleaved_batches_and_async_runs():
    """consecutive runs alternate between two streams / pools and overlap on the device: results of a
    synchronous run must not depend on what was queued around it (other batches, other algorithms, async runs)"""
    ba = datagen.generate(count=300, length=1500, error=0.06, seed=611)
    bb = datagen.generate(count=200, length=2500, error=0.08, seed=612)
    ra, rb = capi.ResidentBatch(ba), capi.ResidentBatch(bb)
    pa = capi.make_params(algo=capi.BANDED, only_score=True, bandwidth=15)
    pb = capi.make_params(algo=capi.QUICKED)
    pc = capi.make_params(algo=capi.HIRSCHBERG)
    for _ in range(3):
        assert ra.run(pa, sync=False) >= 0
        assert rb.run(pb, sync=False) >= 0
    assert rb.run(pc, sync=False) >= 0
    assert ra.run(pa, sync=True) >= 0           # queued behind all of the above
    sa, sta = ra.scores()
    assert rb.run(pb, sync=False) >= 0
    assert rb.run(pb, sync=True) >= 0
    sb, stb = rb.scores()
    cb = rb.cigars()
    assert ra.run(pa, sync=False) >= 0          # leave work in flight while the other batch is read again
    assert rb.run(pc, sync=True) >= 0
    sc, stc = rb.scores()
    cc = rb.cigars()
    ra.sync()
    for i, (p, t) in enumerate(ba.pairs()):
        st, s, _ = oracle_cached(p, t, algo=2, only_score=True, bandwidth=15)
        assert (sta[i], sa[i]) == (st, s), i
    for i, (p, t) in enumerate(bb.pairs()):
        st, s, cg = oracle_cached(p, t, algo=0)
        assert (stb[i], sb[i], cb[i]) == (st, s, cg), i
        st, s, cg = oracle_cached(p, t, algo=3)
        assert (stc[i], sc[i], cc[i]) == (st, s, cg), i
    ra.close(); rb.close()


def test_sam_cigar_styles_and_device_validator(monkeypatch):
    """SAM CIGAR output ("=XID" / "MID") byte-identical to the oracle's restatement of the reference
    printer, and the device-side cigar_check_alignment: in-run verdicts and caller-supplied strings"""
    monkeypatch.setenv("QE_SPLIT_BYTES", "65536")           # Hirschberg roots made of several segments
    batch = datagen.generate(count=150, length=1200, error=0.1, seed=4242)
    pairs = list(batch.pairs())
    rb = capi.ResidentBatch(batch)
    for algo in (capi.QUICKED, capi.BANDED, capi.WINDOWED, capi.HIRSCHBERG):
        kw = dict(algo=algo) if algo != capi.WINDOWED else dict(algo=algo, window_size=2)
        exp = [oracle_cached(p, t, **kw) for p, t in pairs]
        for style in (0, 1, 2):
            assert rb.configure(cigar_style=style, check=True) == 0
            assert rb.run(capi.make_params(**kw), sync=True) >= 0
            sc, st = rb.scores()
            cg = rb.cigars()
            ok = rb.check_results()
            for i, (est, esc, ecg) in enumerate(exp):
                assert (st[i], sc[i]) == (est, esc)
                want = ecg if style == 0 else O.sam_cigar(ecg, style == 1)
                assert cg[i] == want, (algo, style, i)
                assert ok[i] == 1, (algo, style, i)
    rb.configure(cigar_style=0, check=False)
    # caller-supplied strings: good ones (all three styles), broken ones, missing ones
    good = [oracle_cached(p, t, algo=0)[2] for p, t in pairs]
    assert (rb.validate(good) == 1).all()
    assert (rb.validate([O.sam_cigar(c, True) for c in good]) == 1).all()
    bad = list(good)
    bad[0] = None                                            # no string
    bad[1] = good[1] + "1M"                                  # runs past both sequences
    bad[2] = "1X" + good[2]                                  # shifted
    bad[3] = good[3].replace("X", "M", 1) if "X" in good[3] else "1M"      # a mismatch claimed as match
    bad[4] = good[4][:-1]                                    # digits without an operation
    bad[5] = "7Q"                                            # unknown operation
    bad[6] = good[7]                                         # another pair's alignment
    v = rb.validate(bad)
    assert v[0] == -1 and (v[1:7] == 0).all() and (v[7:] == 1).all()
    for i in range(1, 7):                                    # the oracle's validator agrees on the parsable ones
        if i in (4, 5):
            continue
        assert not O.cigar_is_valid(pairs[i][0], pairs[i][1], bad[i])
    rb.close()
    # ragged / empty / non-canonical input: raw-byte semantics of the validator
    mixed = mixed_batch()
    al = capi.QuickedAligner()
    st, out = al.alignBatch(mixed)
    mb = datagen.PairBatch(*_pools(mixed))
    rm = capi.ResidentBatch(mb)
    v = rm.validate([o[2] if o[0] >= 0 else None for o in out])
    for i, o in enumerate(out):
        assert v[i] == (1 if o[0] >= 0 else -1), i
    rm.close()


def _pools(pairs):
    import numpy as np
    pp = np.frombuffer(b"".join(p for p, _ in pairs) or b"\0", dtype=np.uint8).copy()
    tp = np.frombuffer(b"".join(t for _, t in pairs) or b"\0", dtype=np.uint8).copy()
    pl = np.array([len(p) for p, _ in pairs], dtype=np.int32)
    tl = np.array([len(t) for _, t in pairs], dtype=np.int32)
    po = np.concatenate([[0], np.cumsum(pl[:-1])]).astype(np.int64)
    to = np.concatenate([[0], np.cumsum(tl[:-1])]).astype(np.int64)
    return pp, po, pl, tp, to, tl


@pytest.mark.parametrize("wire", [capi.WIRE_2BIT, capi.WIRE_PLANES3])
def test_packed_wire_batches_equal_ascii_batches(wire, monkeypatch):
    """a batch created from the packed wire words gives the scores / CIGARs of the ASCII batch (and of
    the oracle) for every algorithm, including the reversed half passes of forced Hirschberg splits (planes reversed
    on the device) and, for PLANES3, sequences with N"""
    monkeypatch.setenv("QE_SPLIT_BYTES", "32768")
    rng = np.random.default_rng(31 + wire)
    base = datagen.generate(count=90, length=1500, error=0.08, seed=900 + wire)
    pairs = []
    for i, (p, t) in enumerate(base.pairs()):
        p, t = bytearray(p[: 33 + 16 * i]), bytearray(t[: 47 + 16 * i])      # ragged: every row / word remainder
        if wire == capi.WIRE_PLANES3 and i % 3 == 0:
            for k in rng.integers(0, len(p), 3): p[k] = ord("N")
            for k in rng.integers(0, len(t), 2): t[k] = ord("N")
        pairs.append((bytes(p), bytes(t)))
    batch = datagen.PairBatch(*_pools(pairs))
    ra, rp = capi.ResidentBatch(batch), capi.ResidentBatch(batch, wire=wire)
    for kw in (dict(algo=2, only_score=True, bandwidth=15), dict(algo=2), dict(algo=1), dict(algo=1, window_size=2),
               dict(algo=0), dict(algo=3)):
        prm = capi.make_params(**kw)
        for _ in range(2):                                           # twice: the second run reuses reversed planes
            assert ra.run(prm, sync=True) >= 0 and rp.run(prm, sync=True) >= 0
            sa, sta = ra.scores(); sp, stp = rp.scores()
            assert (sa == sp).all() and (sta == stp).all(), kw
            if not kw.get("only_score"):
                assert ra.cigars() == rp.cigars(), kw
        for i in (0, 1, 17, 44, 89):
            est, esc, ecg = oracle_cached(pairs[i][0], pairs[i][1], **kw)
            assert (stp[i], sp[i]) == (est, esc)
    assert rp.configure(cigar_style=1, check=True) == capi.QUICKED_UNIMPLEMENTED
    with pytest.raises(capi.QuickedException):
        rp.validate(["1M"] * len(pairs))
    ra.close(); rp.close()
    if wire == capi.WIRE_2BIT:
        with pytest.raises(capi.QuickedException):
            capi.ResidentBatch(datagen.PairBatch(*_pools([(b"ACGN", b"ACGT")])), wire=wire)


def test_large_batch_rotates_three_stream_sets():
    """batches above 131 072 pairs rotate over three stream / pool sets (qe_driver.hip run_batch): queue several
    asynchronous runs of two algorithms behind each other, then check a synchronous one against the oracle"""
    batch = datagen.generate(140000, 48, 0.08, seed=1212)
    rb = capi.ResidentBatch(batch)
    pa = capi.make_params(algo=capi.BANDED, only_score=True, bandwidth=30)
    pq = capi.make_params(algo=capi.QUICKED)
    for _ in range(4):
        assert rb.run(pa, sync=False) >= 0
    assert rb.run(pq, sync=False) >= 0
    assert rb.run(pa, sync=True) >= 0
    sa, sta = rb.scores()
    for _ in range(2):
        assert rb.run(pa, sync=False) >= 0
    assert rb.run(pq, sync=True) >= 0
    sq, stq = rb.scores()
    cq = rb.cigars()
    rb.close()
    pairs = list(batch.pairs())
    for i in list(range(0, len(pairs), 1009)) + [len(pairs) - 1]:
        st, sc, _ = oracle_cached(*pairs[i], algo=2, only_score=True, bandwidth=30)
        assert (sta[i], sa[i]) == (st, sc), i
        st, sc, cg = oracle_cached(*pairs[i], algo=0)
        assert (stq[i], sq[i], cq[i]) == (st, sc, cg), i
    assert int(sa.astype(np.int64).sum()) == int(sq.astype(np.int64).sum())      # both are the exact distance here


def test_quicked_stage3_with_zero_cutoff_terminates():
    """bandwidth 1 % of a read shorter than 100 bases is a cutoff of 0: the reference's stage-3 loop doubles it to 0
    forever (quicked.c:248-278).  Defined here and in the oracle: the doubling starts from 1 (found by the fuzz)."""
    p = b"ACGTTGCAAGTCCGATAGCTAGCTAGGATCGATCGGGATATAGCGCATTACGCATCAGC"
    t = b"TTGACCAGTGACAGGGTTTACACAGATTTCCACGCGATACCCAGTTTCACGACAGA"
    kw = dict(algo=0, bandwidth=1, window_size=2, overlap_size=1, hew_threshold=(10, 10), hew_percentage=(15, 15))
    est, esc, ecg = oracle_cached(p, t, **kw)
    al = capi.QuickedAligner()
    for k, v in kw.items():
        if k in ("hew_threshold", "hew_percentage"):
            getattr(al._params, k)[0], getattr(al._params, k)[1] = v
        else:
            setattr(al._params, k, v)
    st, out = al.alignBatch([(p, t)] * 3)
    assert all(o == (est, esc, ecg) for o in out)


def test_async_run_fetch_and_reload():
    """sync == 0 leaves the getters' data alone; quicked_batch_fetch brings exactly that run's results (scores,
    statuses, CIGARs, counters); quicked_batch_reload puts other pairs (other n, other lengths) into the same object"""
    ba = datagen.generate(count=200, length=1200, error=0.06, seed=721)
    bb = datagen.generate(count=90, length=2100, error=0.09, seed=722)
    bc = datagen.generate(count=333, length=600, error=0.04, seed=723)
    rb = capi.ResidentBatch(ba)
    p_so = capi.make_params(algo=capi.BANDED, only_score=True, bandwidth=15)
    p_q = capi.make_params(algo=capi.QUICKED)
    p_w = capi.make_params(algo=capi.WINDOWED, window_size=2)
    assert rb.run(p_q, sync=True) >= 0
    s0, st0 = rb.scores()
    c0 = rb.cigars()
    assert rb.run(p_so, sync=False) >= 0
    s1, st1 = rb.scores()
    assert (s1 == s0).all() and (st1 == st0).all() and rb.cigars() == c0          # untouched by the async run
    assert rb.fetch() >= 0
    s1, st1 = rb.scores()
    cnt = rb.counters()
    for i, (p, t) in enumerate(ba.pairs()):
        st, s, _ = oracle_cached(p, t, algo=2, only_score=True, bandwidth=15)
        assert (st1[i], s1[i]) == (st, s), i
    assert cnt[0] > 0                                                             # block-advances of the fetched run
    # two async runs in flight (the documented limit), fetched in order
    for prm, algo_kw in ((p_q, dict(algo=0)), (p_w, dict(algo=1, window_size=2))):
        assert rb.run(prm, sync=False) >= 0
        assert rb.fetch() >= 0
        s, st = rb.scores()
        cg = rb.cigars()
        for i, (p, t) in enumerate(ba.pairs()):
            assert (st[i], s[i], cg[i]) == oracle_cached(p, t, **algo_kw), (algo_kw, i)
    # reload: fewer, longer pairs, then more, shorter ones
    for nb in (bb, bc):
        assert rb.reload(nb) >= 0
        assert rb.run(p_q, sync=False) >= 0
        assert rb.fetch() >= 0
        s, st = rb.scores()
        cg = rb.cigars()
        assert len(s) == len(nb)
        for i, (p, t) in enumerate(nb.pairs()):
            assert (st[i], s[i], cg[i]) == oracle_cached(p, t, algo=0), i
    # a fetch with nothing pending is a no-op, not an error
    assert rb.fetch() >= 0
    rb.close()


def test_quicked_device_side_stage1_equals_the_classic_flow(monkeypatch):
    """QuickEd's fast path (stage-1 rule on the device, align step queued at once from an estimate, pairs that leave
    stage 1 or exceed the estimate aligned afterwards) against the oracle and against QE_QUICKED_FAST=0: same scores,
    statuses, CIGARs and work counters -- synchronously, fetched later, and with two runs of two batches in flight"""
    pairs = []
    for k, (length, err, n) in enumerate(((1500, 0.04, 60), (1500, 0.35, 9), (700, 0.10, 40), (2500, 0.22, 7))):
        pairs += list(datagen.generate(count=n, length=length, error=err, seed=910 + k).pairs())
    pairs += list(datagen.generate(count=6, length=1800, error=0.02, seed=915, indels_num=2, indels_len=150).pairs())
    order = np.random.default_rng(5).permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    mixed = datagen.PairBatch(*_pools(pairs))
    other = datagen.generate(count=130, length=1100, error=0.07, seed=917)
    expect = {id(b): [oracle_cached(p, t, algo=0) for p, t in b.pairs()] for b in (mixed, other)}
    prm = capi.make_params(algo=capi.QUICKED)

    def check(rb, b, tag):
        s, st = rb.scores()
        cg = rb.cigars()
        for i, e in enumerate(expect[id(b)]):
            assert (st[i], s[i], cg[i]) == e, (tag, i)

    monkeypatch.setenv("QE_QUICKED_FAST", "0")
    rb = capi.ResidentBatch(mixed)
    assert rb.run(prm, sync=True) >= 0
    check(rb, mixed, "classic")
    classic_counters = rb.counters()
    rb.close()
    assert classic_counters[6] > 0                                # some pairs do go on to stage 2
    for est in ("0", "40", "3"):                                  # the estimate: planned / too small for most / for all pairs
        monkeypatch.setenv("QE_QUICKED_FAST", "1")
        monkeypatch.setenv("QE_QUICKED_EST", est)
        rb = capi.ResidentBatch(mixed)
        for rep in range(3):                                      # the second run plans from the first one's bounds
            assert rb.run(prm, sync=True) >= 0
            check(rb, mixed, ("sync", est, rep))
            cnt = rb.counters()
            assert list(cnt[:5]) == list(classic_counters[:5]), (est, rep, cnt, classic_counters)
        assert rb.run(prm, sync=False) >= 0
        assert rb.fetch() >= 0
        check(rb, mixed, ("async", est))
        assert list(rb.counters()[:5]) == list(classic_counters[:5])
        # how many pairs the fetch had to align itself: those past stage 1 at least, all of them with an estimate of 3
        assert rb.deferred_pairs() >= classic_counters[6] and (est != "3" or rb.deferred_pairs() == len(mixed))
        # two batches, two runs in flight, fetched in order
        rb2 = capi.ResidentBatch(other)
        assert rb.run(prm, sync=False) >= 0
        assert rb2.run(prm, sync=False) >= 0
        assert rb.fetch() >= 0
        assert rb2.fetch() >= 0
        check(rb, mixed, ("two in flight", est))
        check(rb2, other, ("two in flight", est))
        rb.close()
        rb2.close()


def test_bench_times_the_classic_flow_when_pairs_leave_stage_1():
    """bench.py's timed loop never fetches, and an un-fetched fast QuickEd run leaves the pairs past stage 1 undone: on
    data with such pairs the bench must time the host-driven flow (and say so), on the benchmark's kind of data the fast one"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra, want in ((["--error", "0.02", "--indels-num", "2", "--indels-len", "300"], "classic"),
                        (["--error", "0.05"], "stage-1 rule on the device"), (["--error", "0.35"], "stage-1 rule on the device")):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "quicked", "--pairs", "1500", "--length", "2000",
                              "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-e2e"] + extra,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads(out.stdout.strip().splitlines()[-1])
        f = d["quicked_flow"]
        assert f["timed_flow"].startswith(want), f
        assert (f["stage2_pairs"] > 0) == (want == "classic"), f


def test_cigar_strings_stay_valid_until_free():
    """quicked.c:48-50, 357-361: every string quicked_align returned is owned by the aligner until quicked_free"""
    import ctypes as C
    lib = capi.lib()
    pairs = list(datagen.generate(count=6, length=300, error=0.08, seed=909).pairs())
    for external in (False, True):
        prm = capi.make_params(algo=capi.QUICKED)
        alloc = capi.MMAllocator()
        if external:
            prm.external_allocator = C.pointer(alloc)
        a = capi.Aligner()
        assert lib.quicked_new(C.byref(a), C.byref(prm)) == capi.QUICKED_WIP
        kept = []
        for p, t in pairs:
            assert lib.quicked_align(C.byref(a), p, len(p), t, len(t)) == capi.QUICKED_WIP
            kept.append(C.c_void_p.from_buffer(a, capi.Aligner.cigar.offset).value)
        assert len(set(kept)) == len(kept)
        for addr, (p, t) in zip(kept, pairs):
            assert C.string_at(addr).decode() == oracle_cached(p, t, algo=0)[2]
        assert lib.quicked_free(C.byref(a)) == capi.QUICKED_WIP
        assert not a.cigar


def test_timers_gain_one_sample_per_align(golden):
    """the five ABI timers (quicked.h:61-66) are ticked around the stages the reference brackets
    (quicked.c:184-193, 204-235, 240-275, 283-294), with internal timers and with external_timer = true and the
    caller's timers patched in after quicked_new (benchmark_edit.c:61-65)"""
    import ctypes as C
    lib = capi.lib()
    plain = list(datagen.generate(count=3, length=800, error=0.05, seed=31).pairs())
    heavy = list(datagen.generate(**golden["datasets"]["indel_10kb"]["gen"]).pairs())
    runs = golden["datasets"]["indel_10kb"]["runs"]
    for external in (False, True):
        prm = capi.make_params(algo=capi.QUICKED, external_timer=external)
        a = capi.Aligner()
        assert lib.quicked_new(C.byref(a), C.byref(prm)) == capi.QUICKED_WIP
        mine = [capi.ProfilerTimer() for _ in range(5)]
        if external:
            assert not a.timer and not a.timer_align
            for t in mine:
                C.memset(C.byref(t), 0, C.sizeof(t))
            a.timer, a.timer_windowed_s, a.timer_windowed_l, a.timer_banded, a.timer_align = [C.pointer(t) for t in mine]
        tm = lambda: [x.contents.time_ns.samples for x in (a.timer, a.timer_windowed_s, a.timer_windowed_l, a.timer_banded, a.timer_align)]   # noqa: E731
        for k, (p, t) in enumerate(plain):
            assert lib.quicked_align(C.byref(a), p, len(p), t, len(t)) == capi.QUICKED_WIP
            assert tm() == [k + 1, k + 1, 0, 0, k + 1]
        # pairs with large indels go through WindowEd(L) and, some of them, through the band-doubling stage
        n0 = len(plain)
        l_seen = b_seen = 0
        for k, (p, t) in enumerate(heavy[:8]):
            assert lib.quicked_align(C.byref(a), p, len(p), t, len(t)) == capi.QUICKED_WIP
            s = tm()
            assert s[0] == n0 + k + 1 and s[1] == n0 + k + 1 and s[4] == n0 + k + 1
            assert s[2] in (l_seen, l_seen + 1) and s[3] >= b_seen
            l_seen, b_seen = s[2], s[3]
        assert l_seen > 0 and b_seen > 0, (l_seen, b_seen, list(runs))
        assert a.timer.contents.time_ns.total > 0
        lib.quicked_free(C.byref(a))


@pytest.mark.parametrize("force", ["1", "0"])
def test_wave_formatter_and_wave_join_forced(force, monkeypatch):
    """the one-wave-per-alignment CIGAR formatter (long reads) against the one-lane form and the oracle: forced on
    every CIGAR-producing path, through deep Hirschberg splits (many segments per alignment, runs merging across segment
    borders), ragged / empty / non-ACGT pairs, and the SAM '=XID' style"""
    import ctypes as C
    monkeypatch.setenv("QE_FORMAT_WAVE", force)
    monkeypatch.setenv("QE_SPLIT_BYTES", str(1 << 14))
    lib = O.oracle()
    for gen in (dict(count=90, length=2500, error=0.07, seed=931), dict(count=70, length=300, error=0.25, seed=932),
                dict(count=20, length=9000, error=0.03, seed=933, indels_num=3, indels_len=400)):
        batch = datagen.generate(**gen)
        scores, status, cig, _ = gpu_batch(batch, algo=0)
        for i, (p, t) in enumerate(batch.pairs()):
            st, sc, cg, tr = oracle_cached(p, t, trace=True, algo=0)      # the bound of the reference's stages
            ops = C.create_string_buffer(len(p) + len(t) + 1)
            n = C.c_int64()
            hst = lib.qo_hirschberg(p, len(p), t, len(t), tr["bound"], 1 << 14, ops, C.byref(n), None)
            if hst != O.OK:
                continue        # a split that does not converge at this artificial threshold: partial results are unspecified
            buf = C.create_string_buffer(2 * n.value + 16)
            lib.qo_cigar_rle(ops, n.value, buf)
            assert cig[i] == buf.value.decode(), (gen, i)
    monkeypatch.delenv("QE_SPLIT_BYTES")
    for gen in (dict(count=70, length=2500, error=0.07, seed=935), dict(count=12, length=9000, error=0.03, seed=936, indels_num=3, indels_len=400)):
        batch = datagen.generate(**gen)
        for kw in (dict(algo=0), dict(algo=3, bandwidth=20), dict(algo=2, bandwidth=20), dict(algo=1, window_size=2), dict(algo=1)):
            scores, status, cig, _ = gpu_batch(batch, **kw)
            for i, (p, t) in enumerate(batch.pairs()):
                assert (status[i], scores[i], cig[i]) == oracle_cached(p, t, **kw), (gen, kw, i)
    # byte identity with the oracle (same split threshold: the reference's) on ragged / empty / non-ACGT pairs
    pairs = mixed_batch()
    for kw in (dict(algo=0), dict(algo=2), dict(algo=1), dict(algo=3)):
        al = capi.QuickedAligner()
        for k, v in kw.items():
            setattr(al._params, k, v)
        st, out = al.alignBatch(pairs)
        for i, (p, t) in enumerate(pairs):
            est, esc, ecg = oracle_cached(p, t, **kw)
            assert out[i][0] == est and (est < 0 or (out[i][1], out[i][2]) == (esc, ecg)), (kw, i)
    rb = capi.ResidentBatch(datagen.generate(count=64, length=1200, error=0.1, seed=934))
    rb.configure(cigar_style=1, check=True)
    assert rb.run(capi.make_params(algo=0), sync=True) >= 0
    got = rb.cigars()
    assert (rb.check_results() == 1).all()
    rb.close()
    for (p, t), c in zip(datagen.generate(count=64, length=1200, error=0.1, seed=934).pairs(), got):
        assert c == O.sam_cigar(oracle_cached(p, t, algo=0)[2], True)


def test_wave_per_alignment_kernel_forced(golden, monkeypatch):
    """k_banded_wave (one wavefront per alignment, rows of the band as a systolic array) forced on every score-only
    BandEd pass it is eligible for: golden vectors (incl. the geometry-dependent bandwidth-1 scores), random shapes and
    bandwidths, ragged / N / lower-case input, the block-advance counters, and QuickEd's stage 3 (band doubling)"""
    monkeypatch.setenv("QE_WAVE", "1")
    for name in ("cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len50", "len63", "len64", "len65", "len128", "len130", "len1024", "err35_2kb"):
        entry = golden["datasets"][name]
        batch = datagen.generate(**entry["gen"])
        for label, run in entry["runs"].items():
            if not (run["params"].get("algo") == 2 and run["params"].get("only_score")) and run["params"].get("algo") != 0:
                continue
            scores, status, cig, _ = gpu_batch(batch, **run["params"])
            assert status.tolist() == run["status"] and scores.tolist() == run["score"], (name, label)
    for gen in (dict(count=130, length=1000, error=0.05, seed=301), dict(count=30, length=10000, error=0.05, seed=302),
                dict(count=100, length=200, error=0.15, seed=303), dict(count=100, length=70, error=0.2, seed=304),
                dict(count=40, length=3000, error=0.3, seed=305), dict(count=70, length=1, error=0, seed=306),
                dict(count=70, length=5, error=2, seed=307), dict(count=20, length=4000, error=0.05, seed=308, indels_num=3, indels_len=300)):
        batch = datagen.generate(**gen)
        pairs = list(batch.pairs())
        for bw in (1, 4, 15, 30):
            scores, status, _, cnt = gpu_batch(batch, algo=2, only_score=True, bandwidth=bw)
            adv = 0
            for i, (p, t) in enumerate(pairs):
                st, sc, _, tr = oracle_cached(p, t, trace=True, algo=2, only_score=True, bandwidth=bw)
                assert (status[i], scores[i]) == (st, sc), (gen, bw, i)
                adv += tr["score_block_advances"]
            assert cnt[0] == adv, (gen, bw)
    pairs = mixed_batch()
    al = capi.QuickedAligner()
    al.setAlgorithm(capi.BANDED); al.setOnlyScore(True)
    st, out = al.alignBatch(pairs)
    for i, (p, t) in enumerate(pairs):
        est, esc, _ = oracle_cached(p, t, algo=2, only_score=True)
        assert out[i][0] == est and (est < 0 or out[i][1] == esc), i


def test_bench_two_ranks_end_to_end_on_one_gpu(tmp_path):
    """bench.py's N > 1 path for real: two ranks started by torch.distributed.run exactly as the driver starts them,
    sharing this box's one GPU (test hooks QE_BENCH_SHARE_GPU / QE_BENCH_BACKEND=gloo; RCCL refuses two ranks on one
    device).  The shards must add up: the 2-rank weak run sees pairs [0, 2 n) and its checksum equals a 1-rank run over
    the same 2 n pairs; the strong leg splits n pairs over the ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 3000
    common = ["--steps", "2", "--warmup", "1", "--length", "2000", "--no-cpu-baseline", "--e2e-batches", "3", "--cfg5-pairs", "5000"]
    env = dict(os.environ, QE_BENCH_SHARE_GPU="1", QE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29611", os.path.join(root, "bench.py"), "--gpus", "2", "--pairs", str(n)] + common,
                        capture_output=True, text=True, env=env, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--pairs", str(2 * n), "--no-e2e"] + common,
                        capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["value"] > 0
    assert two["score_checksum"] == one["score_checksum"]                       # shard-of-2 == whole
    assert abs(two["gcups"] / two["value"] - one["gcups"] / one["value"]) < 1e-6 * one["gcups"] / one["value"]
    assert two["strong"]["total_pairs"] == n and two["strong"]["pairs_per_gpu"] == n // 2 and two["strong"]["value"] > 0
    assert two["e2e"]["2bit_pinned"]["value"] > 0 and two["e2e"]["ascii_pinned"]["value"] > 0
    assert "cpu_baseline" not in two
    # config 5's leg: QuickEd + CIGAR, --cfg5-pairs IN TOTAL split over the ranks; the two shards add up to the whole
    c5 = two["workloads"]["cfg5"]
    assert c5["total_pairs"] == 5000 and c5["pairs_per_gpu"] == 2500 and c5["value"] > 0 and "5000 pairs" in c5["data"]
    rq = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "quicked", "--pairs", "5000", "--no-e2e",
                         "--no-strong", "--steps", "2", "--warmup", "1", "--length", "2000", "--no-cpu-baseline"],
                        capture_output=True, text=True, timeout=900)
    assert rq.returncode == 0, rq.stderr[-3000:]
    assert c5["score_checksum"] == json.loads([l for l in rq.stdout.splitlines() if l.startswith("{")][-1])["score_checksum"]


def test_bench_reduces_through_rccl_on_one_gpu():
    """The RCCL path itself, executed: bench.py as ONE rank under torch.distributed.run with QE_FORCE_DIST=1 and the default
    backend -- init_process_group("nccl", device_id=...), the all-reduces of shard.reduce_totals / count_ranks on DEVICE
    tensors, barrier + synchronize around the timed loop, destroy_process_group -- exactly what every rank of the driver's
    N > 1 runs does, on the one GPU this box has (a one-rank communicator).  The line must say so (collective.backend ==
    "nccl", device tensors), ranks_seen == 1, and the figures must equal the plain run's; config 5's leg runs in both."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "1", "--pairs", "3000", "--length", "2000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e",
              "--indel-pairs", "0", "--cfg4-pairs", "0", "--mixed-share", "0", "--cfg5-pairs", "8000"]
    env = dict(os.environ, QE_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    env.pop("QE_BENCH_BACKEND", None)
    env.pop("QE_BENCH_SHARE_GPU", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(root, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    forced = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    plain = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert forced["collective"] == {"backend": "nccl", "tensors": "device", "world_size": 1} and plain["collective"]["backend"] == "none"
    assert forced["ranks_seen"] == 1 and forced["n_gpus"] == 1 and forced["value"] > 0
    assert forced["score_checksum"] == plain["score_checksum"]
    assert forced["workloads"]["quicked"]["score_checksum"] == plain["workloads"]["quicked"]["score_checksum"]
    for line in (forced, plain):
        c5 = line["workloads"]["cfg5_shard"]
        assert c5["pairs_per_gpu"] == 1000 and c5["total_pairs"] == 1000 and c5["value"] > 0 and "configs[4]" in c5["data"]
    assert forced["workloads"]["cfg5_shard"]["score_checksum"] == plain["workloads"]["cfg5_shard"]["score_checksum"]


def test_bench_eight_ranks_dry_run_on_one_gpu():
    """configs[4]'s launch shape without the hardware: EIGHT ranks started exactly as the driver starts them
    (torch.distributed.run --nproc-per-node 8), all on this box's one GPU (the gloo / shared-device test hooks).  Every rank
    must show up in the final reduce (ranks_seen == 8), the 8 shards of 125 pairs must add up to the 1-rank run over the
    same 1 000 pairs (checksum), the strong leg must split them 8 ways, and every rank must keep >= 1 host-pack thread of
    the node's CPUs for its ASCII leg."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 125
    common = ["--steps", "2", "--warmup", "1", "--length", "2000", "--no-cpu-baseline", "--no-workloads", "--e2e-batches", "2"]
    env = dict(os.environ, QE_BENCH_SHARE_GPU="1", QE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                         "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "8", "--pairs", str(n)] + common,
                        capture_output=True, text=True, env=env, timeout=1500)
    assert r8.returncode == 0, r8.stderr[-3000:]
    eight = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][-1])
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--pairs", str(8 * n), "--no-e2e"] + common,
                        capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert eight["n_gpus"] == 8 and eight["ranks_seen"] == 8 and eight["scaling"] == "weak" and eight["value"] > 0
    assert eight["score_checksum"] == one["score_checksum"]                     # shard-of-8 == whole
    assert eight["strong"]["total_pairs"] == n and eight["strong"]["value"] > 0
    assert eight["e2e"]["ascii_hostpacked"]["host_pack_threads_per_uploader"] >= 1
    assert eight["e2e"]["ascii_hostpacked"]["value"] > 0
