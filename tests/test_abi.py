"""CPU-side checks of the drop-in boundary: the library loads, exports every
symbol include/*.h declares, struct layouts match the reference's,
and the calls that need no GPU behave like the reference's."""
import ctypes as C
import os
import re

import pytest

from quicked_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(build.HIP_LIB):
        build.build_hip()
    return capi.lib()


def declared_functions():
    names = set()
    for h in ("quicked.h", "quicked_batch.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(quicked_[a-z_]+)\s*\(", src))
    return names


def test_exports_match_headers(lib):
    decl = declared_functions()
    assert decl == set(capi.EXPORTS)
    for name in decl:
        assert hasattr(lib, name), name


def test_struct_layouts_match_reference():
    # x86-64 SysV layout probed on the reference
    assert C.sizeof(capi.Params) == 48
    assert [getattr(capi.Params, f).offset for f in ("algo", "bandwidth", "window_size", "overlap_size", "hew_threshold",
                                                      "hew_percentage", "only_score", "force_scalar", "external_timer",
                                                      "external_allocator")] == [0, 4, 8, 12, 16, 24, 32, 33, 34, 40]
    assert C.sizeof(capi.Aligner) == 72
    assert [getattr(capi.Aligner, f).offset for f in ("params", "mm_allocator", "cigar", "score", "timer",
                                                       "timer_windowed_s", "timer_windowed_l", "timer_banded",
                                                       "timer_align")] == [0, 8, 16, 24, 32, 40, 48, 56, 64]
    assert C.sizeof(capi.ProfilerTimer) == 88
    assert C.sizeof(capi.MMAllocator) == 56


def test_header_compiles_as_c_and_cpp(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "quicked.h"\n#include "quicked_batch.h"\n'
                   '_Static_assert(sizeof(quicked_params_t) == 48, "params");\n'
                   '_Static_assert(sizeof(quicked_aligner_t) == 72, "aligner");\n'
                   '_Static_assert(sizeof(profiler_timer_t) == 88, "timer");\n'
                   '_Static_assert(sizeof(mm_allocator_t) == 56, "alloc");\nint main(void){return QUICKED_WIP-1;}\n')
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "quicked.h"\n#include "quicked_batch.h"\nint main(){return 0;}\n')
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(cpp),
                    "-o", str(tmp_path / "t2.o")], check=True)


def test_default_params_and_messages(lib):
    p = lib.quicked_default_params()      # quicked.c:308-321
    assert (p.algo, p.bandwidth, p.window_size, p.overlap_size) == (0, 15, 9, 1)
    assert list(p.hew_threshold) == [40, 40] and list(p.hew_percentage) == [15, 15]
    assert (p.only_score, p.force_scalar, p.external_timer) == (False, False, False)
    assert not p.external_allocator
    assert lib.quicked_status_msg(capi.QUICKED_EMPTY_SEQUENCE) == b"ERROR: Tried to align an empty sequence\n"
    assert lib.quicked_status_msg(capi.QUICKED_WIP) == b"QuickEd finished without errors.\n"
    assert lib.quicked_check_error(-4) and not lib.quicked_check_error(1) and not lib.quicked_check_error(0)


def test_new_free_and_early_errors_need_no_gpu(lib):
    p = lib.quicked_default_params()
    a = capi.Aligner()
    assert lib.quicked_new(C.byref(a), C.byref(p)) == capi.QUICKED_WIP          # quicked.c:351
    assert a.score == -1 and a.cigar is None and bool(a.timer) and bool(a.mm_allocator)
    assert C.addressof(a.params.contents) == C.addressof(p)                       # caller's object, not a copy
    assert lib.quicked_align(C.byref(a), b"", 0, b"", 0) == capi.QUICKED_EMPTY_SEQUENCE   # quicked.c:411-414
    assert lib.quicked_align(C.byref(a), b"ACGT", 4, b"", 0) == capi.QUICKED_EMPTY_SEQUENCE
    p.algo = 7
    assert lib.quicked_align(C.byref(a), b"ACGT", 4, b"ACGT", 4) == capi.QUICKED_UNKNOWN_ALGO  # quicked.c:432-433
    assert lib.quicked_free(C.byref(a)) == capi.QUICKED_WIP                       # quicked.c:377
    # external timers: the five pointers are the caller's to patch (benchmark_edit.c:61-65)
    p = lib.quicked_default_params()
    p.external_timer = True
    a = capi.Aligner()
    assert lib.quicked_new(C.byref(a), C.byref(p)) == capi.QUICKED_WIP
    assert not a.timer and not a.timer_align
    assert lib.quicked_free(C.byref(a)) == capi.QUICKED_WIP


def test_wire_serializer_known_answers():
    """quicked_wire_pack is host code: 2-bit codes A0 C1 G2 T3 (dna_text.c:41-46), 32 bases per word; PLANES3 =
    {code bit 0, code bit 1, not-ACGT} per 64 bases; unrepresentable symbols are refused"""
    import ctypes as C
    import numpy as np
    from quicked_amd import capi
    L = capi.lib()
    assert L.quicked_wire_words(0, 2) == 0 and L.quicked_wire_words(32, 2) == 1 and L.quicked_wire_words(33, 2) == 2
    assert L.quicked_wire_words(64, 3) == 3 and L.quicked_wire_words(65, 3) == 6 and L.quicked_wire_words(5, 7) == -1
    out = np.zeros(4, dtype=np.uint64)
    assert L.quicked_wire_pack(b"ACGTTGCA", 8, 2, out.ctypes.data) == 0
    assert int(out[0]) == sum(c << (2 * i) for i, c in enumerate([0, 1, 2, 3, 3, 2, 1, 0]))
    seq = b"ACGTN" + b"T" * 60
    out = np.zeros(6, dtype=np.uint64)
    assert L.quicked_wire_pack(seq, len(seq), 3, out.ctypes.data) == 0
    codes = [{65: 0, 67: 1, 71: 2, 84: 3, 78: 4}[c] for c in seq]
    for r in range(2):
        row = codes[64 * r: 64 * r + 64]
        assert int(out[3 * r]) == sum(1 << i for i, c in enumerate(row) if c < 4 and c & 1)
        assert int(out[3 * r + 1]) == sum(1 << i for i, c in enumerate(row) if c < 4 and c & 2)
        assert int(out[3 * r + 2]) == sum(1 << i for i, c in enumerate(row) if c == 4)
    assert L.quicked_wire_pack(b"ACGN", 4, 2, out.ctypes.data) < 0      # N has no 2-bit code
    assert L.quicked_wire_pack(b"ACgT", 4, 3, out.ctypes.data) < 0      # lower case: raw != encoded compare
    assert L.quicked_wire_pack(b"ACRT", 4, 3, out.ctypes.data) < 0      # IUPAC


def test_wire_pack_pool_equals_the_per_sequence_serializer():
    """quicked_wire_pack_pool (SIMD, multi-threaded; no GPU involved) writes the words quicked_wire_pack writes, on ragged
    lengths around every block border, for both wire formats and every kernel this CPU has; unrepresentable symbols are
    refused with the first offending sequence"""
    import ctypes as C
    import numpy as np
    from quicked_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(7)
    lens = [0, 1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 1000, 4097, 10000] + \
           [int(x) for x in rng.integers(1, 3000, 60)]
    best = L.quicked_wire_pack_isa(-1)
    assert best in (0, 1, 2)
    try:
        for wire, alphabet in ((capi.WIRE_2BIT, b"ACGT"), (capi.WIRE_PLANES3, b"ACGTN")):
            seqs = [bytes(rng.choice(np.frombuffer(alphabet, np.uint8), n).tolist()) for n in lens]
            gap = [int(x) for x in rng.integers(0, 9, len(seqs))]          # sequences need not lie back to back
            pool = bytearray()
            off = []
            for s, g in zip(seqs, gap):
                pool += b"#" * g
                off.append(len(pool))
                pool += s
            pool = np.frombuffer(bytes(pool) + b"#" * 64, np.uint8)
            off = np.array(off, np.int64)
            ln = np.array(lens, np.int32)
            ref_words = []
            for s in seqs:
                w = np.zeros(max(L.quicked_wire_words(len(s), wire), 1), np.uint64)
                assert L.quicked_wire_pack(s, len(s), wire, w.ctypes.data) == 0
                ref_words.append(w[:L.quicked_wire_words(len(s), wire)])
            for isa in range(best + 1):
                assert L.quicked_wire_pack_isa(isa) == isa
                for threads in (1, 3, 0):
                    words, woff = capi.wire_pack_pool(pool, off, ln, wire, threads=threads)
                    for i, w in enumerate(ref_words):
                        got = words[int(woff[i]): int(woff[i]) + len(w)]
                        assert (got == w).all(), (wire, isa, threads, lens[i])
                # a symbol the wire cannot carry: lower case, IUPAC, N in the 2-bit form -- at a block border and in a tail
                for pos_seq, pos, sym in ((20, 5000, b"a"), (18, 100, b"R"), (12, 64, b"n")) + (((19, 4096, b"N"),) if wire == 2 else ()):
                    bad = bytearray(pool.tobytes())
                    bad[int(off[pos_seq]) + pos] = sym[0]
                    badp = np.frombuffer(bytes(bad), np.uint8)
                    woff2, total = capi.wire_offsets(ln, wire)
                    out = np.zeros(total + 1, np.uint64)
                    which = C.c_int64(-7)
                    st = L.quicked_wire_pack_pool(len(ln), badp.ctypes.data, off.ctypes.data, ln.ctypes.data, wire, out.ctypes.data,
                                                  woff2.ctypes.data, 2, C.byref(which))
                    assert st < 0 and which.value == pos_seq, (wire, isa, pos_seq, which.value)
            assert L.quicked_wire_pack_isa(best + 1) == -1 or best == 2
    finally:
        L.quicked_wire_pack_isa(-1)




def _newest_bench_line():
    import glob
    import json
    import re
    def tag(f):      # (round, milestone): r05_z < r06_a < r06_b2 < r10_a; a name that does not parse sorts first instead of failing
        m = re.match(r"r(\d+)_([A-Za-z]*)(\d*)_bench_line", os.path.basename(f))
        return (int(m.group(1)), m.group(2).lower(), int(m.group(3) or 0)) if m else (-1, "", 0)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_bench_line.json")), key=tag)
    assert files, "no committed bench line"
    f = files[-1]
    return f, json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])


def test_newest_committed_bench_line_follows_the_contract():
    """the line bench.py printed on the MI355X for the newest milestone under profiles/ (the driver's command): the contract
    fields, roofline + cpu_baseline, the flat end-to-end / single-batch keys of round 5, every workload object with its own
    baseline, the 12.5 k-pair share incl. the mixed leg's early-finish statistics"""
    f, d = _newest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "ranks_seen", "e2e", "workloads", "strong_share",
              "value_definition", "kernel_value", "e2e_value", "e2e_ascii_link_value", "single_batch_value"):
        assert k in d, (f, k)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    # round 6: which collective backend reduced the line's figures ("none" for a plain N = 1 run), config 5's share of one GPU,
    # the loop's measured rate next to the nominal instruction-mix bound
    assert d["collective"]["backend"] in ("none", "nccl", "gloo")
    c5 = d["workloads"]["cfg5_shard"]
    assert c5["pairs_per_gpu"] == 125000 and c5["value"] > 0 and "configs[4]" in c5["data"]
    assert 0 < d["valu"]["aggregate_frac_of_measured_loop"] < 1.05
    assert "workload" in d["config"] and "model" not in d["config"] and "DEVICE-RESIDENT" in d["config"]["workload"]
    assert abs(d["value"] - d["config"]["pairs_per_gpu"] / d["ms_per_step"] * 1e3) / d["value"] < 1e-6
    assert d["kernel_value"] == d["value"] and d["e2e_value"] == d["e2e"]["ascii_hostpacked"]["value"]
    assert d["value"] > d["single_batch_value"] > 0 and d["e2e"]["2bit_pinned"]["value"] > d["e2e_ascii_link_value"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "frac_of_measured_copy"):
        assert k in r, (f, k)
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < r["frac_of_measured_copy"] <= 1
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, (f, k)
    assert c["kind"] in ("reference", "port") and c["gpu_scores_identical_on_sample"] is True
    w = d["workloads"]
    for wl in ("quicked", "cfg4", "quicked_indels", "quicked_mixed"):
        assert w[wl]["value"] > 0 and w[wl]["cpu_baseline"]["value"] > 0, (f, wl)
    assert w["quicked"]["score_checksum"] == d["score_checksum"]            # BandEd (bandwidth 15) and QuickEd agree on every distance
    # round 6, late: QuickEd with only_score takes its scores from one score-only pass over the fill's cells
    qs = w["quicked_score"]
    assert qs["scores_equal_workloads_quicked"] is True and qs["traceback_steps"] == 0 and qs["value"] > w["quicked"]["value"]
    assert w["quicked_indels"]["quicked_flow"]["stage2_pairs"] > 0 and w["quicked_indels"]["quicked_flow"]["stage3_pairs"] > 0
    s = d["strong_share"]
    assert s["pairs_per_gpu"] == 12500
    for wl in ("banded_score", "quicked"):
        assert s[wl]["value"] > s[wl]["single_batch_value"] > 0 and s[wl]["runs_in_flight"] >= 3
    assert s["quicked_mixed"]["value"] > 0 and s["quicked_mixed"]["early_finish_flows"]["merged_flows"] >= 0
    assert s["quicked_score"]["scores_equal_workloads_quicked"] is True and s["quicked_score"]["value"] > s["quicked"]["value"]
    assert s["quicked_score"]["single_batch_latency_ms"] < s["quicked"]["single_batch_latency_ms"]


def test_readme_numbers_are_generated_from_the_committed_bench_line():
    """README.md's measured paragraph is tools/readme_numbers.py's output for the newest profiles/<tag>_bench_line.json"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("readme_numbers", os.path.join(ROOT, "tools", "readme_numbers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(ROOT, "README.md")).read()
    held = text[text.index(mod.BEGIN) + len(mod.BEGIN):text.index(mod.END)].strip()
    assert held == mod.paragraph().strip()


def test_committed_kernel_resource_table_has_no_scratch():
    """profiles/<newest tag>_kernel_resources.txt (tools/kernel_resources.sh): every kernel of the build it describes fits
    its registers -- incl. round 5's cooperative forms"""
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_kernel_resources.txt"))
    assert files
    lines = [l for l in open(os.path.join(ROOT, "profiles", files[-1])).read().splitlines() if l.strip()]
    assert len(lines) >= 20
    for l in lines:
        assert "ScratchSize [bytes/lane]: 0" in l, l
    for name in ("k_banded<true>", "k_windowed_cp", "k_windowed_quad", "k_banded_sys<4, true>", "k_banded_sys<6, false>", "k_traceback_sys<4>"):
        assert any(name in l for l in lines), name
