"""The oracle (oracle/quicked_oracle.c) against the golden vectors captured
from the compiled reference (tests/golden/make_golden.py).  Runs anywhere."""
import hashlib

import pytest

import oracle_lib as O
from quicked_amd import datagen


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest() if s is not None else None


def test_kats(golden):
    for k in golden["kats"]:
        st, sc, cg = O.oracle_align(k["pattern"].encode(), k["text"].encode(), **k["params"])
        assert (st, cg) == (k["status"], k["cigar"]), k
        if st >= 0:
            assert sc == k["score"], k


def test_status_messages():
    lib = O.oracle()
    # quicked.c:387-400; tests/CMakeLists.txt:11 greps the EMPTY_SEQUENCE one
    assert lib.qo_status_msg(O.EMPTY_SEQUENCE) == b"ERROR: Tried to align an empty sequence\n"
    assert lib.qo_status_msg(O.UNKNOWN_ALGO) == b"ERROR: Unknown algorithm selection\n"
    assert lib.qo_status_msg(O.WIP) == b"QuickEd finished without errors.\n"
    assert lib.qo_status_msg(O.OK) == b"QuickEd finished without errors.\n"


DATASETS = ["cfg1_1kb_5pct", "cfg2_10kb_5pct", "indel_10kb", "len50", "len63", "len64", "len65", "len128",
            "len130", "len1024", "err35_2kb", "cfg4_100kb_10pct", "cfg4_indel_100kb"]


@pytest.mark.parametrize("name", DATASETS)
def test_dataset(golden, name):
    entry = golden["datasets"][name]
    batch = datagen.generate(**entry["gen"])
    pairs = list(batch.pairs())
    assert [len(p) for p, _ in pairs] == entry["plen"], "generator drifted from the fixture"
    for label, run in entry["runs"].items():
        for i, (p, t) in enumerate(pairs):
            st, sc, cg = O.oracle_align(p, t, **run["params"])
            assert st == run["status"][i], (name, label, i)
            assert sc == run["score"][i], (name, label, i)
            if "cigar_sha256" in run:
                assert sha(cg) == run["cigar_sha256"][i], (name, label, i)
                assert O.cigar_is_valid(p, t, cg)


def test_exact_distance_agrees_with_golden_quicked(golden):
    entry = golden["datasets"]["cfg1_1kb_5pct"]
    batch = datagen.generate(**entry["gen"])
    lib = O.oracle()
    for i, (p, t) in enumerate(batch.pairs()):
        assert lib.qo_exact_distance(p, len(p), t, len(t)) == entry["runs"]["quicked"]["score"][i]


def test_sam_cigar_known_answers():
    """SAM CIGAR restatement (cigar.c:194-240, 504-529): X folds into M before merging unless mismatches are shown"""
    assert O.sam_cigar("2M1X1M", True) == "2=1X1="
    assert O.sam_cigar("2M1X1M", False) == "4M"
    assert O.sam_cigar("4M6D", True) == "4=6D" and O.sam_cigar("4M6D", False) == "4M6D"
    assert O.sam_cigar("1X3M2I1X1M1D", False) == "1X3M2I2M1D"      # the reference does not fold the very first op
    assert O.sam_cigar("3X2M", False) == "1X4M"
    assert O.sam_cigar("2M1X3M2I1X1M1D", False) == "6M2I2M1D"
    assert O.sam_cigar("1X3M2I1X1M1D", True) == "1X3=2I1X1=1D"


def test_stage3_zero_cutoff_is_defined():
    """the oracle leaves the reference's never-ending doubling of a zero cutoff (quicked.c:248-278) after one step"""
    p = b"ACGTTGCAAGTCCGATAGCTAGCTAGGATCGATCGGGATATAGCGCATTACGCATCAGC"
    t = b"TTGACCAGTGACAGGGTTTACACAGATTTCCACGCGATACCCAGTTTCACGACAGA"
    st, sc, cg = O.oracle_align(p, t, trace=True, algo=0, bandwidth=1, window_size=2, overlap_size=1,
                                hew_threshold=(10, 10), hew_percentage=(15, 15))[:3]
    assert st == 1 and sc == 31 and O.cigar_is_valid(p, t, cg)


def test_oracle_and_generator_are_clean_under_asan_ubsan():
    """`make -C oracle asan`: the oracle restatement + the seeded generator under AddressSanitizer / UBSan on the CPU
    (the reference's build has the same switches, CMakeLists.txt:43-49; GPU sanitizers are unavailable on the pool)"""
    import os
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "all checks passed under ASAN + UBSAN" in r.stdout
