"""ctypes fronts for the ORACLE (oracle/liboracle.so) and, when it was built in
the container that has /root/reference, the compiled reference itself
(oracle/_ref/libquicked_ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libquicked_ref.so")

QUICKED, WINDOWED, BANDED, HIRSCHBERG = 0, 1, 2, 3
OK, ERROR, FAIL_NON_CONVERGENCE, UNKNOWN_ALGO, EMPTY_SEQUENCE, UNIMPLEMENTED, WIP = 0, -1, -2, -3, -4, -10, 1


class QoParams(C.Structure):
    _fields_ = [("algo", C.c_int32), ("bandwidth", C.c_uint32), ("window_size", C.c_uint32),
                ("overlap_size", C.c_uint32), ("hew_threshold", C.c_uint32 * 2),
                ("hew_percentage", C.c_uint32 * 2), ("only_score", C.c_int32), ("force_scalar", C.c_int32)]


class QoTrace(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("ws_score", "ws_hew", "wl_score", "wl_hew", "wl_fwd_score", "wl_rev_score")] + \
               [("stage", C.c_int32), ("banded_calls", C.c_int32)] + \
               [(n, C.c_int64) for n in ("bound", "hirschberg_splits", "leaves", "score_block_advances",
                                         "fill_block_advances", "window_block_steps", "traceback_steps")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_oracle = None


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)


def oracle():
    global _oracle
    if _oracle is None:
        src = os.path.join(ORACLE_DIR, "quicked_oracle.c")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.qo_default_params.argtypes = [C.POINTER(QoParams)]
        lib.qo_status_msg.restype = C.c_char_p
        lib.qo_status_msg.argtypes = [C.c_int]
        lib.qo_align.restype = C.c_int
        lib.qo_align.argtypes = [C.POINTER(QoParams), C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(QoTrace)]
        lib.qo_free.argtypes = [C.c_void_p]
        lib.qo_banded_score.restype = C.c_int64
        lib.qo_banded_score.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, C.c_int,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.qo_banded_align.restype = C.c_int64
        lib.qo_banded_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, C.c_char_p,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.qo_windowed.restype = C.c_int
        lib.qo_windowed.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_char_p,
                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.qo_hirschberg.restype = C.c_int
        lib.qo_hirschberg.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int64, C.c_uint64, C.c_char_p,
                                      C.POINTER(C.c_int64), C.POINTER(QoTrace)]
        lib.qo_cigar_rle.restype = C.c_int64
        lib.qo_cigar_rle.argtypes = [C.c_char_p, C.c_int64, C.c_char_p]
        lib.qo_cigar_score.restype = C.c_int64
        lib.qo_cigar_score.argtypes = [C.c_char_p, C.c_int64]
        lib.qo_cigar_check.restype = C.c_int
        lib.qo_cigar_check.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int64]
        lib.qo_cigar_sam.restype = C.c_int64
        lib.qo_cigar_sam.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_char_p]
        lib.qo_rle_to_ops.restype = C.c_int64
        lib.qo_rle_to_ops.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
        lib.qo_exact_distance.restype = C.c_int64
        lib.qo_exact_distance.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        _oracle = lib
    return _oracle


def default_params(**kw):
    p = QoParams()
    oracle().qo_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("hew_threshold", "hew_percentage"):
            getattr(p, k)[0], getattr(p, k)[1] = v
        else:
            setattr(p, k, v)
    return p


def oracle_align(pattern, text, trace=False, **kw):
    """-> (status, score, cigar or None[, trace dict])"""
    lib = oracle()
    p = default_params(**kw)
    score = C.c_int(-1)
    cig = C.c_void_p()
    tr = QoTrace()
    st = lib.qo_align(C.byref(p), pattern, len(pattern), text, len(text), C.byref(score), C.byref(cig), C.byref(tr))
    cigar = None
    if cig.value:
        cigar = C.string_at(cig.value).decode()
        lib.qo_free(cig)
    if trace:
        return st, score.value, cigar, tr.as_dict()
    return st, score.value, cigar


def rle_to_ops(rle):
    lib = oracle()
    n = sum(int(x) for x in __import__("re").findall(r"(\d+)[MXID]", rle))
    buf = C.create_string_buffer(n + 1)
    got = lib.qo_rle_to_ops(rle.encode(), buf, n)
    assert got == n
    return buf.raw[:n]


def sam_cigar(rle, show_mismatches):
    """oracle: the reference's RLE string -> SAM CIGAR ("=XID" or, with X folded into M, "MID")"""
    if rle is None:
        return None
    ops = rle_to_ops(rle)
    buf = C.create_string_buffer(2 * len(ops) + 16)
    oracle().qo_cigar_sam(ops, len(ops), 1 if show_mismatches else 0, buf)
    return buf.value.decode()


class RefCigar(C.Structure):
    """cigar_t, quicked_utils/include/cigar.h:33-46"""
    _fields_ = [("operations", C.c_char_p), ("cigar_buffer", C.POINTER(C.c_uint32)), ("cigar_length", C.c_int),
                ("max_operations", C.c_int), ("begin_offset", C.c_int), ("end_offset", C.c_int),
                ("score", C.c_int), ("end_v", C.c_int), ("end_h", C.c_int)]


def ref_sam_cigar(ops, show_mismatches):
    """the compiled reference's cigar_sprint_SAM_CIGAR (cigar.c:504-529) over an operations string"""
    lib = ref()
    lib.cigar_sprint_SAM_CIGAR.restype = C.c_int
    lib.cigar_sprint_SAM_CIGAR.argtypes = [C.c_char_p, C.c_int, C.POINTER(RefCigar), C.c_bool]
    n = len(ops)
    cbuf = (C.c_uint32 * (n + 1))()
    cg = RefCigar(ops, cbuf, 0, n, 0, n, 0, -1, -1)
    out = C.create_string_buffer(12 * n + 16)     # the reference passes buf_size to every snprintf: be generous
    lib.cigar_sprint_SAM_CIGAR(out, 12 * n + 16, C.byref(cg), bool(show_mismatches))
    return out.value.decode()


def cigar_is_valid(pattern, text, rle):
    ops = rle_to_ops(rle)
    return bool(oracle().qo_cigar_check(pattern, len(pattern), text, len(text), ops, len(ops)))


# --------------------------------------------------------------------------- #
# the compiled reference (only where oracle/_ref was built)
# --------------------------------------------------------------------------- #
class RefParams(C.Structure):
    """quicked_params_t, quicked/quicked.h:43-54"""
    _fields_ = [("algo", C.c_int), ("bandwidth", C.c_uint), ("window_size", C.c_uint), ("overlap_size", C.c_uint),
                ("hew_threshold", C.c_uint * 2), ("hew_percentage", C.c_uint * 2),
                ("only_score", C.c_bool), ("force_scalar", C.c_bool), ("external_timer", C.c_bool),
                ("external_allocator", C.c_void_p)]


class RefAligner(C.Structure):
    """quicked_aligner_t, quicked/quicked.h:56-67"""
    _fields_ = [("params", C.POINTER(RefParams)), ("mm_allocator", C.c_void_p), ("cigar", C.c_char_p),
                ("score", C.c_int), ("timer", C.c_void_p), ("timer_windowed_s", C.c_void_p),
                ("timer_windowed_l", C.c_void_p), ("timer_banded", C.c_void_p), ("timer_align", C.c_void_p)]


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.quicked_default_params.restype = RefParams
        lib.quicked_new.argtypes = [C.POINTER(RefAligner), C.POINTER(RefParams)]
        lib.quicked_align.argtypes = [C.POINTER(RefAligner), C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        lib.quicked_free.argtypes = [C.POINTER(RefAligner)]
        lib.quicked_status_msg.restype = C.c_char_p
        _ref = lib
    return _ref


def ref_align(pattern, text, **kw):
    """Runs the compiled reference through its public C-ABI -> (status, score, cigar)."""
    lib = ref()
    p = lib.quicked_default_params()
    for k, v in kw.items():
        if k in ("hew_threshold", "hew_percentage"):
            getattr(p, k)[0], getattr(p, k)[1] = v
        else:
            setattr(p, k, v)
    a = RefAligner()
    st = lib.quicked_new(C.byref(a), C.byref(p))
    assert st >= 0
    # NUL-terminated copies: the reference's SSE window kernel reads text[tlen]
    st = lib.quicked_align(C.byref(a), pattern + b"\0", len(pattern), text + b"\0", len(text))
    score = a.score
    cigar = a.cigar.decode() if a.cigar else None
    lib.quicked_free(C.byref(a))
    return st, score, cigar


# --------------------------------------------------------------------------- #
# edlib 1.2.6 (the reference's vendored copy, tools/align_benchmark/external/edlib) compiled into oracle/_ref: the
# independent exact-distance opinion the reference's own `--check score` uses (benchmark_check.c)
# --------------------------------------------------------------------------- #
EDLIB_SO = os.path.join(ORACLE_DIR, "_ref", "libedlib_ref.so")


class EdlibAlignConfig(C.Structure):
    """edlib.h: EdlibAlignConfig {k, mode, task, additionalEqualities, additionalEqualitiesLength}"""
    _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int), ("additionalEqualities", C.c_void_p),
                ("additionalEqualitiesLength", C.c_int)]


class EdlibAlignResult(C.Structure):
    _fields_ = [("status", C.c_int), ("editDistance", C.c_int), ("endLocations", C.POINTER(C.c_int)),
                ("startLocations", C.POINTER(C.c_int)), ("numLocations", C.c_int), ("alignment", C.POINTER(C.c_ubyte)),
                ("alignmentLength", C.c_int), ("alphabetLength", C.c_int)]


_edlib = None


def have_edlib():
    return os.path.exists(EDLIB_SO)


def edlib_distance(pattern, text):
    """global (NW) edit distance by edlib; bytes compare as bytes (no N wildcard, no case folding)"""
    global _edlib
    if _edlib is None:
        lib = C.CDLL(EDLIB_SO)
        lib.edlibAlign.restype = EdlibAlignResult
        lib.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, EdlibAlignConfig]
        lib.edlibFreeAlignResult.argtypes = [EdlibAlignResult]
        lib.edlibFreeAlignResult.restype = None
        _edlib = lib
    cfg = EdlibAlignConfig(-1, 0, 0, None, 0)          # k = -1 (no bound), EDLIB_MODE_NW, EDLIB_TASK_DISTANCE
    r = _edlib.edlibAlign(pattern, len(pattern), text, len(text), cfg)
    d = r.editDistance
    assert r.status == 0
    _edlib.edlibFreeAlignResult(r)
    return d
