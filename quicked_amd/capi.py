"""ctypes binding of the C-ABI in include/quicked.h + include/quicked_batch.h.

Loads ``quicked_amd/libquicked_hip.so`` (built in-tree by ``build.py``) and
fails loudly when it is missing: there is no Python or CPU fallback path.
"""
import ctypes as C
import os

import numpy as np

from . import build

QUICKED, WINDOWED, BANDED, HIRSCHBERG = 0, 1, 2, 3
QUICKED_OK, QUICKED_ERROR, QUICKED_FAIL_NON_CONVERGENCE = 0, -1, -2
QUICKED_UNKNOWN_ALGO, QUICKED_EMPTY_SEQUENCE, QUICKED_UNIMPLEMENTED, QUICKED_WIP = -3, -4, -10, 1


class ProfilerCounter(C.Structure):
    _fields_ = [("total", C.c_uint64), ("samples", C.c_uint64), ("min", C.c_uint64), ("max", C.c_uint64),
                ("m_oldM", C.c_double), ("m_newM", C.c_double), ("m_oldS", C.c_double), ("m_newS", C.c_double)]


class Timespec(C.Structure):
    _fields_ = [("tv_sec", C.c_long), ("tv_nsec", C.c_long)]


class ProfilerTimer(C.Structure):
    _fields_ = [("begin_timer", Timespec), ("time_ns", ProfilerCounter), ("accumulated", C.c_uint64)]


class MMAllocator(C.Structure):
    _fields_ = [("request_ticker", C.c_uint64), ("segment_size", C.c_uint64), ("segments", C.c_void_p),
                ("segments_free", C.c_void_p), ("current_segment_idx", C.c_uint64),
                ("malloc_requests", C.c_void_p), ("malloc_requests_freed", C.c_uint64)]


class Params(C.Structure):
    """quicked_params_t (include/quicked.h; reference quicked/quicked.h:43-54)"""
    _fields_ = [("algo", C.c_int), ("bandwidth", C.c_uint), ("window_size", C.c_uint), ("overlap_size", C.c_uint),
                ("hew_threshold", C.c_uint * 2), ("hew_percentage", C.c_uint * 2),
                ("only_score", C.c_bool), ("force_scalar", C.c_bool), ("external_timer", C.c_bool),
                ("external_allocator", C.POINTER(MMAllocator))]


class Aligner(C.Structure):
    """quicked_aligner_t (include/quicked.h; reference quicked/quicked.h:56-67)"""
    _fields_ = [("params", C.POINTER(Params)), ("mm_allocator", C.POINTER(MMAllocator)), ("cigar", C.c_char_p),
                ("score", C.c_int), ("timer", C.POINTER(ProfilerTimer)),
                ("timer_windowed_s", C.POINTER(ProfilerTimer)), ("timer_windowed_l", C.POINTER(ProfilerTimer)),
                ("timer_banded", C.POINTER(ProfilerTimer)), ("timer_align", C.POINTER(ProfilerTimer))]


EXPORTS = ["quicked_check_error", "quicked_status_msg", "quicked_default_params", "quicked_new", "quicked_free",
           "quicked_align", "quicked_set_device", "quicked_device_count", "quicked_align_batch", "quicked_batch_create",
           "quicked_batch_destroy", "quicked_batch_run", "quicked_batch_sync", "quicked_batch_scores",
           "quicked_batch_cigar_bytes", "quicked_batch_cigars", "quicked_batch_counters",
           "quicked_batch_kernel_time", "quicked_batch_kernel_times", "quicked_host_alloc", "quicked_host_free",
           "quicked_batch_configure", "quicked_batch_check_results", "quicked_batch_validate",
           "quicked_wire_words", "quicked_wire_pack", "quicked_batch_create_packed",
           "quicked_batch_reload", "quicked_batch_reload_packed", "quicked_batch_fetch", "quicked_pool_stats", "quicked_batch_cigar_view",
           "quicked_batch_deferred_pairs", "quicked_wire_pack_pool", "quicked_wire_offsets", "quicked_wire_pack_isa", "quicked_pool_trim", "quicked_early_finish_stats",
           "quicked_debug_reload_env"]

_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("QUICKED_HIP_LIB") or build.HIP_LIB       # A/B runs of two builds: QUICKED_HIP_LIB=<other .so>
    if not os.path.exists(path):
        try:
            build.build_hip()            # hipcc --offload-arch=gfx950, in-tree
        except Exception as e:           # noqa: BLE001
            raise RuntimeError(f"{path} is missing and could not be built ({e}): run "
                               "`python -c 'import __graft_entry__ as g; g.build()'`; there is no fallback path") from e
    # The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that
    # share one serialise; a thread's runs rotate over up to 12 stream sets.  The runtime reads the variable when it
    # initialises, so the embedding application sets it before its first HIP call; this binding is one.
    # 20: with 24 queues in existence the hardware oversubscribes and small-batch streams lose a third of their rate.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
    L = C.CDLL(path)
    L.quicked_check_error.restype = C.c_bool
    L.quicked_check_error.argtypes = [C.c_int]
    L.quicked_status_msg.restype = C.c_char_p
    L.quicked_status_msg.argtypes = [C.c_int]
    L.quicked_default_params.restype = Params
    L.quicked_default_params.argtypes = []
    L.quicked_new.argtypes = [C.POINTER(Aligner), C.POINTER(Params)]
    L.quicked_free.argtypes = [C.POINTER(Aligner)]
    L.quicked_align.argtypes = [C.POINTER(Aligner), C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.quicked_set_device.argtypes = [C.c_int]
    L.quicked_align_batch.argtypes = [C.POINTER(Aligner), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    L.quicked_batch_create.restype = C.c_void_p
    L.quicked_batch_create.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_destroy.argtypes = [C.c_void_p]
    L.quicked_batch_destroy.restype = None
    L.quicked_batch_run.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int]
    L.quicked_batch_sync.argtypes = [C.c_void_p]
    L.quicked_batch_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_cigar_bytes.restype = C.c_int64
    L.quicked_batch_cigar_bytes.argtypes = [C.c_void_p]
    L.quicked_batch_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.quicked_batch_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    if hasattr(L, "quicked_batch_kernel_times"):      # (an older build loaded through QUICKED_HIP_LIB for an A/B run lacks the newer symbols)
        L.quicked_batch_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.quicked_batch_check_results.argtypes = [C.c_void_p, C.c_void_p]
    L.quicked_batch_validate.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.quicked_wire_words.restype = C.c_int64
    L.quicked_wire_words.argtypes = [C.c_int32, C.c_int]
    L.quicked_wire_pack.argtypes = [C.c_char_p, C.c_int32, C.c_int, C.c_void_p]
    L.quicked_batch_create_packed.restype = C.c_void_p
    L.quicked_batch_create_packed.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_reload.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_reload_packed.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.quicked_batch_fetch.argtypes = [C.c_void_p]
    L.quicked_pool_stats.argtypes = [C.c_void_p]
    if hasattr(L, "quicked_early_finish_stats"):
        L.quicked_early_finish_stats.argtypes = [C.c_void_p]
    L.quicked_batch_cigar_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.quicked_batch_deferred_pairs.restype = C.c_int64
    L.quicked_batch_deferred_pairs.argtypes = [C.c_void_p]
    L.quicked_wire_pack_pool.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.POINTER(C.c_int64)]
    L.quicked_wire_offsets.restype = C.c_int64
    L.quicked_wire_offsets.argtypes = [C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    L.quicked_wire_pack_isa.argtypes = [C.c_int]
    L.quicked_host_alloc.restype = C.c_void_p
    L.quicked_host_alloc.argtypes = [C.c_size_t]
    L.quicked_host_free.argtypes = [C.c_void_p]
    L.quicked_host_free.restype = None
    _LIB = L
    return L


def reload_env():
    """The library parses its QE_* switches once (qe_pool.h: SwitchTable); code that changes one in os.environ while the
    library is loaded -- the in-process parity tests that force a kernel form, bench.py's classic-flow leg -- calls this to
    have it parsed again.  No-op while the library is not loaded (its first use parses the environment as it is then)."""
    if _LIB is not None and hasattr(_LIB, "quicked_debug_reload_env"):
        _LIB.quicked_debug_reload_env()


def pool_trim():
    """quicked_pool_trim: the calling thread's device pools go back to the device"""
    return lib().quicked_pool_trim()


def pool_stats():
    """quicked_pool_stats of the calling thread: dict(pool_bytes, reclaim_events, sets, sub_batches, pool_budget) plus the
    process-wide book: bytes all pools of the thread's device hold, contexts in existence, contexts on lease"""
    v = np.zeros(8, dtype=np.int64)
    lib().quicked_pool_stats(v.ctypes.data)
    return dict(pool_bytes=int(v[0]), reclaim_events=int(v[1]), sets=int(v[2]), sub_batches=int(v[3]), pool_budget=int(v[4]),
                device_pool_bytes=int(v[5]), contexts=int(v[6]), contexts_leased=int(v[7]))


def early_finish_stats():
    """quicked_early_finish_stats: dict(flows, batches, merged_flows, merged_batches)"""
    v = np.zeros(4, dtype=np.int64)
    lib().quicked_early_finish_stats(v.ctypes.data)
    return dict(flows=int(v[0]), batches=int(v[1]), merged_flows=int(v[2]), merged_batches=int(v[3]))


def make_params(**kw):
    p = lib().quicked_default_params()
    for k, v in kw.items():
        if k in ("hew_threshold", "hew_percentage"):
            getattr(p, k)[0], getattr(p, k)[1] = v
        else:
            setattr(p, k, v)
    return p


class QuickedException(Exception):
    def __init__(self, status):
        self.status = status
        super().__init__(lib().quicked_status_msg(status).decode())


class QuickedAligner:
    """Mirror of the reference's C++/Python binding (bindings/cpp/quicked.hpp:40-66,
    bindings/python/quicked.cpp:33-45): same method names, same behaviour."""

    def __init__(self):
        self._lib = lib()
        self._params = self._lib.quicked_default_params()
        self._aligner = Aligner()
        st = self._lib.quicked_new(C.byref(self._aligner), C.byref(self._params))
        if self._lib.quicked_check_error(st):
            raise QuickedException(st)

    def __del__(self):
        try:
            self._lib.quicked_free(C.byref(self._aligner))
        except Exception:
            pass

    def align(self, pattern, text):
        pattern = pattern.encode() if isinstance(pattern, str) else pattern
        text = text.encode() if isinstance(text, str) else text
        st = self._lib.quicked_align(C.byref(self._aligner), pattern, len(pattern), text, len(text))
        if self._lib.quicked_check_error(st):
            raise QuickedException(st)
        return st

    def setAlgorithm(self, algo): self._params.algo = int(algo)
    def setOnlyScore(self, v): self._params.only_score = bool(v)
    def setBandwidth(self, v): self._params.bandwidth = int(v)
    def setWindowSize(self, v): self._params.window_size = int(v)
    def setOverlapSize(self, v): self._params.overlap_size = int(v)
    def setForceScalar(self, v): self._params.force_scalar = bool(v)

    def setHEWThreshold(self, v):
        self._params.hew_threshold[0] = self._params.hew_threshold[1] = int(v)

    def setHEWPercentage(self, v):
        self._params.hew_percentage[0] = self._params.hew_percentage[1] = int(v)

    def getScore(self): return self._aligner.score
    def getCigar(self): return self._aligner.cigar.decode() if self._aligner.cigar else "NULL"

    # additive: the batch entry point
    def alignBatch(self, pairs):
        n = len(pairs)
        pats = (C.c_char_p * n)(*[p for p, _ in pairs])
        txts = (C.c_char_p * n)(*[t for _, t in pairs])
        pl = (C.c_int * n)(*[len(p) for p, _ in pairs])
        tl = (C.c_int * n)(*[len(t) for _, t in pairs])
        scores = (C.c_int * n)(*([-1] * n))
        status = (C.c_int * n)()
        cigs = (C.c_char_p * n)()
        want = not self._params.only_score
        st = self._lib.quicked_align_batch(C.byref(self._aligner), n, pats, pl, txts, tl, scores,
                                           cigs if want else None, status)
        out = [(status[i], scores[i], (cigs[i].decode() if (want and cigs[i]) else None)) for i in range(n)]
        return st, out


WIRE_2BIT, WIRE_PLANES3 = 2, 3


def wire_offsets(length, wire):
    """dense word layout of sequences of these lengths: -> (int64 word offsets, total words)"""
    length = np.ascontiguousarray(length, dtype=np.int32)
    woff = np.zeros(len(length), dtype=np.int64)
    total = lib().quicked_wire_offsets(len(length), length.ctypes.data, wire, woff.ctypes.data)
    if total < 0:
        raise QuickedException(QUICKED_ERROR)
    return woff, int(total)


def wire_pack_pool(pool, off, length, wire, threads=0, out=None):
    """host-side serializer over a byte pool, one C call (quicked_wire_pack_pool: SIMD, multi-threaded): -> (uint64 words
    back to back, word offsets); `out` = a preallocated uint64 array (e.g. pinned) to pack into.  QuickedException if a
    sequence holds a symbol the wire format cannot represent."""
    L = lib()
    length = np.ascontiguousarray(length, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int64)
    woff, total = wire_offsets(length, wire)
    words = out if out is not None else np.zeros(total + 1, dtype=np.uint64)
    assert words.dtype == np.uint64 and len(words) >= total
    bad = C.c_int64(-1)
    st = L.quicked_wire_pack_pool(len(length), pool.ctypes.data, off.ctypes.data, length.ctypes.data, wire,
                                  words.ctypes.data, woff.ctypes.data, int(threads), C.byref(bad))
    if st < 0:
        raise QuickedException(st)
    return words, woff


class ResidentBatch:
    """quicked_batch_* : upload once, run many times (what bench.py times).  wire = WIRE_2BIT / WIRE_PLANES3 sends the
    packed form (quicked_batch_create_packed) instead of the ASCII pools."""

    def __init__(self, batch, wire=None):
        self._h = None
        self._lib = lib()
        self.n = len(batch)
        self._keep = batch
        if wire is None:
            self._h = self._lib.quicked_batch_create(
                self.n, batch.pattern_pool.ctypes.data, batch.pattern_off.ctypes.data, batch.pattern_len.ctypes.data,
                batch.text_pool.ctypes.data, batch.text_off.ctypes.data, batch.text_len.ctypes.data)
        else:
            pw, po = wire_pack_pool(batch.pattern_pool, batch.pattern_off, batch.pattern_len, wire)
            tw, to = wire_pack_pool(batch.text_pool, batch.text_off, batch.text_len, wire)
            self._wire = (pw, po, tw, to)
            self._h = self._lib.quicked_batch_create_packed(
                self.n, wire, pw.ctypes.data, po.ctypes.data, batch.pattern_len.ctypes.data,
                tw.ctypes.data, to.ctypes.data, batch.text_len.ctypes.data)
        if not self._h:
            raise RuntimeError("quicked_batch_create failed (no GPU / out of memory?)")

    @classmethod
    def from_wire(cls, batch, wire, pw, po, tw, to):
        """a batch from wire words that are already serialized (what a client holding packed data calls)"""
        self = cls.__new__(cls)
        self._lib = lib()
        self.n = len(batch)
        self._keep = (batch, pw, po, tw, to)
        self._h = self._lib.quicked_batch_create_packed(
            self.n, wire, pw.ctypes.data, po.ctypes.data, batch.pattern_len.ctypes.data,
            tw.ctypes.data, to.ctypes.data, batch.text_len.ctypes.data)
        if not self._h:
            raise RuntimeError("quicked_batch_create_packed failed")
        return self

    def run(self, params, sync=True):
        return self._lib.quicked_batch_run(self._h, C.byref(params), 1 if sync else 0)

    def sync(self):
        return self._lib.quicked_batch_sync(self._h)

    def fetch(self):
        """results of the last sync=False run -> host (quicked_batch_fetch)"""
        return self._lib.quicked_batch_fetch(self._h)

    def reload(self, batch):
        """new pairs into the same batch object (quicked_batch_reload): the device arena is reused"""
        self._keep = batch
        self.n = len(batch)
        return self._lib.quicked_batch_reload(
            self._h, self.n, batch.pattern_pool.ctypes.data, batch.pattern_off.ctypes.data, batch.pattern_len.ctypes.data,
            batch.text_pool.ctypes.data, batch.text_off.ctypes.data, batch.text_len.ctypes.data)

    def reload_wire(self, batch, wire, pw, po, tw, to):
        self._keep = (batch, pw, po, tw, to)
        self.n = len(batch)
        return self._lib.quicked_batch_reload_packed(
            self._h, self.n, wire, pw.ctypes.data, po.ctypes.data, batch.pattern_len.ctypes.data,
            tw.ctypes.data, to.ctypes.data, batch.text_len.ctypes.data)

    def configure(self, cigar_style=0, check=False):
        """cigar_style 0 = reference RLE "MXID", 1 = SAM "=XID", 2 = SAM "MID"; check = device-side validator"""
        return self._lib.quicked_batch_configure(self._h, cigar_style, 1 if check else 0)

    def validate(self, cigars):
        """device-side cigar_check_alignment of one CIGAR string (or None) per pair -> int32 verdicts"""
        off = np.full(self.n, -1, dtype=np.int64)
        blob = bytearray()
        for i, c in enumerate(cigars):
            if c is not None:
                off[i] = len(blob)
                blob += c.encode() + b"\0"
        ok = np.zeros(self.n, dtype=np.int32)
        st = self._lib.quicked_batch_validate(self._h, bytes(blob), len(blob), off.ctypes.data, ok.ctypes.data)
        if st < 0:
            raise QuickedException(st)
        return ok

    def check_results(self):
        ok = np.zeros(self.n, dtype=np.int32)
        self._lib.quicked_batch_check_results(self._h, ok.ctypes.data)
        return ok

    def scores(self):
        s = np.zeros(self.n, dtype=np.int32)
        st = np.zeros(self.n, dtype=np.int32)
        self._lib.quicked_batch_scores(self._h, s.ctypes.data, st.ctypes.data)
        return s, st

    def cigars(self):
        nb = self._lib.quicked_batch_cigar_bytes(self._h)
        pool = np.zeros(max(nb, 1), dtype=np.uint8)
        off = np.zeros(self.n, dtype=np.int64)
        self._lib.quicked_batch_cigars(self._h, pool.ctypes.data, off.ctypes.data)
        raw = pool.tobytes()
        out = []
        for o in off:
            if o < 0:
                out.append(None)
            else:
                e = raw.index(b"\0", o)
                out.append(raw[o:e].decode())
        return out

    def cigar_view(self):
        """-> (uint8 view of the batch's pinned string pool, int64 view of the per-pair offsets), no copies; valid until
        the next synchronous run / fetch / reload of the batch"""
        pool, off = C.c_void_p(), C.c_void_p()
        st = self._lib.quicked_batch_cigar_view(self._h, C.byref(pool), C.byref(off))
        if st < 0:
            raise QuickedException(st)
        nb = self._lib.quicked_batch_cigar_bytes(self._h)
        pv = np.ctypeslib.as_array((C.c_uint8 * max(nb, 1)).from_address(pool.value))[:nb] if (nb and pool.value) else np.zeros(0, np.uint8)
        ov = np.ctypeslib.as_array((C.c_int64 * self.n).from_address(off.value)) if self.n else np.zeros(0, np.int64)
        return pv, ov

    def deferred_pairs(self):
        """pairs of the last fetched / synchronous QuickEd run that were aligned at fetch time (quicked_batch.h)"""
        return int(self._lib.quicked_batch_deferred_pairs(self._h))

    def counters(self):
        c = np.zeros(8, dtype=np.int64)
        self._lib.quicked_batch_counters(self._h, c.ctypes.data)
        return c

    def kernel_time(self):
        """-> (sum of dominant-kernel HIP-event ms, launches) since the last call"""
        ms, n = C.c_double(0), C.c_int64(0)
        self._lib.quicked_batch_kernel_time(self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def kernel_times(self):
        """-> (ms[4], launches[4]) by kind since the last call: 0 score-only passes, 1 fills, 2 Hirschberg half passes"""
        ms, n = np.zeros(4, dtype=np.float64), np.zeros(4, dtype=np.int64)
        self._lib.quicked_batch_kernel_times(self._h, ms.ctypes.data, n.ctypes.data)
        return ms, n

    def close(self):
        if self._h:
            self._lib.quicked_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def pinned_copy(batch):
    """The same pairs with the two byte pools in pinned host memory (quicked_host_alloc)."""
    import numpy as _np
    L = lib()
    out = []
    for pool in (batch.pattern_pool, batch.text_pool):
        ptr = L.quicked_host_alloc(max(pool.nbytes, 1))
        if not ptr:
            raise MemoryError("quicked_host_alloc failed")
        arr = _np.ctypeslib.as_array((C.c_uint8 * max(pool.nbytes, 1)).from_address(ptr))
        arr[:pool.nbytes] = pool
        out.append((arr, ptr))
    from .datagen import PairBatch
    pb = PairBatch(out[0][0], batch.pattern_off, batch.pattern_len, out[1][0], batch.text_off, batch.text_len)
    pb._pinned = [p for _, p in out]
    return pb


def pinned_array(arr):
    """a copy of a numpy array in pinned host memory -> (array, pointer for quicked_host_free)"""
    import numpy as _np
    L = lib()
    ptr = L.quicked_host_alloc(max(arr.nbytes, 8))
    if not ptr:
        raise MemoryError("quicked_host_alloc failed")
    out = _np.ctypeslib.as_array((C.c_uint8 * max(arr.nbytes, 8)).from_address(ptr))[:arr.nbytes].view(arr.dtype)
    out[:] = arr
    return out, ptr


def pinned_free(pb):
    for p in getattr(pb, "_pinned", []):
        lib().quicked_host_free(p)
    pb._pinned = []
