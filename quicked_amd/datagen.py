"""Seeded synthetic read-pair generator (ctypes front for csrc/datagen.c).

Reproduces the distribution of the reference's tools/generate_dataset
(generate_dataset.c:52-63,108-199) with a counter-based PRNG so that pair ``i``
depends only on ``(seed, i)``; see SURVEY.md 8(d).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libqe_datagen.so")
        if not os.path.exists(path):
            from . import build
            build.build_datagen()
        lib = C.CDLL(path)
        lib.qe_gen_pattern_capacity.restype = C.c_int64
        lib.qe_gen_pattern_capacity.argtypes = [C.c_int64, C.c_double, C.c_int64, C.c_int64]
        lib.qe_gen_pair.restype = C.c_int64
        lib.qe_gen_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_double, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p]
        lib.qe_gen_batch.restype = None
        lib.qe_gen_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = lib
    return _LIB


DEFAULT_SEED = 0x51CED


class PairBatch:
    """Pairs stored back to back in two byte pools (the batch wire format)."""

    def __init__(self, pattern_pool, pattern_off, pattern_len, text_pool, text_off, text_len):
        self.pattern_pool = pattern_pool      # np.uint8
        self.pattern_off = pattern_off        # np.int64
        self.pattern_len = pattern_len        # np.int32
        self.text_pool = text_pool
        self.text_off = text_off
        self.text_len = text_len

    def __len__(self):
        return len(self.pattern_len)

    def pattern(self, i):
        o, n = int(self.pattern_off[i]), int(self.pattern_len[i])
        return self.pattern_pool[o:o + n].tobytes()

    def text(self, i):
        o, n = int(self.text_off[i]), int(self.text_len[i])
        return self.text_pool[o:o + n].tobytes()

    def pairs(self):
        for i in range(len(self)):
            yield self.pattern(i), self.text(i)

    def cells(self):
        return int((self.pattern_len.astype(np.int64) * self.text_len.astype(np.int64)).sum())

    def concat(self, other):
        """the pairs of `self` followed by those of `other` (mixed data sets: ordinary reads + a few with large indels)"""
        return PairBatch(np.concatenate([self.pattern_pool, other.pattern_pool]),
                         np.concatenate([self.pattern_off, other.pattern_off + len(self.pattern_pool)]),
                         np.concatenate([self.pattern_len, other.pattern_len]),
                         np.concatenate([self.text_pool, other.text_pool]),
                         np.concatenate([self.text_off, other.text_off + len(self.text_pool)]),
                         np.concatenate([self.text_len, other.text_len]))


def generate(count, length, error, seed=DEFAULT_SEED, first=0, indels_num=0, indels_len=0):
    """``count`` pairs: text = ``length`` uniform ACGT bases, pattern = text with
    ceil(length*error) (or ``int(error)`` if >= 1) sequential mismatch/ins/del edits."""
    lib = _lib()
    cap = lib.qe_gen_pattern_capacity(length, float(error), indels_num, indels_len)
    ppool = np.zeros(count * cap, dtype=np.uint8)
    tpool = np.zeros(count * length, dtype=np.uint8)
    poff = np.zeros(count, dtype=np.int64)
    toff = np.zeros(count, dtype=np.int64)
    plen = np.zeros(count, dtype=np.int32)
    tlen = np.zeros(count, dtype=np.int32)
    lib.qe_gen_batch(seed, first, count, length, float(error), indels_num, indels_len,
                     ppool.ctypes.data, poff.ctypes.data, plen.ctypes.data,
                     tpool.ctypes.data, toff.ctypes.data, tlen.ctypes.data)
    return PairBatch(ppool, poff, plen, tpool, toff, tlen)
