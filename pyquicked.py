"""`pyquicked` -- the module name of the reference's Python binding (bindings/python/quicked.cpp:27-66),
served by the ctypes mirror over libquicked_hip.so, so that `from pyquicked import QuickedAligner,
QuickedException` (examples/bindings/basic.py of the reference) works unchanged with this repository on
PYTHONPATH."""
import enum as _enum

from quicked_amd import capi as _capi

QuickedAligner = _capi.QuickedAligner
QuickedException = _capi.QuickedException


class QuickedAlgo(_enum.IntEnum):
    QUICKED = _capi.QUICKED
    WINDOWED = _capi.WINDOWED
    BANDED = _capi.BANDED
    HIRSCHBERG = _capi.HIRSCHBERG


class QuickedStatus(_enum.IntEnum):
    QUICKED_OK = 0
    QUICKED_ERROR = -1
    QUICKED_FAIL_NON_CONVERGENCE = -2
    QUICKED_UNKNOWN_ALGO = -3
    QUICKED_EMPTY_SEQUENCE = -4
    QUICKED_UNIMPLEMENTED = -10
    QUICKED_WIP = 1


# pybind11's export_values(): the enum members are module attributes too
globals().update(QuickedAlgo.__members__)
globals().update(QuickedStatus.__members__)
