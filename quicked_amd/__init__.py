"""quicked_amd -- MI355X-native (gfx950) drop-in for the QuickEd alignment hot path.

The product is ``libquicked_hip.so`` (hand-written HIP kernels behind the
reference's ``quicked_new / quicked_align / quicked_free`` C-ABI plus an
additive batch API).  This package holds its sources (``csrc/``), the build
recipe, a ctypes binding that mirrors the reference's ``pyquicked`` module
(bindings/python/quicked.cpp:27-66) and the seeded data generator.
"""
__all__ = ["build", "datagen"]
