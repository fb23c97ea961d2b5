"""In-tree builds: the gfx950 HIP C-ABI library and the host-side data generator.

Everything is compiled with explicit hipcc / gcc commands into ``quicked_amd/``
so the ``.so`` files travel with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(HERE, "libquicked_hip.so")
DATAGEN_LIB = os.path.join(HERE, "libqe_datagen.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    subprocess.run(cmd, check=True)


def build_datagen(force=False):
    src = os.path.join(CSRC, "datagen.c")
    if force or _newer(DATAGEN_LIB, [src]):
        _run(["gcc", "-O3", "-fPIC", "-shared", "-fopenmp", "-Wall", "-Wextra", src, "-lm", "-o", DATAGEN_LIB])
    return DATAGEN_LIB


def hip_sources():
    out = []
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".cpp", ".h", ".hpp")):
            out.append(os.path.join(CSRC, name))
    out.append(os.path.join(ROOT, "include", "quicked.h"))
    out.append(os.path.join(ROOT, "include", "quicked_batch.h"))
    return out


def build_hip(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # one translation unit: qe_driver.hip includes qe_kernels.hip
    if force or _newer(HIP_LIB, hip_sources()):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC,
               os.path.join(CSRC, "qe_driver.hip"), "-o", HIP_LIB]
        _run(cmd)
    return HIP_LIB


HARNESS = os.path.join(ROOT, "tools", "bin", "align_benchmark")


def build_harness(force=False):
    """tools/align_benchmark.cpp: the reference CLI's interface over the C-ABI batch call."""
    src = os.path.join(ROOT, "tools", "align_benchmark.cpp")
    os.makedirs(os.path.dirname(HARNESS), exist_ok=True)
    if force or _newer(HARNESS, [src, HIP_LIB, os.path.join(ROOT, "include", "quicked_batch.h")]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", src, "-I", os.path.join(ROOT, "include"), "-L", HERE, "-lquicked_hip",
              "-Wl,-rpath," + HERE, "-o", HARNESS])
    return HARNESS


def build_all(force=False):
    build_datagen(force)
    build_hip(force)
    build_harness(force)
