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
    # Three translation units: the device side (qe_driver.hip, which includes the kernels of qe_kernels.hip: hipcc), the
    # C-ABI (qe_capi.cpp: host code over the HIP runtime API, g++) and the host-only SIMD packer (qe_hostpack.cpp: x86
    # intrinsics with per-function targets, g++; never seen by the device pass).  qe_pool.h / qe_batch.h are what they share.
    if force or _newer(HIP_LIB, hip_sources()):
        rocm_inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
        host = ["g++", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra", "-pthread", "-c",
                "-I", os.path.join(ROOT, "include"), "-I", CSRC]
        hostpack_o = os.path.join(HERE, "qe_hostpack.o")
        _run(host + [os.path.join(CSRC, "qe_hostpack.cpp"), "-o", hostpack_o])
        capi_o = os.path.join(HERE, "qe_capi.o")
        _run(host + ["-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__", "-I", rocm_inc, os.path.join(CSRC, "qe_capi.cpp"), "-o", capi_o])
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC,
               os.path.join(CSRC, "qe_driver.hip"), "-Wl," + hostpack_o, "-Wl," + capi_o, "-lpthread", "-o", HIP_LIB]      # -Wl,: hipcc would compile a bare .o as HIP source
        cmd += os.environ.get("QE_HIPCC_FLAGS", "").split()      # experiments: -D switches of qe_kernels.hip
        _run(cmd)
    return HIP_LIB


HARNESS = os.path.join(ROOT, "tools", "bin", "align_benchmark")


def build_harness(force=False):
    """tools/align_benchmark.cpp: the reference CLI's interface over the C-ABI batch call."""
    src = os.path.join(ROOT, "tools", "align_benchmark.cpp")
    os.makedirs(os.path.dirname(HARNESS), exist_ok=True)
    if force or _newer(HARNESS, [src, HIP_LIB, os.path.join(ROOT, "include", "quicked_batch.h")]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", src, "-I", os.path.join(ROOT, "include"), "-L", HERE, "-lquicked_hip",
              "-ldl", "-Wl,-rpath," + HERE, "-o", HARNESS])
    return HARNESS


REF = "/root/reference"
REF_CALLERS = os.path.join(ROOT, "tools", "bin", "ref_callers")


def build_ref_callers(force=False):
    """Drop-in proof: the reference's OWN callers -- tests/quicked_harness.c, examples/*.c and the C++
    binding with its examples -- compiled unmodified, from where they lie under /root/reference, against
    the REFERENCE's headers, and linked against libquicked_hip.so instead of libquicked.a.  Only where the
    reference tree exists; the binaries travel with the snapshot like the in-tree .so files."""
    if not os.path.isdir(os.path.join(REF, "quicked")):
        return None
    os.makedirs(REF_CALLERS, exist_ok=True)
    inc = ["-I", REF, "-I", os.path.join(REF, "quicked"), "-I", os.path.join(REF, "quicked", "include")]
    link = ["-L", HERE, "-lquicked_hip", "-Wl,-rpath," + HERE]
    jobs = [(os.path.join(REF, "tests", "quicked_harness.c"), "quicked_harness", "gcc", [])]
    for name in sorted(os.listdir(os.path.join(REF, "examples"))):
        if name.endswith(".c"):
            jobs.append((os.path.join(REF, "examples", name), "example_" + name[:-2], "gcc", []))
    cpp_binding = os.path.join(REF, "bindings", "cpp", "quicked.cpp")
    for name in sorted(os.listdir(os.path.join(REF, "examples", "bindings"))):
        if name.endswith(".cpp"):
            jobs.append((os.path.join(REF, "examples", "bindings", name), "binding_" + name[:-4] + "_cpp", "g++",
                         [cpp_binding, "-I", os.path.join(REF, "bindings", "cpp")]))
    for src, out, cc, extra in jobs:
        exe = os.path.join(REF_CALLERS, out)
        if force or _newer(exe, [src, HIP_LIB]):
            _run([cc, "-O2", "-w", src] + extra + inc + link + ["-o", exe])
    return REF_CALLERS


VALU_RATE = os.path.join(ROOT, "tools", "bin", "valu_rate")


def build_valu_rate(force=False):
    """tools/valu_rate.hip: per-instruction VALU issue rates of gfx950 (the table behind DESIGN.md 4.1)"""
    src = os.path.join(ROOT, "tools", "valu_rate.hip")
    os.makedirs(os.path.dirname(VALU_RATE), exist_ok=True)
    if force or _newer(VALU_RATE, [src, os.path.join(CSRC, "qe_kernels.hip"), os.path.join(CSRC, "qe_types.h")]):
        _run(["hipcc", "--offload-arch=gfx950", "-O3", "-w", src, "-o", VALU_RATE])
    return VALU_RATE


PMC_CALIB = os.path.join(ROOT, "tools", "bin", "pmc_calib")


def build_pmc_calib(force=False):
    """tools/pmc_calib.hip: known-byte-count streams that calibrate FETCH_SIZE / WRITE_SIZE per access width"""
    src = os.path.join(ROOT, "tools", "pmc_calib.hip")
    os.makedirs(os.path.dirname(PMC_CALIB), exist_ok=True)
    if force or _newer(PMC_CALIB, [src]):
        _run(["hipcc", "--offload-arch=gfx950", "-O3", "-w", src, "-o", PMC_CALIB])
    return PMC_CALIB


SEGV_TRACE = os.path.join(ROOT, "tools", "bin", "libsegvtrace.so")


def build_segv_trace(force=False):
    """tools/segv_trace.c: LD_PRELOAD backtrace helper for diagnosis runs on the GPU box (never linked into the library)"""
    src = os.path.join(ROOT, "tools", "segv_trace.c")
    os.makedirs(os.path.dirname(SEGV_TRACE), exist_ok=True)
    if force or _newer(SEGV_TRACE, [src]):
        _run(["gcc", "-O1", "-g", "-fPIC", "-shared", src, "-o", SEGV_TRACE])
    return SEGV_TRACE


def build_all(force=False):
    build_segv_trace(force)
    build_datagen(force)
    build_hip(force)
    build_harness(force)
    build_ref_callers(force)
    build_valu_rate(force)
    build_pmc_calib(force)
