"""Host logic of the device-memory design (quicked_amd/csrc/qe_pool.h) without a GPU: the lease of contexts across threads that
end, the arithmetic of the per-device book a planning thread reads (ledger_plan), the API scope's lock discipline.  The test
program is plain C++ (tests/native/pool_unit.cpp) over the header the library itself is built from; it makes no HIP call."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir(os.path.join(ROCM, "include", "hip")),
                    reason="needs g++ and the HIP headers")
def test_pool_book_and_leases(tmp_path):
    exe = str(tmp_path / "pool_unit")
    cmd = ["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-D__HIP_PLATFORM_AMD__",
           "-I" + os.path.join(ROCM, "include"), "-I" + os.path.join(ROOT, "quicked_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "pool_unit.cpp"),
           "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert built.returncode == 0, built.stderr[-2000:]
    ran = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert ran.returncode == 0 and "pool_unit ok" in ran.stdout, (ran.stdout + ran.stderr)[-2000:]
