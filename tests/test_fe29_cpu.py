"""The unsaturated 9 x 29-bit field arithmetic of the MSM's bucket accumulator (csrc/fe29.cuh, csrc/curve29.cuh) on the host:
tests/helpers/fe29_check.hip compiles against the library's own headers and compares products, squares, biased subtractions at
their documented input bounds, the round trip with the saturated 2^256-Montgomery form, and chains of mixed additions (with
the doubling and the inverse special cases) against the saturated code as group elements, on both Pasta fields.  Host code
only: no GPU needed (the device build of the same functions is what k_msm_accumulate<.., U29 = true> runs; GPU parity:
tests/test_gpu_msm.py and the BZH_ACC_SATURATED settings of tests/test_gpu_env_paths.py)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unsaturated_field_arithmetic_equals_the_saturated_one(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "fe29_check")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-I", os.path.join(ROOT, "battlezips-halo2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "helpers", "fe29_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 2 and all(ln.endswith("-> OK") for ln in lines), out.stdout
