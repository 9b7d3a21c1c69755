"""The host side of libbzh2.so multiplies field elements too (witness synthesis, Jacobian -> affine read-backs, challenge
algebra): csrc/field.cuh gives the host pass a 64-bit-limb Montgomery product.  tests/helpers/host_field_check.hip compiles
against the library's own header and compares it with the 32-bit CIOS on all four fields (host code only: no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_64_bit_field_product_equals_the_32_bit_one(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "host_field_check")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-I", os.path.join(ROOT, "battlezips-halo2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "helpers", "host_field_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 4 and all(ln.endswith("-> OK") for ln in lines), out.stdout
    assert all("64-bit limbs" in ln for ln in lines), "the host pass did not take the 64-bit path:\n" + out.stdout
