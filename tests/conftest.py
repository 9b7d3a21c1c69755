import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "battlezips-halo2_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (the .so files are git-ignored): build it once, as __graft_entry__.build() does;
    # hipcc cross-compiles gfx950 without a GPU.  A failed build is left for the tests that load the library to report.
    lib = os.path.join(ROOT, "battlezips-halo2_amd", "libbzh2.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "battlezips-halo2_amd", "csrc"), "-j8", "ARCH=gfx950"],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle_c():
    import coracle
    coracle.build()
    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One bzh2 context on cuda:0 for the whole GPU session; fails loudly if the
    HIP library is missing or no device is visible (no CPU fallback exists)."""
    # torch bundles its own HIP runtime: it must initialise before libbzh2.so's (the system one) does,
    # otherwise torch later reports "No HIP GPUs are available" in the same process
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    import bzh2
    bzh2.load()
    if bzh2.device_count() < 1:
        pytest.fail("bzh2: no GPU visible but a gpu-marked test was selected")
    ctx = bzh2.Context(0)
    yield ctx
    ctx.close()
