import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch bundles its own HIP runtime: if libmort_hip.so (linked to /opt/rocm's) initialises HIP first, torch.cuda later reports
    # "No HIP GPUs are available".  The tests that hand torch tensors to the C ABI need torch's runtime up first, whatever subset runs.
    try:
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    # build the CPU pieces once (host scene layer + oracle); the HIP library is built by
    # __graft_entry__.build() / `make hip` and must already exist for the gpu tests
    need = [os.path.join(ROOT, "mort_amd", "lib", "libmort_host.so"), os.path.join(ROOT, "oracle", "libmort_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "host", "oracle"])


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_ctx():
    """One mort_ctx on cuda:0 for the whole gpu session.  No fallback: a missing library or GPU fails loudly."""
    from mort_amd import hip
    ctx = hip.Context(0)
    yield ctx
    ctx.close()
