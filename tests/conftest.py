import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library is a build product (git-ignored): on a fresh checkout compile it before the
    first test needs it (hipcc cross-compiles gfx950 without a GPU; a few minutes, once)."""
    lib = ROOT / "gcd_amd" / "libgcd_amd.so"
    if not lib.exists():
        from gcd_amd.csrc import build as _b
        _b.build(verbose=False)


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 in fp64 (b is the reference)."""
    import torch
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from gcd_amd import _lib
    _lib.load()  # the HIP extension must be there on a GPU box: fail, do not skip
    return torch.device("cuda:0")
