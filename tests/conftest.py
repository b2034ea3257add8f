import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library through its C-ABI (built on demand; hipcc cross-compiles without a GPU)."""
    from visualrwkv_amd import build, hip_lib as hl
    build.build()
    return hl.load()


@pytest.fixture(scope="session")
def emu_lib():
    """Host lockstep emulation of the device kernels (tests/emu)."""
    from tests.emu.build import build_emu
    import ctypes
    return ctypes.CDLL(build_emu())
