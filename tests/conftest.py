import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:  # test modules share helpers (np_model, test_gpu_fast.check_tolerance)
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build libaptgpu.so (hipcc cross-compiles without a GPU) when a fresh checkout has none:
    the host-side tests (C-ABI symbols, FIR design, WAV header walk) load it."""
    import noaa_apt_amd as apt
    if not os.path.exists(apt.lib_path()):
        apt.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (oracle/libaptoracle.so), built on demand."""
    from oracle import binding
    binding.lib()
    return binding
