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


@pytest.fixture(autouse=True)
def _no_cached_sessions_across_gpu_tests(request):
    """The one-shot entry points keep idle sessions (plan + device buffers) per (settings, rate, sync, mode); the library's
    A/B switches (APTGPU_FUSED_ANY, APTGPU_PHASE_*, APTGPU_FUSED_PAD ...) are read when a plan is created.  A test that sets
    one must not be served the plan an earlier test left behind — which of them did depended on the order and the -k
    filter of the run — so every GPU test starts and ends with an empty cache."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import noaa_apt_amd as apt
    apt.cache_clear()
    yield
    apt.cache_clear()
