import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU parity checker (oracle/libzpc_oracle.so) -- test infrastructure, built on demand."""
    import ctypes
    import subprocess
    path = os.path.join(ROOT, "oracle", "libzpc_oracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libzpc_oracle.so"])
    return ctypes.CDLL(path)


@pytest.fixture(scope="session")
def hiplib():
    from zpc_amd import lib
    return lib()


@pytest.fixture(scope="session")
def pol():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zpc_amd import rocm_exec
    return rocm_exec()


def pytest_collection_modifyitems(config, items):
    """The full-size runs (minutes each, 64 Mi particles) go last: under `-x` one of them must never hide the parity tests."""
    late = [it for it in items if "test_fullsize_gpu" in it.nodeid]
    if late:
        rest = [it for it in items if "test_fullsize_gpu" not in it.nodeid]
        items[:] = rest + late
