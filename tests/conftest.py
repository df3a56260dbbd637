"""pytest configuration: the `gpu` marker and library fixtures.

CPU suite (`-m "not gpu"`): oracle vs golden vectors / partial reference build,
host logic, C-ABI surface.  GPU suite (`-m gpu`): parity of the HIP path against
the oracle, always through the C-ABI.
"""
import importlib
import os
import subprocess
import sys

import pytest

# torch bundles its own HIP runtime: when a test process uses both torch and libsift3d_hip.so (device buffers for
# the *_dev entry points), torch has to be loaded first so that the process ends up with ONE runtime.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("3d_sift_cuda_amd")


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def built(pkg):
    """Make sure the product libraries exist (cross-compiles on CPU-only hosts)."""
    if not (os.path.exists(pkg.LIB_HIP) and os.path.exists(pkg.LIB_HOST)):
        pkg.build()
    return pkg


@pytest.fixture(scope="session")
def ctx_factory(built):
    def make(nx, ny, nz):
        return built.Context(nx, ny, nz, device=0)
    return make
