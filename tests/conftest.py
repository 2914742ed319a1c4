import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


try:                       # before libmirt.so is loaded: torch ships its own HIP runtime, and whichever copy is loaded first
    import torch  # noqa: F401  is the one that sees the GPU (tests that hand torch buffers to libmirt need both to)
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from cpulibs import GOLDEN

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def oracle():
    from cpulibs import CpuTracer

    return CpuTracer("oracle")


@pytest.fixture(scope="session")
def mirt():
    """The HIP library, initialised on device 0.  Fails loudly (no fallback) when there is no GPU."""
    from raytracer_amd import capi

    capi.init(0)
    return capi
