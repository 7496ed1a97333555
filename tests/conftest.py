import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libhop_emu.so")


@pytest.fixture(scope="session")
def hop():
    import hop_loader
    mod = hop_loader.load()
    if os.environ.get("HOP_TEST_EMU"):
        # TEST INFRASTRUCTURE (tests/emu): the `-m gpu` tests run against the product's kernel sources compiled for the functional HIP model on
        # the CPU -- only ever selected by this variable, which tests/test_emu_kernels_cpu.py sets for its child runs.  The product itself
        # (api.lib() without this) loads libhop.so or fails.
        from hop_amd import api
        api.LIB_PATH = EMU_LIB
        api._lib = None
    return mod


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
