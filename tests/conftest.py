import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _emulated_kernels_if_requested():
    """Developer switch: FCD_TEST_EMU=1 python -m pytest tests -m gpu runs the numpy-input GPU tests on
    tests/hipemu's lockstep emulation of the kernels (no GPU needed; torch-tensor tests still need one).
    Never set by the driver: on the GPU box the -m gpu tests run on the real libfcd_hip.so."""
    if os.environ.get("FCD_TEST_EMU", "0") != "1":
        yield
        return
    from emu_util import emulated_kernels
    with emulated_kernels():
        yield
