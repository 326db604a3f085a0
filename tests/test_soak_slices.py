"""Slices of the off-suite soaks (tools/beam_soak.py, tools/duplex_soak.py, tools/hostjob_soak.py) in the CPU suite, on
the lockstep wave64 emulator: the same kernels, the same oracle, special posteriors (NaN, +inf, > 1, zeros, negative)
injected into random draws -- the kind of input that found the max-mode merge-order defect."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

from emu_util import emulated_kernels  # noqa: E402


def _run(mod, argv):
    old = sys.argv
    sys.argv = ["soak"] + argv
    try:
        return mod.main()
    finally:
        sys.argv = old


def test_beam_soak_slice():
    with emulated_kernels():
        import beam_soak
        assert _run(beam_soak, ["200000", "8"]) == 0


def test_duplex_soak_slice():
    with emulated_kernels():
        import duplex_soak
        assert _run(duplex_soak, ["100365", "8"]) == 0   # (100369: one of the seeds that found the defect)


def test_hostjob_soak_slice():
    with emulated_kernels():
        import hostjob_soak
        import fast_ctc_decode_amd as fcd
        assert hostjob_soak.run(fcd, 700000, 20) == 0


def test_tools_compile():
    """every developer script under tools/ at least parses (they are not imported by the suite otherwise)"""
    import glob
    import py_compile
    root = os.path.join(os.path.dirname(HERE), "tools")
    scripts = sorted(glob.glob(os.path.join(root, "*.py")))
    assert len(scripts) > 15
    for path in scripts:
        py_compile.compile(path, doraise=True)
