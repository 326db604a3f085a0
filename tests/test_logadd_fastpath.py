"""Host-side check of csrc/logadd_fast.h (the fast correctly-rounded exp / ln_1p the duplex kernels use):
the same header compiled for the CPU must reproduce the x87 long-double reference on a strided sample of
every fast-path domain.  The exhaustive run (all 3e9 f32 arguments, 0 wrong) is recorded in
tools/verify/verify_logadd.cpp; on the GPU tests/test_gpu_duplex.py::test_logspace_arithmetic_bits checks
the device build of the same header bit for bit on a million operand pairs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_fast_paths_match_long_double_reference(tmp_path):
    exe = str(tmp_path / "verify_logadd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread",
                           os.path.join(ROOT, "tools", "verify", "verify_logadd.cpp"), "-o", exe])
    out = subprocess.run([exe, "4", "499"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all fast paths verified" in out.stdout, out.stdout
    assert " 0 wrong" in out.stdout
