"""CPU-only: the C-ABI library builds (hipcc cross-compiles gfx950), loads, and exports every
symbol include/fcd.h (the drop-in boundary) and include/fcd_debug.h (test hooks, developer instruments) declare.
No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from fast_ctc_decode_amd import _native, build
    build.build()
    return _native.load()


def header_functions(headers=("fcd.h", "fcd_debug.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(fcd_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_the_boundary_header_holds_no_test_hooks():
    """what a binding generator consumes: no fcd_debug_*, probes or sweeps in fcd.h"""
    for n in header_functions(("fcd.h",)):
        assert not re.search(r"debug|_probe|_sweep|_profile", n), n
    assert len(header_functions(("fcd_debug.h",))) >= 9


def test_header_symbols_exported(lib):
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libfcd_hip.so does not export %s" % n


def test_binding_covers_header():
    from fast_ctc_decode_amd import _native
    assert sorted(_native.SYMBOLS) == header_functions()


def test_version_and_strings(lib):
    assert lib.fcd_version() == 1
    # exact SearchError Display strings (src/lib.rs:46-53)
    assert lib.fcd_status_string(1) == b"Ran out of search space (beam_cut_threshold too high)"
    assert lib.fcd_status_string(2) == b"Failed to compare values (NaNs in input?)"
    assert lib.fcd_status_string(3) == b"Invalid envelope values"


def test_phred_host_helper(lib):
    # K5 (src/search.rs:513-525) through the product's host helper
    import numpy as np
    f32 = np.float32
    probs = [f32(0.0), f32(0.5), f32(1.0) - f32(1e-1), f32(1.0) - f32(1e-2), f32(1.0) - f32(1e-3),
             f32(1.0) - f32(1e-4), f32(1.0) - f32(1e-5), f32(1.0) - f32(1e-6), f32(1.0)]
    assert "".join(chr(lib.fcd_phred(float(p), 1.0, 0.0)) for p in probs) == "!$+5?IIII"


def test_no_gpu_fails_loudly(lib):
    """Without a device the product must raise, never fall back to a CPU path."""
    import numpy as np
    import fast_ctc_decode_amd as fcd
    if lib.fcd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fcd.beam_search(np.full((10, 5), 0.2, np.float32), "NACGT")


def test_validation_without_gpu():
    """Argument validation (src/lib.rs:331-349) happens before any device work."""
    import numpy as np
    import fast_ctc_decode_amd as fcd
    x = np.full((10, 5), 0.2, np.float32)
    with pytest.raises(ValueError, match="alphabet size 4 does not match probability matrix inner dimension 5"):
        fcd.beam_search(x, "NACG")
    with pytest.raises(ValueError, match="beam_size cannot be 0"):
        fcd.beam_search(x, "NACGT", 0)
    with pytest.raises(ValueError, match="beam_cut_threshold must be at least 0.0"):
        fcd.beam_search(x, "NACGT", 5, -0.1)
    with pytest.raises(ValueError, match="beam_cut_threshold cannot be more than 0.2"):
        fcd.beam_search(x, "NACGT", 5, 0.2)
    with pytest.raises(ValueError, match="Empty alphabet given"):
        fcd.viterbi_search(x, "")
    with pytest.raises(ValueError, match="alphabet size does not match probability matrix dimensions"):
        fcd.viterbi_search(x, "NACG")
    with pytest.raises(TypeError):
        fcd.beam_search(x.astype(np.float64), "NACGT")
    with pytest.raises(TypeError):
        fcd.beam_search(x)


def test_compiled_module_surface_and_validation():
    """The compiled drop-in module (csrc/pymodule.cpp): reference names, defaults, messages."""
    import numpy as np
    import fast_ctc_decode as m
    assert m.__version__ == "0.3.7"
    for name in ("beam_search", "beam_search_duplex", "viterbi_search", "crf_greedy_search",
                 "crf_beam_search", "crf_beam_search_duplex"):  # src/lib.rs:619-624
        assert callable(getattr(m, name))
    x = np.full((10, 5), 0.2, np.float32)
    with pytest.raises(ValueError, match="alphabet size 6 does not match probability matrix inner dimension 5"):
        m.beam_search(x, "NACGTX")
    with pytest.raises(ValueError, match="beam_size cannot be 0"):
        m.beam_search(network_output=x, alphabet="NACGT", beam_size=0)
    with pytest.raises(ValueError, match="beam_cut_threshold cannot be more than 0.33333334"):
        m.beam_search(x[:, :3], "NAB", 5, 0.5)
    with pytest.raises(ValueError, match="inner axes of the network outputs do not match"):
        m.beam_search_duplex(x, x[:, :4], "NACGT")
    with pytest.raises(ValueError, match="the lengths of network_output_1 and envelope do not match"):
        m.beam_search_duplex(x, x, "NACGT", np.zeros((3, 2), np.uint64))
    with pytest.raises(ValueError, match="the inner axis of envelope must have size 2"):
        m.beam_search_duplex(x, x, "NACGT", np.zeros((10, 3), np.uint64))
    with pytest.raises(ValueError, match="Empty alphabet given"):
        m.viterbi_search(x, [])
    with pytest.raises(TypeError):
        m.viterbi_search(x.astype(np.float64), "NACGT")
    with pytest.raises(TypeError):
        m.crf_beam_search(x, x[0], "NACGT")  # rank 2 instead of 3
