"""One behaviour, two host layers: the compiled drop-in module `fast_ctc_decode` (csrc/pymodule.cpp) and the
Python mirror `fast_ctc_decode_amd` must reject malformed arguments with the SAME exception type and the SAME
text -- the reference's (src/lib.rs:143-146,190-195,331-349,423-468; tests/test_decode.py:45-98,217-225).
Argument checking happens before any device work, so this runs without a GPU."""
import numpy as np
import pytest

import fast_ctc_decode as compiled
import fast_ctc_decode_amd as mirror

X = np.full((6, 5), 0.2, np.float32)
X3 = np.full((6, 4, 5), 0.2, np.float32)
INIT = np.array([0, 0, 1, 0], np.float32)
ENV = np.stack([np.zeros(6, np.uint64), np.full(6, 6, np.uint64)], 1)

CASES = [
    # (function, args, kwargs)
    ("beam_search", (X, "NACG"), {}),                                  # alphabet too short
    ("beam_search", (X, "NACGTT"), {}),                                # too long
    ("beam_search", (X, "NACGT"), {"beam_size": 0}),
    ("beam_search", (X, "NACGT"), {"beam_size": -1}),
    ("beam_search", (X, "NACGT"), {"beam_size": -1, "beam_cut_threshold": 5.0}),  # extraction errors come first
    ("beam_search", (X, "NACGT"), {"beam_size": 2.5}),
    ("beam_search", (X, "NACGT"), {"beam_size": True}),
    ("beam_search", (X, "NACGT"), {"beam_cut_threshold": -0.1}),
    ("beam_search", (X, "NACGT"), {"beam_cut_threshold": 0.2}),        # == 1/len(alphabet)
    ("beam_search", (X, "NACGT"), {"beam_cut_threshold": 0.5}),
    ("beam_search", (X, "NACG"), {"beam_size": 0, "beam_cut_threshold": -1.0}),   # validation ORDER: alphabet first
    ("beam_search", (X, "NACGT"), {"beam_size": 0, "beam_cut_threshold": -1.0}),  # then beam_size
    ("beam_search", (X.astype(np.float64), "NACGT"), {}),
    ("beam_search", (X3, "NACGT"), {}),
    ("beam_search", (X.tolist(), "NACGT"), {}),
    ("beam_search", (X, 5), {}),
    ("beam_search", (X,), {}),                                         # missing argument
    ("viterbi_search", (X, ""), {}),
    ("viterbi_search", (X, "NACG"), {}),
    ("viterbi_search", (X.astype(np.float16), "NACGT"), {}),
    ("viterbi_search", (X,), {}),
    ("crf_beam_search", (X3, INIT, ""), {}),
    ("crf_beam_search", (X3, INIT, "NACG"), {}),
    ("crf_beam_search", (X, INIT, "NACGT"), {}),
    ("crf_beam_search", (X3, INIT.astype(np.float64), "NACGT"), {}),
    ("crf_beam_search", (X3, INIT, "NACGT"), {"beam_size": -3}),
    ("crf_greedy_search", (X3, INIT, ""), {}),
    ("crf_greedy_search", (X3, INIT, "NACGTA"), {}),
    ("beam_search_duplex", (X, np.full((6, 4), 0.25, np.float32), "NACGT"), {}),          # inner axes differ
    ("beam_search_duplex", (X, X, "NACG"), {}),
    ("beam_search_duplex", (X, X, "NACGT"), {"beam_size": 0}),
    ("beam_search_duplex", (X, X, "NACGT"), {"beam_cut_threshold": 0.2}),
    ("beam_search_duplex", (X, X, "NACGT"), {"envelope": ENV[:5]}),                        # wrong length
    ("beam_search_duplex", (X, X, "NACGT"), {"envelope": np.zeros((6, 3), np.uint64)}),    # wrong inner axis
    ("beam_search_duplex", (X, X, "NACGT"), {"envelope": ENV.astype(np.int64)}),           # wrong dtype
    ("crf_beam_search_duplex", (X3, INIT, X3[:, :, :4].copy(), INIT, "NACGT"), {}),
    ("crf_beam_search_duplex", (X3, INIT, X3, INIT, "NACGT"), {"beam_size": 0}),
    ("crf_beam_search_duplex", (X3, INIT, X3, INIT, "NACGT"), {"envelope": ENV[:2]}),
]


def outcome(module, name, args, kwargs):
    try:
        getattr(module, name)(*args, **kwargs)
    except Exception as e:  # noqa: BLE001 -- the point is to compare whatever is raised
        return type(e).__name__, str(e)
    return "returned", ""


@pytest.mark.parametrize("case", range(len(CASES)))
def test_same_rejection(case):
    name, args, kwargs = CASES[case]
    a = outcome(compiled, name, args, kwargs)
    b = outcome(mirror, name, args, kwargs)
    assert a[0] != "returned" and b[0] != "returned", (name, kwargs, a, b)
    assert a[0] == b[0], (name, kwargs, a, b)
    if a[0] != "TypeError":   # TypeError texts of missing / mistyped arguments come from the binding generator
        assert a[1] == b[1], (name, kwargs, a, b)


def test_reference_messages():
    """The texts the reference's own tests pin (tests/test_decode.py:62-98,217-225)."""
    for m in (compiled, mirror):
        with pytest.raises(ValueError, match="beam_size cannot be 0"):
            m.beam_search(X, "NACGT", beam_size=0)
        with pytest.raises(ValueError, match="beam_cut_threshold must be at least 0.0"):
            m.beam_search(X, "NACGT", beam_cut_threshold=-0.1)
        with pytest.raises(ValueError, match="beam_cut_threshold cannot be more than 0.2"):
            m.beam_search(X, "NACGT", beam_cut_threshold=0.2)
        with pytest.raises(ValueError, match="alphabet size 4 does not match probability matrix inner dimension 5"):
            m.beam_search(X, "NACG")
        with pytest.raises(ValueError, match="Empty alphabet given"):
            m.viterbi_search(X, "")
        with pytest.raises(ValueError, match="alphabet size does not match probability matrix dimensions"):
            m.viterbi_search(X, "NACG")


def test_portable_list_path_gives_identical_objects():
    """csrc/pymodule.cpp fills list[int] paths from worker threads with bulk reference counts -- CPython-with-a-GIL
    internals.  Built with -DFCD_PORTABLE_LISTS=1 (what Py_GIL_DISABLED / the limited API / other interpreters select)
    every entry is stored under the GIL with its own Py_INCREF: the objects must be the same, and so must the counts
    (the shared ints return to their baseline when the results are dropped).  On the emulated kernels."""
    import gc
    import importlib.util
    import sys

    import numpy as np

    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "hipemu"))
    import build as emu_build
    from emu_util import emu_compiled_module
    spec = importlib.util.spec_from_file_location("fcd_hipemu_portable.fast_ctc_decode", emu_build.build_pymodule(portable=True))
    portable = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(portable)
    fast = emu_compiled_module()
    assert portable._portable_lists is True and fast._portable_lists is False
    rng = np.random.default_rng(1)
    x = rng.random((9, 600, 5), dtype=np.float32)
    x /= np.linalg.norm(x, axis=-1, keepdims=True)
    a = portable.beam_search_batch(x, "NACGT", 5, 0.1, paths="list")
    b = fast.beam_search_batch(x, "NACGT", 5, 0.1, paths="list")
    assert a == b and all(type(p) is list and all(type(v) is int for v in p) for _, p in a)
    big = [v for _, p in a for v in p if v > 256]
    some = big[len(big) // 2]
    uses = sum(1 for _, p in a for v in p if v is some)
    held = sum(1 for v in big if v is some)
    before = sys.getrefcount(some)
    del big, a
    gc.collect()
    assert before - sys.getrefcount(some) == uses + held


def test_unknown_tie_order_in_the_environment_is_reported():
    """a typo in FCD_TIE_ORDER must not silently select the other order (csrc/capi.hip)"""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from fast_ctc_decode_amd import _native as nat; print(nat.default_tie_order())"
    env = dict(os.environ, FCD_TIE_ORDER="stabel")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "FCD_TIE_ORDER" in r.stderr and "stabel" in r.stderr and r.stdout.strip() == "pdq178"
    env["FCD_TIE_ORDER"] = "stable"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "stable" and "FCD_TIE_ORDER" not in r.stderr


def test_std_form_of_the_quicksort_replay_comes_from_the_environment():
    """FCD_PDQ178_STD_FORM (include/fcd.h): read once at load time, applied to the device when a handle is created --
    here on the emulated library: a list of the vector file whose permutation depends on the form comes out accordingly"""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(ROOT, "tools", "verify", "pdq178_vectors.json")))
    idx = next(i for i, c in enumerate(doc["cases"]) if "perm_gp" in c and "perm_g" in c and "perm_p" in c)
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, 'tests')\n"
        "from emu_util import emulated_kernels\n"
        "from fast_ctc_decode_amd import _native as nat\n"
        "from test_pdq178 import device_sort, _HostBuf\n"
        "c = json.load(open('tools/verify/pdq178_vectors.json'))['cases'][%d]\n"
        "p = np.array(c['bits'], np.uint32).view(np.float32)\n"
        "with emulated_kernels() as lib:\n"
        "    h = nat.default_handle(0)\n"
        "    out, lens = device_sort(lib, h, [p], _HostBuf, lambda d, shape, dt: d.a.reshape(shape))\n"
        "    got = (out[0, :lens[0]] & np.uint64(0xFFFFFFFF)).astype(np.int64).tolist()\n"
        "    print(lib.fcd_debug_get_pdq178_std_form(), [k for k in ('perm', 'perm_g', 'perm_p', 'perm_gp') if c[k] == got])\n" % idx)
    for value, want in (("", "0 ['perm']"), ("1", "1 ['perm_g']"), ("2", "2 ['perm_p']"), ("3", "3 ['perm_gp']")):
        env = dict(os.environ, FCD_PDQ178_STD_FORM=value)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stderr
        assert r.stdout.strip().splitlines()[-1] == want, (value, r.stdout, r.stderr)
        assert "is not 0, 1, 2 or 3" not in r.stderr
        # (a non-zero form says once that the duplex searches keep form 0: ADVICE r5)
        assert ("applies to the 1-D searches" in r.stderr) == (value not in ("", "0"))
    env = dict(os.environ, FCD_PDQ178_STD_FORM="7")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "is not 0, 1, 2 or 3" in r.stderr and r.stdout.strip().splitlines()[-1] == "0 ['perm']"
