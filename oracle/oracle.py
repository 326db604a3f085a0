"""ctypes front-end of the CPU oracle (oracle/fcd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  It gives the C restatement the same Python
surface as the reference's PyO3 module (/root/reference/src/lib.rs:142-628) so the known-answer
tests can be written exactly like the reference's own tests/test_decode.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfcd_oracle.so")

OK, RAN_OUT_OF_BEAM, INCOMPARABLE, INVALID_ENVELOPE, PANIC = 0, 1, 2, 3, 100
LOGSUMEXP, MAXMODE, MATH_CR = 0, 1, 4

# src/lib.rs:46-53
_MESSAGES = {
    RAN_OUT_OF_BEAM: "Ran out of search space (beam_cut_threshold too high)",
    INCOMPARABLE: "Failed to compare values (NaNs in input?)",
    INVALID_ENVELOPE: "Invalid envelope values",
    PANIC: "reference would panic (process abort) on this input",
}


def build():
    """(Re)build libfcd_oracle.so with oracle/Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    src = os.path.join(_HERE, "fcd_oracle.c")
    if (not os.path.exists(_LIB_PATH)) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        build()
    lib = C.CDLL(_LIB_PATH)
    i64, f32, i32 = C.c_int64, C.c_float, C.c_int
    P = C.c_void_p
    lib.fcdo_viterbi_search.argtypes = [P, i64, i64, i64, i64, i32, f32, f32, P, P, P, P]
    lib.fcdo_beam_search.argtypes = [P, i64, i64, i64, i64, i64, f32, i32, P, P, P, P]
    lib.fcdo_beam_search_ex.argtypes = [P, i64, i64, i64, i64, i64, f32, i32, P, P, P, P, P]
    lib.fcdo_beam_search_all_tie_orders.argtypes = [P, i64, i64, i64, i64, i64, f32, i32, P, P, P, i64, P, P, P]
    lib.fcdo_crf_beam_search_ex.argtypes = [P, i64, i64, i64, i64, i64, i64, P, i64, i64, i64, f32, P, P, P, P]
    lib.fcdo_beam_search_batch_ex.argtypes = [P, i64, i64, i64, i64, f32, i32, P, P, P, P, P, i32, i64]
    lib.fcdo_crf_beam_search.argtypes = [P, i64, i64, i64, i64, i64, i64, P, i64, i64, i64, f32, P, P, P]
    lib.fcdo_crf_greedy_search.argtypes = [P, i64, i64, i64, i64, i64, i64, P, i64, i64, f32, f32, P, P, P, P]
    lib.fcdo_beam_search_duplex.argtypes = [P, i64, i64, i64, P, i64, i64, i64, i64, P, i64, i64, i64, f32, i32, i32, P, P]
    lib.fcdo_crf_beam_search_duplex.argtypes = [P, i64, P, P, i64, i64, P, i64, P, P, i64, i64, i64, i64, P, i64, i64, i64, f32, i32, P, P]
    lib.fcdo_beam_search_batch.argtypes = [P, i64, i64, i64, i64, f32, i32, P, P, P, P, i32, i64]
    lib.fcdo_viterbi_batch.argtypes = [P, i64, i64, i64, i32, P, P, P, i32]
    lib.fcdo_phred.argtypes = [f32, f32, f32]
    lib.fcdo_phred.restype = C.c_char
    lib.fcdo_tree_new.argtypes = [i64]
    lib.fcdo_tree_new.restype = P
    lib.fcdo_tree_free.argtypes = [P]
    lib.fcdo_tree_add_node.argtypes = [P, C.c_int32, i64, i64]
    lib.fcdo_tree_add_node.restype = C.c_int32
    lib.fcdo_tree_get_child.argtypes = [P, C.c_int32, i64]
    lib.fcdo_tree_get_child.restype = C.c_int32
    lib.fcdo_tree_label.argtypes = [P, C.c_int32]
    lib.fcdo_tree_label.restype = i64
    lib.fcdo_tree_parent.argtypes = [P, C.c_int32]
    lib.fcdo_tree_parent.restype = C.c_int32
    lib.fcdo_tree_data.argtypes = [P, C.c_int32]
    lib.fcdo_tree_data.restype = i64
    lib.fcdo_tree_len.argtypes = [P]
    lib.fcdo_tree_len.restype = i64
    lib.fcdo_secondary_get.argtypes = [P, i64, i64, i64, P, P]
    lib.fcdo_secondary_update_max.argtypes = [P, i64, i64, i64, i64, i32]
    lib.fcdo_secondary_update_max.restype = f32
    lib.fcdo_logspace_add.argtypes = [f32, f32, i32]
    lib.fcdo_logspace_add.restype = f32
    lib.fcdo_logspace_add_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, i32]
    lib.fcdo_logspace_add_batch.restype = None
    lib.fcdo_libm_apply.argtypes = [i32, C.c_void_p, C.c_void_p, C.c_int64]
    lib.fcdo_libm_apply.restype = None
    lib.fcdo_logadd_calls.argtypes = [i32]
    lib.fcdo_duplex_tie_steps.argtypes = [C.POINTER(i64), i32]
    lib.fcdo_duplex_tie_steps.restype = None
    lib.fcdo_logadd_calls.restype = i64
    lib.fcdo_duplex_last_ambiguous.argtypes = [C.POINTER(i64)]
    lib.fcdo_duplex_last_ambiguous.restype = None
    lib.fcdo_set_unstable_sort.argtypes = [i32]
    lib.fcdo_set_unstable_sort.restype = None
    lib.fcdo_get_unstable_sort.restype = i32
    lib.fcdo_set_pdq_std_form.argtypes = [i32]
    lib.fcdo_set_pdq_std_form.restype = None
    lib.fcdo_get_pdq_std_form.restype = i32
    lib.fcdo_pdq_break_patterns_calls.argtypes = [i32]
    lib.fcdo_pdq_break_patterns_calls.restype = i64
    lib.fcdo_pdq_partial_shift_calls.argtypes = [i32]
    lib.fcdo_pdq_partial_shift_calls.restype = i64
    lib.fcdo_set_external_recurse.argtypes = [P]
    lib.fcdo_set_external_recurse.restype = None
    lib.fcdo_test_pdqsort.argtypes = [P, P, i64]
    lib.fcdo_test_pdqsort.restype = None
    return lib


lib = _load()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _seq_to_vec(alphabet):
    # src/lib.rs:143-146: every element goes through str()
    return [str(x) for x in tuple(alphabet)]


def _f32(a, ndim, name):
    # PyO3 extracts &PyArrayN<f32>: anything else is a TypeError
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != ndim:
        raise TypeError("%s must be a %dD numpy float32 array" % (name, ndim))
    return a


def _estrides(a):
    return [s // a.itemsize for s in a.strides]


def _raise(status):
    if status != OK:
        raise RuntimeError(_MESSAGES.get(status, "status %d" % status))


def phred(prob, qscale=1.0, qbias=0.0):
    return lib.fcdo_phred(prob, qscale, qbias).decode("latin1")


def viterbi_search_raw(network_output, collapse_repeats=True, qscale=1.0, qbias=0.0):
    """-> (labels[int32], path[int64], quals[uint32 code points])"""
    x = network_output
    T, N = x.shape
    rs, cs = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    quals = np.empty(max(T, 1), np.uint32)
    n = C.c_int64(0)
    st = lib.fcdo_viterbi_search(_ptr(x), T, N, rs, cs, int(collapse_repeats), qscale, qbias,
                                 _ptr(labels), _ptr(path), _ptr(quals), C.byref(n))
    _raise(st)
    return labels[: n.value], path[: n.value], quals[: n.value]


def viterbi_search(network_output, alphabet, qstring=False, qscale=1.0, qbias=0.0,
                   collapse_repeats=True):
    """src/lib.rs:170-212"""
    x = _f32(network_output, 2, "network_output")
    alphabet = _seq_to_vec(alphabet)
    if len(alphabet) == 0:
        raise ValueError("Empty alphabet given")
    if len(alphabet) != x.shape[1]:
        raise ValueError("alphabet size does not match probability matrix dimensions")
    labels, path, quals = viterbi_search_raw(x, collapse_repeats, qscale, qbias)
    seq = "".join(alphabet[l] for l in labels)
    if qstring:
        seq += "".join(chr(q) for q in quals)
    return seq, [int(p) for p in path]


def beam_search_raw(network_output, beam_size, beam_cut_threshold, collapse_repeats=True):
    """-> (status, labels[int32], path[int64], n_nodes)"""
    x = network_output
    T, N = x.shape
    rs, cs = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    n = C.c_int64(0)
    nn = C.c_int64(0)
    st = lib.fcdo_beam_search(_ptr(x), T, N, rs, cs, beam_size, beam_cut_threshold,
                              int(collapse_repeats), _ptr(labels), _ptr(path), C.byref(n),
                              C.byref(nn))
    return st, labels[: n.value], path[: n.value], nn.value


def beam_search_ambiguous(network_output, beam_size, beam_cut_threshold, collapse_repeats=True):
    """-> (status, labels, path, (n_gt20_kept_ties, n_critical_ties)): fcdo_beam_search_ex, the search plus
    the two tie counters of SURVEY 8a A4 (semantics in fcd_oracle.h)."""
    x = network_output
    T, N = x.shape
    rs, cs = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    n, nn = C.c_int64(0), C.c_int64(0)
    na = (C.c_int64 * 2)(0, 0)
    st = lib.fcdo_beam_search_ex(_ptr(x), T, N, rs, cs, beam_size, beam_cut_threshold,
                                 int(collapse_repeats), _ptr(labels), _ptr(path), C.byref(n),
                                 C.byref(nn), na)
    return st, labels[: n.value], path[: n.value], (int(na[0]), int(na[1]))


def beam_search_all_tie_orders(network_output, beam_size, beam_cut_threshold, collapse_repeats=True,
                               max_branches=4096):
    """Replays the search under every resolution of the ties that can change its result (truncation-boundary
    groups at any step, the top after the last step).
    -> (status, labels, path, n_branches, all_equal, complete): all_equal and complete => this is the
    reference's output under ANY tie order."""
    x = network_output
    T, N = x.shape
    rs, cs = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    n, nb, nd, comp = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
    st = lib.fcdo_beam_search_all_tie_orders(_ptr(x), T, N, rs, cs, beam_size, beam_cut_threshold,
                                             int(collapse_repeats), _ptr(labels), _ptr(path), C.byref(n),
                                             max_branches, C.byref(nb), C.byref(nd), C.byref(comp))
    return st, labels[: n.value], path[: n.value], nb.value, nd.value == 1, bool(comp.value)


def crf_beam_search_ambiguous(network_output, init_state, beam_size, beam_cut_threshold):
    """-> (status, labels, path, n_ambiguous) of fcdo_crf_beam_search_ex (labels in sequence order)"""
    x, init = network_output, init_state
    T, S, N = x.shape
    s0, s1, s2 = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    n = C.c_int64(0)
    na = (C.c_int64 * 2)(0, 0)
    st = lib.fcdo_crf_beam_search_ex(_ptr(x), T, S, N, s0, s1, s2, _ptr(init), init.shape[0],
                                     _estrides(init)[0], beam_size, beam_cut_threshold,
                                     _ptr(labels), _ptr(path), C.byref(n), na)
    return st, labels[: n.value], path[: n.value], (int(na[0]), int(na[1]))


def _check_beam_args(n_alpha, inner, beam_size, thr):
    # src/lib.rs:331-349, in this order
    max_beam_cut = np.float32(1.0) / np.float32(n_alpha) if n_alpha else np.float32(np.inf)
    if n_alpha != inner:
        raise ValueError("alphabet size %d does not match probability matrix inner dimension %d"
                         % (n_alpha, inner))
    if beam_size == 0:
        raise ValueError("beam_size cannot be 0")
    if np.float32(thr) < np.float32(-0.0):
        raise ValueError("beam_cut_threshold must be at least 0.0")
    if np.float32(thr) >= max_beam_cut:
        raise ValueError("beam_cut_threshold cannot be more than %s" % max_beam_cut)


def beam_search(network_output, alphabet, beam_size=5, beam_cut_threshold=0.0,
                collapse_repeats=True):
    """src/lib.rs:318-365"""
    x = _f32(network_output, 2, "network_output")
    alphabet = _seq_to_vec(alphabet)
    _check_beam_args(len(alphabet), x.shape[1], beam_size, beam_cut_threshold)
    st, labels, path, _ = beam_search_raw(x, beam_size, beam_cut_threshold, collapse_repeats)
    _raise(st)
    return "".join(alphabet[l] for l in labels), [int(p) for p in path]


def crf_beam_search(network_output, init_state, alphabet, beam_size=5, beam_cut_threshold=0.0):
    """src/lib.rs:252-286 (the wrapper validates only the alphabet)"""
    x = _f32(network_output, 3, "network_output")
    init = _f32(init_state, 1, "init_state")
    alphabet = _seq_to_vec(alphabet)
    if len(alphabet) == 0:
        raise ValueError("Empty alphabet given")
    if x.shape[2] != len(alphabet):
        raise ValueError("alphabet size does not match probability matrix dimensions")
    T, S, N = x.shape
    s0, s1, s2 = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    n = C.c_int64(0)
    st = lib.fcdo_crf_beam_search(_ptr(x), T, S, N, s0, s1, s2, _ptr(init), init.shape[0],
                                  _estrides(init)[0], beam_size, beam_cut_threshold,
                                  _ptr(labels), _ptr(path), C.byref(n))
    _raise(st)
    labels = labels[: n.value]
    # src/search.rs:146-156: labels are pushed leaf->root and the *characters* reversed
    seq = "".join(alphabet[l] for l in labels[::-1])[::-1]
    return seq, [int(p) for p in path[: n.value]]


def crf_greedy_search(network_output, init_state, alphabet, qstring=False, qscale=1.0, qbias=0.0):
    """src/lib.rs:214-250"""
    x = _f32(network_output, 3, "network_output")
    init = _f32(init_state, 1, "init_state")
    alphabet = _seq_to_vec(alphabet)
    if len(alphabet) == 0:
        raise ValueError("Empty alphabet given")
    if x.shape[2] != len(alphabet):
        raise ValueError("alphabet size does not match probability matrix dimensions")
    T, S, N = x.shape
    s0, s1, s2 = _estrides(x)
    labels = np.empty(max(T, 1), np.int32)
    path = np.empty(max(T, 1), np.int64)
    quals = np.empty(max(T, 1), np.uint32)
    n = C.c_int64(0)
    st = lib.fcdo_crf_greedy_search(_ptr(x), T, S, N, s0, s1, s2, _ptr(init), init.shape[0],
                                    _estrides(init)[0], qscale, qbias, _ptr(labels), _ptr(path),
                                    _ptr(quals), C.byref(n))
    _raise(st)
    seq = "".join(alphabet[l] for l in labels[: n.value])
    if qstring:
        seq += "".join(chr(q) for q in quals[: n.value])
    return seq, [int(p) for p in path[: n.value]]


def _envelope(envelope, T1, T2):
    if envelope is None:  # src/lib.rs:459-468
        env = np.empty((T1, 2), np.uint64)
        env[:, 0] = 0
        env[:, 1] = T2
        return env
    if not isinstance(envelope, np.ndarray) or envelope.dtype != np.uint64 or envelope.ndim != 2:
        raise TypeError("envelope must be a 2D numpy uint64 array")
    if envelope.shape[0] != T1:
        raise ValueError("the lengths of network_output_1 and envelope do not match")
    if envelope.shape[1] != 2:
        raise ValueError("the inner axis of envelope must have size 2")
    return envelope


def beam_search_duplex(network_output_1, network_output_2, alphabet, envelope=None, beam_size=5,
                       beam_cut_threshold=0.0, collapse_repeats=True, logadd_mode=LOGSUMEXP):
    """src/lib.rs:401-488.  logadd_mode is not a reference argument: it selects the reference's
    build-time `fastexp` feature (MAXMODE) or --no-default-features (LOGSUMEXP)."""
    x1 = _f32(network_output_1, 2, "network_output_1")
    x2 = _f32(network_output_2, 2, "network_output_2")
    alphabet = _seq_to_vec(alphabet)
    if x1.shape[1] != x2.shape[1]:
        raise ValueError("inner axes of the network outputs do not match")
    _check_beam_args(len(alphabet), x1.shape[1], beam_size, beam_cut_threshold)
    env = _envelope(envelope, x1.shape[0], x2.shape[0])
    T1, N = x1.shape
    labels = np.empty(max(T1, 1), np.int32)
    n = C.c_int64(0)
    r1, c1 = _estrides(x1)
    r2, c2 = _estrides(x2)
    e0, e1 = _estrides(env)
    st = lib.fcdo_beam_search_duplex(_ptr(x1), T1, r1, c1, _ptr(x2), x2.shape[0], r2, c2, N,
                                     _ptr(env), e0, e1, beam_size, beam_cut_threshold,
                                     int(collapse_repeats), logadd_mode, _ptr(labels), C.byref(n))
    _raise(st)
    return "".join(alphabet[l] for l in labels[: n.value])


def crf_beam_search_duplex(network_output_1, init_state_1, network_output_2, init_state_2,
                           alphabet, envelope=None, beam_size=5, beam_cut_threshold=0.0,
                           logadd_mode=LOGSUMEXP):
    """src/lib.rs:490-578"""
    x1 = _f32(network_output_1, 3, "network_output_1")
    x2 = _f32(network_output_2, 3, "network_output_2")
    i1 = _f32(init_state_1, 1, "init_state_1")
    i2 = _f32(init_state_2, 1, "init_state_2")
    alphabet = _seq_to_vec(alphabet)
    if x1.shape[2] != x2.shape[2]:
        raise ValueError("inner axes of the network outputs do not match")
    _check_beam_args(len(alphabet), x1.shape[2], beam_size, beam_cut_threshold)
    env = _envelope(envelope, x1.shape[0], x2.shape[0])
    if x1.shape[1] != x2.shape[1]:
        raise RuntimeError(_MESSAGES[PANIC])  # assert_eq! src/duplex.rs:666
    T1, S, N = x1.shape
    labels = np.empty(max(T1, 1), np.int32)
    n = C.c_int64(0)
    st1 = np.array(_estrides(x1), np.int64)
    st2 = np.array(_estrides(x2), np.int64)
    e0, e1 = _estrides(env)
    st = lib.fcdo_crf_beam_search_duplex(_ptr(x1), T1, _ptr(st1), _ptr(i1), i1.shape[0],
                                         _estrides(i1)[0], _ptr(x2), x2.shape[0], _ptr(st2),
                                         _ptr(i2), i2.shape[0], _estrides(i2)[0], S, N, _ptr(env),
                                         e0, e1, beam_size, beam_cut_threshold, logadd_mode,
                                         _ptr(labels), C.byref(n))
    _raise(st)
    # src/duplex.rs:825-833: pushed leaf->root, characters reversed
    return "".join(alphabet[l] for l in labels[: n.value][::-1])[::-1]


def beam_search_batch(x, beam_size, thr, collapse=True, n_threads=1, n_passes=1, out=None, ambiguous=None):
    """x: (B,T,N) C-contiguous f32 -> (labels (B,T) i32, path (B,T) i64, lens (B,), status (B,)).
    `out` may carry pre-touched output arrays (so that page faults stay out of a timed call).
    `ambiguous`: optional int64 (B, 2) array that receives the per-read tie counters of beam_search_ambiguous."""
    x = np.ascontiguousarray(x, np.float32)
    B, T, N = x.shape
    if out is None:
        out = batch_outputs(B, T)
    labels, path, lens, status = out
    lib.fcdo_beam_search_batch_ex(_ptr(x), B, T, N, beam_size, thr, int(collapse), _ptr(labels),
                                  _ptr(path), _ptr(lens), _ptr(status),
                                  None if ambiguous is None else _ptr(ambiguous), n_threads, n_passes)
    return labels, path, lens, status


def batch_outputs(B, T):
    labels = np.zeros((B, max(T, 1)), np.int32)
    path = np.zeros((B, max(T, 1)), np.int64)
    labels.fill(0)  # touch every page now
    path.fill(0)
    return labels, path, np.zeros(B, np.int64), np.zeros(B, np.int32)


def viterbi_batch(x, collapse=True, n_threads=1):
    x = np.ascontiguousarray(x, np.float32)
    B, T, N = x.shape
    labels = np.zeros((B, max(T, 1)), np.int32)
    path = np.zeros((B, max(T, 1)), np.int64)
    lens = np.zeros(B, np.int64)
    lib.fcdo_viterbi_batch(_ptr(x), B, T, N, int(collapse), _ptr(labels), _ptr(path), _ptr(lens),
                           n_threads)
    return labels, path, lens


def duplex_tie_steps(reset=False):
    """Tie statistics of the duplex searches run since the last reset (not thread-safe): dict with the number of
    pruning steps, of steps with > 20 candidates in which a kept candidate ties with another, of steps with a tie
    across the truncation boundary, and of reads whose final top two candidates tie (src/duplex.rs:620,807)."""
    out = (C.c_int64 * 4)()
    lib.fcdo_duplex_tie_steps(out, 1 if reset else 0)
    return {"steps": out[0], "gt20_kept_tie": out[1], "boundary_tie": out[2], "final_top_tie": out[3]}


class unstable_sort:
    """with unstable_sort("pdqsort"): the beam searches order EQUAL probabilities above 20 candidates the way the
    oracle's restatement of Rust 1.78's pdqsort leaves them (fcd_oracle.c: pinned against a compiled rustc-1.65 std
    except for two routines std changed in 2023, tools/verify/rust165_pdqsort.py) -- the default since round 4,
    like the product's FCD_TIE_PDQ178; with unstable_sort("stable"): ties keep ascending node order (FCD_TIE_STABLE)."""

    def __init__(self, mode):
        self.mode = {"stable": 0, "pdqsort": 1}[mode]

    def __enter__(self):
        self.prev = lib.fcdo_get_unstable_sort()
        lib.fcdo_set_unstable_sort(self.mode)

    def __exit__(self, *exc):
        lib.fcdo_set_unstable_sort(self.prev)
        return False


class pdq_std_form:
    """with pdq_std_form(bits): the two routines of std's pdqsort that changed between rustc 1.65 and 1.78 take their
    EARLIER form (bit 0: break_patterns' generator, bit 1: partial_insertion_sort's shifting; fcd_oracle.c) -- with
    both, the restatement equals the compiled 1.65 routine tools/verify/rust165_pdqsort.py finds in this image.
    0 = Rust 1.78 as recalled: the default, and the kernels' only form."""

    def __init__(self, bits):
        self.bits = int(bits)

    def __enter__(self):
        self.prev = lib.fcdo_get_pdq_std_form()
        lib.fcdo_set_pdq_std_form(self.bits)

    def __exit__(self, *exc):
        lib.fcdo_set_pdq_std_form(self.prev)
        return False


class external_recurse:
    """with external_recurse(address): the quicksort above 20 candidates is run by the routine at `address` -- a compiled
    core::slice::sort::recurse over 24-byte records ordered by their first u64 (tools/verify/rust165_pdqsort.py) --
    instead of the restatement.  Process-wide test hook."""

    def __init__(self, address):
        self.address = address

    def __enter__(self):
        lib.fcdo_set_external_recurse(C.c_void_p(self.address))

    def __exit__(self, *exc):
        lib.fcdo_set_external_recurse(None)
        return False


def pdq_path_counts(reset=False):
    """(break_patterns calls, partial_insertion_sort shifts) of this thread's quicksorts since the last reset"""
    r = 1 if reset else 0
    return int(lib.fcdo_pdq_break_patterns_calls(r)), int(lib.fcdo_pdq_partial_shift_calls(r))


def pdqsort_desc(prob, node):
    """(test hook) -> (prob, node) sorted by descending prob with the pdqsort restatement"""
    p = np.ascontiguousarray(prob, np.float32).copy()
    n = np.ascontiguousarray(node, np.int32).copy()
    lib.fcdo_test_pdqsort(_ptr(p), _ptr(n), len(p))
    return p, n


def duplex_last_ambiguous():
    """(amb0, amb1) of this thread's most recent duplex search -- the duplex twin of BatchResult.ambiguous."""
    out = (C.c_int64 * 2)()
    lib.fcdo_duplex_last_ambiguous(out)
    return int(out[0]), int(out[1])
