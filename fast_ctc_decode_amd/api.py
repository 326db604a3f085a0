"""Host-side mirror of fast_ctc_decode's Python surface (/root/reference/src/lib.rs:142-628)
over the HIP C ABI (include/fcd.h).

Single-read functions keep the reference's names, argument order, defaults, validation order,
exception types and messages, so they are a drop-in.  The `*_batch` functions are additive:
they decode many reads per launch (numpy -> staged through the C ABI's *_host entry points;
torch tensors on an AMD GPU -> zero-copy through the *_dev entry points on the tensor's
current stream).  All searching happens on the GPU: there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _native as nat

__version__ = "0.3.7"  # the reference version this surface mirrors (Cargo.toml:3)

_cm = None


def _compiled():
    """The compiled host layer (csrc/pymodule.cpp, module `fast_ctc_decode`): the batch functions on HOST inputs
    are its batch functions -- one implementation of chunk streaming, string and path building."""
    global _cm
    if _cm is None:
        import importlib.util
        import os

        from . import build as _build
        nat.load()  # libfcd_hip.so (and torch's HIP runtime before it) first
        path = _build.pymodule_path()
        if not os.path.exists(path):
            _build.build_pymodule()
        spec = importlib.util.spec_from_file_location("fast_ctc_decode", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cm = mod
    return _cm


# ---------------------------------------------------------------------------------------------
# argument conversion / validation, as the PyO3 wrappers do it
# ---------------------------------------------------------------------------------------------
def _seq_to_vec(alphabet):
    """src/lib.rs:143-146: PySequence -> tuple -> str() of every element."""
    try:
        return [str(x) for x in tuple(alphabet)]
    except TypeError:
        raise TypeError("alphabet must be a sequence")


def _as_f32(a, ndim, name):
    """PyO3 extracts &PyArrayN<f32>; anything else is a TypeError (no implicit casts)."""
    if not isinstance(a, np.ndarray):
        raise TypeError("argument '%s': expected numpy.ndarray, got %s" % (name, type(a).__name__))
    if a.dtype != np.float32 or a.ndim != ndim:
        raise TypeError("argument '%s': expected a %d-dimensional float32 array, got %d-dimensional %s"
                        % (name, ndim, a.ndim, a.dtype))
    return a


def _usize(v, name):
    """PyO3's extraction of a `usize` argument: it happens before the function body, i.e. before every check
    of src/lib.rs -- a negative int is an OverflowError, anything but an int a TypeError."""
    if isinstance(v, bool) or not isinstance(v, (int, np.integer)):
        raise TypeError("argument '%s': expected an integer" % name)
    if v < 0:
        raise OverflowError("can't convert negative int to unsigned")
    return int(v)


def _check_beam_args(n_alpha, inner, beam_size, thr):
    """src/lib.rs:331-349 -- the order of these checks is part of the behaviour."""
    _usize(beam_size, "beam_size")
    f32 = np.float32
    max_beam_cut = f32(1.0) / f32(n_alpha) if n_alpha else f32(np.inf)
    if n_alpha != inner:
        raise ValueError("alphabet size %d does not match probability matrix inner dimension %d"
                         % (n_alpha, inner))
    if beam_size == 0:
        raise ValueError("beam_size cannot be 0")
    if f32(thr) < f32(-0.0):
        raise ValueError("beam_cut_threshold must be at least 0.0")
    if f32(thr) >= max_beam_cut:
        raise ValueError("beam_cut_threshold cannot be more than %s" % max_beam_cut)


def _check_greedy_alphabet(n_alpha, inner):
    """src/lib.rs:190-195,227-232,264-269"""
    if n_alpha == 0:
        raise ValueError("Empty alphabet given")
    if n_alpha != inner:
        raise ValueError("alphabet size does not match probability matrix dimensions")


def _raise_status(st):
    if st != nat.ST_OK:
        raise RuntimeError(nat.status_string(st))  # src/lib.rs:363 map_err -> PyRuntimeError


def _estrides(a):
    return [s // a.itemsize for s in a.strides]


def _dense(a):
    """The C ABI's host staging copies one contiguous span; negative strides need a copy."""
    if any(s < 0 for s in a.strides):
        return np.ascontiguousarray(a)
    return a


# ---------------------------------------------------------------------------------------------
# low-level batched calls (host numpy buffers)
# ---------------------------------------------------------------------------------------------
class _HostOut:
    def __init__(self, B, T, want_path=True, want_qual=False, want_amb=False):
        w = max(int(T), 1)
        self.ambiguous = np.zeros((B, 2), np.uint32) if want_amb else None
        self.labels = np.zeros((B, w), np.uint8)
        self.path = np.zeros((B, w), np.uint32) if want_path else None
        self.qual = np.zeros((B, w), np.float32) if want_qual else None
        self.out_len = np.zeros(B, np.uint32)
        self.status = np.zeros(B, np.int32)
        self.res = nat.Result(
            self.labels.ctypes.data, self.path.ctypes.data if want_path else None,
            self.qual.ctypes.data if want_qual else None, self.out_len.ctypes.data,
            self.status.ctypes.data, w, self.ambiguous.ctypes.data if want_amb else None)


def _np_dtype_code(x, input_dtype=None):
    """float32 / float16 arrays carry their type; bfloat16 (numpy has none) travels as uint16 bit patterns with
    input_dtype="bfloat16"."""
    if input_dtype in ("bfloat16", nat.DTYPE_BF16):
        if x.dtype != np.uint16:
            raise TypeError("bfloat16 posteriors are passed as a uint16 array of bit patterns")
        return nat.DTYPE_BF16
    if x.dtype == np.float32:
        return nat.DTYPE_F32
    if x.dtype == np.float16:
        return nat.DTYPE_F16
    raise TypeError("expected a float32 or float16 array, got %s" % x.dtype)


def _host_batch(x, crf, lengths=None, input_dtype=None):
    """x: (B,T,N) or (B,T,S,N) numpy view (float32, float16, or uint16 bfloat16 bits) -> nat.Batch (the caller
    keeps x alive)."""
    st = _estrides(x)
    if crf:
        B, T, S, N = x.shape
        b = nat.Batch(x.ctypes.data, B, T, S, N, st[0], st[1], st[2], st[3], None)
    else:
        B, T, N = x.shape
        b = nat.Batch(x.ctypes.data, B, T, 1, N, st[0], st[1], 0, st[2], None)
    if lengths is not None:
        b.lengths = lengths.ctypes.data
    b.dtype = _np_dtype_code(x, input_dtype)
    return b


def _np_lengths(lengths, B):
    if lengths is None:
        return None
    l = np.ascontiguousarray(np.asarray(lengths), np.int64)
    if l.shape != (B,):
        raise ValueError("lengths must have shape (n_reads,)")
    return l


# ---------------------------------------------------------------------------------------------
# drop-in single-read API
# ---------------------------------------------------------------------------------------------
def _qual_chars(probs, qscale, qbias):
    lib = nat.load()
    return "".join(chr(lib.fcd_phred(float(p), float(qscale), float(qbias))) for p in probs)


_coalescer = None


def set_coalescing(max_batch=256, max_wait_us=0, device=0):
    """Route the per-read `viterbi_search` / `beam_search` / `crf_beam_search` / `crf_greedy_search` calls of ALL threads
    through one coalescer
    (include/fcd.h, csrc/coalesce.hip): calls that are in flight at the same time are decoded by one batched
    launch instead of one single-wavefront launch each.  Results do not change.  `max_batch=0` switches it
    off again.  Not in the reference (which has no batch notion); meant for callers that keep its per-read,
    many-threads calling pattern."""
    global _coalescer
    old, _coalescer = _coalescer, None
    if old is not None:
        old.close()
    if max_batch:
        _coalescer = nat.Coalescer(device, max_batch, max_wait_us)


def coalescing_stats():
    """{'calls', 'launches', 'largest_batch'} of the active coalescer, or None."""
    c = _coalescer
    return None if c is None else c.stats()


def viterbi_search(network_output, alphabet, qstring=False, qscale=1.0, qbias=0.0,
                   collapse_repeats=True):
    """Greedy (best-path) CTC decode.  Mirrors src/lib.rs:170-212 -> search.rs:320-383.

    Returns (str, list[int]): the sequence (with the quality string appended when `qstring`)
    and the row index of every emitted label."""
    x = _as_f32(network_output, 2, "network_output")
    alpha = _seq_to_vec(alphabet)
    _check_greedy_alphabet(len(alpha), x.shape[1])
    if x.shape[0] == 0:
        raise RuntimeError("network_output is empty (the reference asserts and aborts here)")
    x = _dense(x)
    out = _HostOut(1, x.shape[0], want_qual=bool(qstring))
    b = _host_batch(x[None], False)
    co = _coalescer
    if co is not None:
        with co:
            co.check(co.lib.fcd_coalescer_viterbi_search(co.ptr, C.byref(b), int(bool(collapse_repeats)),
                                                         C.byref(out.res)))
    else:
        h = nat.default_handle()
        h.check(h.lib.fcd_viterbi_search_host(h.ptr, C.byref(b), int(bool(collapse_repeats)),
                                              C.byref(out.res)))
    n = int(out.out_len[0])
    seq = "".join(alpha[l] for l in out.labels[0, :n])
    if qstring:
        seq += _qual_chars(out.qual[0, :n], qscale, qbias)
    return seq, [int(p) for p in out.path[0, :n]]


def beam_search(network_output, alphabet, beam_size=5, beam_cut_threshold=0.0,
                collapse_repeats=True):
    """CTC prefix beam search.  Mirrors src/lib.rs:318-365 -> search.rs:159-301."""
    x = _as_f32(network_output, 2, "network_output")
    alpha = _seq_to_vec(alphabet)
    _check_beam_args(len(alpha), x.shape[1], beam_size, beam_cut_threshold)
    x = _dense(x)
    out = _HostOut(1, x.shape[0])
    b = _host_batch(x[None], False)
    co = _coalescer
    if co is not None:
        with co:
            co.check(co.lib.fcd_coalescer_beam_search(co.ptr, C.byref(b), int(beam_size), float(beam_cut_threshold),
                                                      int(bool(collapse_repeats)), C.byref(out.res)))
    else:
        h = nat.default_handle()
        h.check(h.lib.fcd_beam_search_host(h.ptr, C.byref(b), int(beam_size), float(beam_cut_threshold),
                                           int(bool(collapse_repeats)), nat.KERNEL_AUTO,
                                           C.byref(out.res)))
    _raise_status(int(out.status[0]))
    n = int(out.out_len[0])
    return "".join(alpha[l] for l in out.labels[0, :n]), [int(p) for p in out.path[0, :n]]


def crf_beam_search(network_output, init_state, alphabet, beam_size=5, beam_cut_threshold=0.0):
    """Mirrors src/lib.rs:252-286 -> search.rs:38-157 (this wrapper validates only the alphabet)."""
    x = _as_f32(network_output, 3, "network_output")
    init = _as_f32(init_state, 1, "init_state")
    beam_size = _usize(beam_size, "beam_size")
    alpha = _seq_to_vec(alphabet)
    _check_greedy_alphabet(len(alpha), x.shape[2])
    if x.size == 0 or init.size == 0:
        raise RuntimeError("network_output/init_state is empty (the reference asserts and aborts here)")
    if beam_size < 1:
        raise RuntimeError(nat.status_string(nat.ST_RAN_OUT_OF_BEAM))  # truncate(0) -> empty beam
    x = _dense(x)
    init = np.ascontiguousarray(init)
    out = _HostOut(1, x.shape[0])
    b = _host_batch(x[None], True)
    co = _coalescer
    if co is not None:
        with co:
            co.check(co.lib.fcd_coalescer_crf_beam_search(co.ptr, C.byref(b), init.ctypes.data, init.shape[0],
                                                          int(beam_size), float(beam_cut_threshold), C.byref(out.res)))
    else:
        h = nat.default_handle()
        h.check(h.lib.fcd_crf_beam_search_host(h.ptr, C.byref(b), init.ctypes.data, init.shape[0],
                                               init.shape[0], int(beam_size), float(beam_cut_threshold),
                                               C.byref(out.res)))
    _raise_status(int(out.status[0]))
    n = int(out.out_len[0])
    labels = out.labels[0, :n]
    # search.rs:146-156: labels are appended leaf->root and the CHARACTERS reversed at the end
    seq = "".join(alpha[l] for l in labels[::-1])[::-1]
    return seq, [int(p) for p in out.path[0, :n]]


def crf_greedy_search(network_output, init_state, alphabet, qstring=False, qscale=1.0, qbias=0.0):
    """Mirrors src/lib.rs:214-250 -> search.rs:385-423."""
    x = _as_f32(network_output, 3, "network_output")
    init = _as_f32(init_state, 1, "init_state")
    alpha = _seq_to_vec(alphabet)
    _check_greedy_alphabet(len(alpha), x.shape[2])
    if x.size == 0 or init.size == 0:
        raise RuntimeError("network_output/init_state is empty (the reference asserts and aborts here)")
    x = _dense(x)
    init = np.ascontiguousarray(init)
    out = _HostOut(1, x.shape[0], want_qual=bool(qstring))
    b = _host_batch(x[None], True)
    co = _coalescer
    if co is not None:
        with co:
            co.check(co.lib.fcd_coalescer_crf_greedy_search(co.ptr, C.byref(b), init.ctypes.data, init.shape[0],
                                                            C.byref(out.res)))
    else:
        h = nat.default_handle()
        h.check(h.lib.fcd_crf_greedy_search_host(h.ptr, C.byref(b), init.ctypes.data, init.shape[0],
                                                 init.shape[0], C.byref(out.res)))
    _raise_status(int(out.status[0]))
    n = int(out.out_len[0])
    seq = "".join(alpha[l] for l in out.labels[0, :n])
    if qstring:
        seq += _qual_chars(out.qual[0, :n], qscale, qbias)
    return seq, [int(p) for p in out.path[0, :n]]


def crf_greedy_search_batch_raw(network_outputs, init_states, lengths=None, qual=False):
    """(B,T,S,N) posteriors (numpy or torch ROCm tensor) + (B,n_init) initial state scores ->
    BatchResult of search::crf_greedy_search (src/search.rs:385-423) per read."""
    if _is_torch_cuda(network_outputs):
        import torch
        init = torch.as_tensor(init_states, dtype=torch.float32, device=network_outputs.device).contiguous()
        r = _torch_call("fcd_crf_greedy_search_dev", network_outputs, True, lengths,
                        (C.c_void_p(init.data_ptr()), int(init.shape[1]), int(init.shape[1])), want_qual=qual)
        r._keep = r._keep + (init,)
        return r
    x = _stack_host(network_outputs, 4)
    init = np.ascontiguousarray(np.asarray(init_states, np.float32))
    B, T, S, N = x.shape
    if init.ndim != 2 or init.shape[0] != B:
        raise ValueError("init_states must have shape (n_reads, n_init)")
    h = nat.default_handle()
    out = _HostOut(B, T, want_qual=qual)
    b = _host_batch(x, True, _np_lengths(lengths, B))
    h.check(h.lib.fcd_crf_greedy_search_host(h.ptr, C.byref(b), init.ctypes.data, init.shape[1],
                                             init.shape[1], C.byref(out.res)))
    return BatchResult(out.labels, out.path, out.out_len, out.status, out.qual)


def crf_greedy_search_batch(network_outputs, init_states, alphabet, qstring=False, qscale=1.0, qbias=0.0,
                            lengths=None, paths="list"):
    """Batched crf_greedy_search: element i equals crf_greedy_search(network_outputs[i], init_states[i], ...)."""
    if _device_tensor(network_outputs) is None:
        return _compiled().crf_greedy_search_batch(_host_input(network_outputs), np.asarray(init_states, np.float32),
                                                   alphabet, bool(qstring), qscale, qbias, lengths, paths)
    alpha = _seq_to_vec(alphabet)
    _check_greedy_alphabet(len(alpha), network_outputs.shape[-1])
    r = crf_greedy_search_batch_raw(network_outputs, init_states, lengths, qual=qstring).cpu()
    res = r.sequences(alpha, paths=paths if paths is not None else "array")
    if qstring:
        res = [(s + _qual_chars(r.qual[i, :len(p)], qscale, qbias), p) for i, (s, p) in enumerate(res)]
    return res


_DEFAULT_LOGADD = [nat.LOGADD_LOGSUMEXP]


_MODE_CODES = {"logsumexp": nat.LOGADD_LOGSUMEXP, "max": nat.LOGADD_MAX, "logsumexp_glibc235": nat.LOGADD_LOGSUMEXP_GLIBC235,
               nat.LOGADD_LOGSUMEXP: nat.LOGADD_LOGSUMEXP, nat.LOGADD_MAX: nat.LOGADD_MAX,
               nat.LOGADD_LOGSUMEXP_GLIBC235: nat.LOGADD_LOGSUMEXP_GLIBC235}


def set_duplex_logadd_mode(mode):
    """Select the duplex log-space addition: "logsumexp" (the reference built with
    --no-default-features; BASELINE.json's north star; ln / exp / ln_1p correctly rounded), "max" (the reference's
    default `fastexp` feature, whose exp() is identically 0.0 -- what the PyPI wheels compute; SURVEY.md finding 3) or
    "logsumexp_glibc235" (logsumexp on glibc 2.35's expf / logf / log1pf, bit for bit: what the reference computes on
    such a host -- csrc/glibc235_math.h)."""
    _DEFAULT_LOGADD[0] = _MODE_CODES[mode]


def set_tie_order(order):
    """How the beam searches order EQUAL probabilities among more than 20 candidates: "pdq178" (default: what Rust
    1.78's sort_unstable_by -- the reference wheels' toolchain -- leaves them in) or "stable" (ascending node index).
    Process-wide; the same switch as the compiled module's set_tie_order and the C ABI's fcd_set_default_tie_order."""
    nat.set_default_tie_order(order)


def tie_order():
    return nat.default_tie_order()


def set_overlap(streams, device=0):
    """Device-tensor batches only: let successive beam_search_batch_raw / crf_beam_search_batch_raw calls of this thread
    overlap on `streams` (2 .. 8) internal HIP streams of the thread's handle (include/fcd.h, fcd_set_overlap).  A batch is
    as slow as its slowest read and 4096 reads fill half the chip, so independent batches issued back to back finish
    sooner when they share it (BASELINE config 2: 0.94 M -> 1.4 M reads/s; config 3 under the default tie order: 141 k ->
    300 k).  Each call still starts behind everything torch's current stream held when it was made; a result is
    complete when its `.cpu()` / `.sequences()` has run (they join first) or after overlap_join().  0 = off (default)."""
    nat.default_handle(device).set_overlap(streams)


def overlap_join(device=0):
    """torch's current stream waits for every overlapping call made so far (set_overlap)."""
    import torch
    h = nat.default_handle(device)
    h.set_stream(torch.cuda.current_stream(device).cuda_stream)
    h.overlap_join()


import os as _os
if _os.environ.get("FCD_DUPLEX_LOGADD"):   # the same switch the compiled module reads at import
    set_duplex_logadd_mode(_os.environ["FCD_DUPLEX_LOGADD"])


def _check_envelope(envelope, T1):
    """src/lib.rs:445-456"""
    if envelope is None:
        return
    if not isinstance(envelope, np.ndarray) or envelope.dtype != np.uint64 or envelope.ndim != 2:
        raise TypeError("argument 'envelope': expected a 2-dimensional uint64 array")
    if envelope.shape[0] != T1:
        raise ValueError("the lengths of network_output_1 and envelope do not match")
    if envelope.shape[1] != 2:
        raise ValueError("the inner axis of envelope must have size 2")


def _default_envelope(B, T1, T2, lengths_2=None):
    """src/lib.rs:459-468: every row searches the whole of read 2 -- of THAT pair's read 2 when the
    batch is ragged (an upper bound past the pair's own T2 is an out-of-range slice in the reference)."""
    env = np.empty((B, max(T1, 1), 2), np.uint64)
    env[:, :, 0] = 0
    if lengths_2 is None:
        env[:, :, 1] = T2
    else:
        l2 = np.asarray(lengths_2.cpu() if hasattr(lengths_2, "cpu") else lengths_2, np.int64)
        env[:, :, 1] = np.clip(l2, 0, T2).astype(np.uint64)[:, None]
    return env


def beam_search_duplex_batch_raw(network_outputs_1, network_outputs_2, envelopes=None, beam_size=5,
                                 beam_cut_threshold=0.0, collapse_repeats=True, lengths_1=None,
                                 lengths_2=None, logadd_mode=None, count_ambiguous=False):
    """(B,T1,N) and (B,T2,N) posteriors + (B,T1,2) uint64 envelopes -> BatchResult (labels only).
    `count_ambiguous`: also fill BatchResult.ambiguous, the tie counters of the prune (include/fcd.h)."""
    mode = _DEFAULT_LOGADD[0] if logadd_mode is None else _MODE_CODES[logadd_mode]
    if _is_torch_cuda(network_outputs_1):
        import torch
        x1, x2 = network_outputs_1, network_outputs_2
        d1, d2 = _torch_dtype_code(x1), _torch_dtype_code(x2)
        B, T1, N = x1.shape
        T2 = x2.shape[1]
        dev = x1.device
        if envelopes is None:
            env = torch.from_numpy(_default_envelope(B, T1, T2, lengths_2).view(np.int64)).to(dev)
        elif isinstance(envelopes, np.ndarray):
            env = torch.from_numpy(np.ascontiguousarray(envelopes, np.uint64).view(np.int64)).to(dev)
        else:
            env = envelopes.contiguous()  # int64 tensor holding the u64 bit patterns
        h = nat.default_handle(dev.index or 0)
        s1, s2 = x1.stride(), x2.stride()
        b1 = nat.Batch(x1.data_ptr(), B, T1, 1, N, s1[0], s1[1], 0, s1[2], None, d1)
        b2 = nat.Batch(x2.data_ptr(), B, T2, 1, N, s2[0], s2[1], 0, s2[2], None, d2)
        keep = [x1, x2, env]
        if lengths_1 is not None:
            l1 = torch.as_tensor(lengths_1, dtype=torch.int64, device=dev).contiguous()
            b1.lengths = l1.data_ptr()
            keep.append(l1)
        if lengths_2 is not None:
            l2 = torch.as_tensor(lengths_2, dtype=torch.int64, device=dev).contiguous()
            b2.lengths = l2.data_ptr()
            keep.append(l2)
        w = max(int(T1), 1)
        labels = torch.empty((B, w), dtype=torch.uint8, device=dev)
        out_len = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        amb = torch.zeros((B, 2), dtype=torch.int32, device=dev) if count_ambiguous else None
        res = nat.Result(labels.data_ptr(), None, None, out_len.data_ptr(), status.data_ptr(), w,
                         amb.data_ptr() if count_ambiguous else None)
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        h.check(h.lib.fcd_beam_search_duplex_dev(
            h.ptr, C.byref(b1), C.byref(b2), C.c_void_p(env.data_ptr()), int(env.shape[1]),
            int(beam_size), float(beam_cut_threshold), int(bool(collapse_repeats)), int(mode),
            C.byref(res)))
        r = BatchResult(labels, None, out_len, status, ambiguous=amb)
        r._handle, r._keep = h, keep
        if h.overlap:  # (see _torch_call)
            h._inflight.append((keep, labels, out_len, status, amb))
            if len(h._inflight) > 256:
                h.overlap_join()
        return r
    x1 = _stack_host(network_outputs_1, 3)
    x2 = _stack_host(network_outputs_2, 3)
    B, T1, N = x1.shape
    T2 = x2.shape[1]
    if x2.shape[0] != B:
        raise ValueError("both batches must hold the same number of reads")
    env = _default_envelope(B, T1, T2, lengths_2) if envelopes is None else np.ascontiguousarray(envelopes, np.uint64)
    if env.shape[0] != B or env.ndim != 3 or env.shape[2] != 2 or env.shape[1] < T1:
        raise ValueError("envelopes must have shape (n_pairs, T1, 2)")
    h = nat.default_handle()
    out = _HostOut(B, T1, want_path=False, want_amb=count_ambiguous)
    l1, l2 = _np_lengths(lengths_1, B), _np_lengths(lengths_2, B)
    b1, b2 = _host_batch(x1, False, l1), _host_batch(x2, False, l2)
    h.check(h.lib.fcd_beam_search_duplex_host(
        h.ptr, C.byref(b1), C.byref(b2), env.ctypes.data, int(env.shape[1]), int(beam_size),
        float(beam_cut_threshold), int(bool(collapse_repeats)), int(mode), C.byref(out.res)))
    r = BatchResult(out.labels, None, out.out_len, out.status, ambiguous=out.ambiguous)
    r._handle = h
    return r


def beam_search_duplex(network_output_1, network_output_2, alphabet, envelope=None, beam_size=5,
                       beam_cut_threshold=0.0, collapse_repeats=True, *, logadd_mode=None):
    """Mirrors src/lib.rs:401-488 -> duplex.rs:443-650.  Returns the consensus sequence (str).
    `logadd_mode` (keyword-only, not a reference argument) overrides set_duplex_logadd_mode()."""
    x1 = _as_f32(network_output_1, 2, "network_output_1")
    x2 = _as_f32(network_output_2, 2, "network_output_2")
    alpha = _seq_to_vec(alphabet)
    if x1.shape[1] != x2.shape[1]:
        raise ValueError("inner axes of the network outputs do not match")
    _check_beam_args(len(alpha), x1.shape[1], beam_size, beam_cut_threshold)
    _check_envelope(envelope, x1.shape[0])
    if x1.shape[0] == 0:
        raise RuntimeError("network_output_1 is empty (the reference indexes envelope[(0,1)] and aborts)")
    co = _coalescer
    if co is not None:  # concurrent per-pair calls share a launch (set_coalescing; csrc/coalesce.hip)
        mode = _DEFAULT_LOGADD[0] if logadd_mode is None else _MODE_CODES[logadd_mode]
        x1, x2 = _dense(x1), _dense(x2)
        env = (_default_envelope(1, x1.shape[0], x2.shape[0])[0] if envelope is None
               else np.ascontiguousarray(envelope, np.uint64))
        r = _HostOut(1, x1.shape[0], want_path=False)
        b1, b2 = _host_batch(x1[None], False), _host_batch(x2[None], False)
        with co:
            co.check(co.lib.fcd_coalescer_beam_search_duplex(
                co.ptr, C.byref(b1), C.byref(b2), env.ctypes.data, int(beam_size), float(beam_cut_threshold),
                int(bool(collapse_repeats)), int(mode), C.byref(r.res)))
    else:
        env = None if envelope is None else np.ascontiguousarray(envelope)[None]
        r = beam_search_duplex_batch_raw(_dense(x1)[None], _dense(x2)[None], env, beam_size,
                                         beam_cut_threshold, collapse_repeats, logadd_mode=logadd_mode)
    _raise_status(int(r.status[0]))
    n = int(r.out_len[0])
    return "".join(alpha[l] for l in r.labels[0, :n])


def beam_search_duplex_batch(network_outputs_1, network_outputs_2, alphabet, envelopes=None,
                             beam_size=5, beam_cut_threshold=0.0, collapse_repeats=True,
                             lengths_1=None, lengths_2=None, logadd_mode=None):
    """Batched beam_search_duplex -> list[str]."""
    if _device_tensor(network_outputs_1) is None:  # host input: the compiled layer (one host layer, not two)
        mode = int(_DEFAULT_LOGADD[0] if logadd_mode is None else _MODE_CODES[logadd_mode])
        return _compiled().beam_search_duplex_batch(_host_input(network_outputs_1), _host_input(network_outputs_2),
                                                    alphabet, envelopes, beam_size, beam_cut_threshold,
                                                    collapse_repeats, lengths_1, lengths_2, True, mode)
    alpha = _seq_to_vec(alphabet)
    if network_outputs_1.shape[-1] != network_outputs_2.shape[-1]:
        raise ValueError("inner axes of the network outputs do not match")
    _check_beam_args(len(alpha), network_outputs_1.shape[-1], beam_size, beam_cut_threshold)
    r = beam_search_duplex_batch_raw(network_outputs_1, network_outputs_2, envelopes, beam_size,
                                     beam_cut_threshold, collapse_repeats, lengths_1, lengths_2,
                                     logadd_mode).cpu()
    return [s for s, _ in r.sequences(alpha)]


def estimate_envelope_batch(network_outputs_1, network_outputs_2, band=64, lengths_1=None, lengths_2=None):
    """Alignment-band estimator for the duplex searches: (B,T1,N) and (B,T2,N) posteriors ->
    (B,T1,2) envelopes usable as `envelopes=` of beam_search_duplex_batch*.

    NOT a reference function (the reference defaults to the full matrix and only anticipates a
    better default, /root/reference/src/lib.rs:376-378): both reads are decoded greedily
    (viterbi_search), the label sequences are aligned globally on the GPU, matched labels anchor
    read-1 time to read-2 time, and row i becomes [centre(i) - band, centre(i) + band + 1) with the
    rows forced to start at 0, end at T2 and touch (specification: tests/envelope_model.py).
    Device tensors in -> int64 torch tensor holding the u64 bit patterns; numpy in -> uint64 array."""
    if _is_torch_cuda(network_outputs_1):
        import torch
        x1, x2 = network_outputs_1, network_outputs_2
        B, T1 = int(x1.shape[0]), int(x1.shape[1])
        T2 = int(x2.shape[1])
        dev = x1.device
        r1 = viterbi_search_batch_raw(x1, True, lengths_1)
        r2 = viterbi_search_batch_raw(x2, True, lengths_2)
        env = torch.zeros((B, max(T1, 1), 2), dtype=torch.int64, device=dev)
        keep = [r1, r2]
        l1 = l2 = None
        if lengths_1 is not None:
            l1 = torch.as_tensor(lengths_1, dtype=torch.int64, device=dev).contiguous()
        if lengths_2 is not None:
            l2 = torch.as_tensor(lengths_2, dtype=torch.int64, device=dev).contiguous()
        h = r1._handle
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        h.check(h.lib.fcd_duplex_envelope_dev(
            h.ptr, B, r1.labels.data_ptr(), r1.path.data_ptr(), r1.out_len.data_ptr(), int(r1.labels.shape[1]),
            None if l1 is None else l1.data_ptr(), T1,
            r2.labels.data_ptr(), r2.path.data_ptr(), r2.out_len.data_ptr(), int(r2.labels.shape[1]),
            None if l2 is None else l2.data_ptr(), T2, int(band), env.data_ptr(), int(env.shape[1])))
        env._keep = (keep, l1, l2)
        return env
    x1 = _stack_host(network_outputs_1, 3)
    x2 = _stack_host(network_outputs_2, 3)
    B, T1, _ = x1.shape
    T2 = x2.shape[1]
    if x2.shape[0] != B:
        raise ValueError("both batches must hold the same number of reads")
    r1 = viterbi_search_batch_raw(x1, True, lengths_1)
    r2 = viterbi_search_batch_raw(x2, True, lengths_2)
    env = np.zeros((B, max(T1, 1), 2), np.uint64)
    l1, l2 = _np_lengths(lengths_1, B), _np_lengths(lengths_2, B)
    h = nat.default_handle()
    lab1, lab2 = np.ascontiguousarray(r1.labels), np.ascontiguousarray(r2.labels)
    p1, p2 = np.ascontiguousarray(r1.path).view(np.uint32), np.ascontiguousarray(r2.path).view(np.uint32)
    n1, n2 = np.ascontiguousarray(r1.out_len).view(np.uint32), np.ascontiguousarray(r2.out_len).view(np.uint32)
    h.check(h.lib.fcd_duplex_envelope_host(
        h.ptr, B, lab1.ctypes.data, p1.ctypes.data, n1.ctypes.data, int(lab1.shape[1]),
        None if l1 is None else l1.ctypes.data, T1,
        lab2.ctypes.data, p2.ctypes.data, n2.ctypes.data, int(lab2.shape[1]),
        None if l2 is None else l2.ctypes.data, T2, int(band), env.ctypes.data, int(env.shape[1])))
    return env[:, :T1]


def estimate_envelope(network_output_1, network_output_2, band=64):
    """Single pair: (T1,N), (T2,N) posteriors -> (T1,2) uint64 envelope for beam_search_duplex(envelope=...)."""
    x1 = np.asarray(network_output_1)
    x2 = np.asarray(network_output_2)
    if x1.ndim != 2 or x2.ndim != 2:
        raise ValueError("expected two (T, N) matrices")
    if x1.shape[0] == 0:
        return np.zeros((0, 2), np.uint64)
    return estimate_envelope_batch(_dense(x1)[None], _dense(x2)[None], band)[0]


def crf_beam_search_duplex_batch_raw(network_outputs_1, init_states_1, network_outputs_2,
                                     init_states_2, envelopes=None, beam_size=5,
                                     beam_cut_threshold=0.0, lengths_1=None, lengths_2=None,
                                     logadd_mode=None, count_ambiguous=False):
    """(B,T1,S,N) / (B,T2,S,N) posteriors (numpy, or torch ROCm tensors: zero-copy), (B,n_init)
    initial state scores, (B,T1,2) uint64 envelopes -> BatchResult (labels only)."""
    mode = _DEFAULT_LOGADD[0] if logadd_mode is None else _MODE_CODES[logadd_mode]
    if _is_torch_cuda(network_outputs_1):
        import torch
        x1, x2 = network_outputs_1, network_outputs_2
        d1, d2 = _torch_dtype_code(x1), _torch_dtype_code(x2)
        B, T1, S, N = x1.shape
        T2 = x2.shape[1]
        dev = x1.device
        i1 = torch.as_tensor(init_states_1, dtype=torch.float32, device=dev).contiguous()
        i2 = torch.as_tensor(init_states_2, dtype=torch.float32, device=dev).contiguous()
        if x2.shape[0] != B or i1.shape[0] != B or i2.shape[0] != B or i1.ndim != 2 or i2.ndim != 2:
            raise ValueError("all inputs must hold the same number of pairs")
        if envelopes is None:
            env = torch.from_numpy(_default_envelope(B, T1, T2, lengths_2).view(np.int64)).to(dev)
        elif isinstance(envelopes, np.ndarray):
            env = torch.from_numpy(np.ascontiguousarray(envelopes, np.uint64).view(np.int64)).to(dev)
        else:
            env = envelopes.contiguous()
        h = nat.default_handle(dev.index or 0)
        s1, s2 = x1.stride(), x2.stride()
        b1 = nat.Batch(x1.data_ptr(), B, T1, S, N, s1[0], s1[1], s1[2], s1[3], None, d1)
        b2 = nat.Batch(x2.data_ptr(), B, T2, S, N, s2[0], s2[1], s2[2], s2[3], None, d2)
        keep = [x1, x2, env, i1, i2]
        if lengths_1 is not None:
            l1 = torch.as_tensor(lengths_1, dtype=torch.int64, device=dev).contiguous()
            b1.lengths = l1.data_ptr()
            keep.append(l1)
        if lengths_2 is not None:
            l2 = torch.as_tensor(lengths_2, dtype=torch.int64, device=dev).contiguous()
            b2.lengths = l2.data_ptr()
            keep.append(l2)
        w = max(int(T1), 1)
        labels = torch.empty((B, w), dtype=torch.uint8, device=dev)
        out_len = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        amb = torch.zeros((B, 2), dtype=torch.int32, device=dev) if count_ambiguous else None
        res = nat.Result(labels.data_ptr(), None, None, out_len.data_ptr(), status.data_ptr(), w,
                         amb.data_ptr() if count_ambiguous else None)
        h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        h.check(h.lib.fcd_crf_beam_search_duplex_dev(
            h.ptr, C.byref(b1), C.c_void_p(i1.data_ptr()), int(i1.shape[1]), int(i1.shape[1]), C.byref(b2),
            C.c_void_p(i2.data_ptr()), int(i2.shape[1]), int(i2.shape[1]), C.c_void_p(env.data_ptr()),
            int(env.shape[1]), int(beam_size), float(beam_cut_threshold), int(mode), C.byref(res)))
        r = BatchResult(labels, None, out_len, status, ambiguous=amb)
        r._handle, r._keep = h, keep
        if h.overlap:  # (see _torch_call)
            h._inflight.append((keep, labels, out_len, status, amb))
            if len(h._inflight) > 256:
                h.overlap_join()
        return r
    x1 = _stack_host(network_outputs_1, 4)
    x2 = _stack_host(network_outputs_2, 4)
    i1 = np.ascontiguousarray(np.asarray(init_states_1, np.float32))
    i2 = np.ascontiguousarray(np.asarray(init_states_2, np.float32))
    B, T1, S, N = x1.shape
    T2 = x2.shape[1]
    if x2.shape[0] != B or i1.shape[0] != B or i2.shape[0] != B or i1.ndim != 2 or i2.ndim != 2:
        raise ValueError("all inputs must hold the same number of pairs")
    env = _default_envelope(B, T1, T2, lengths_2) if envelopes is None else np.ascontiguousarray(envelopes, np.uint64)
    if env.shape[0] != B or env.ndim != 3 or env.shape[2] != 2 or env.shape[1] < T1:
        raise ValueError("envelopes must have shape (n_pairs, T1, 2)")
    h = nat.default_handle()
    out = _HostOut(B, T1, want_path=False, want_amb=count_ambiguous)
    l1, l2 = _np_lengths(lengths_1, B), _np_lengths(lengths_2, B)
    b1, b2 = _host_batch(x1, True, l1), _host_batch(x2, True, l2)
    h.check(h.lib.fcd_crf_beam_search_duplex_host(
        h.ptr, C.byref(b1), i1.ctypes.data, int(i1.shape[1]), int(i1.shape[1]), C.byref(b2),
        i2.ctypes.data, int(i2.shape[1]), int(i2.shape[1]), env.ctypes.data, int(env.shape[1]),
        int(beam_size), float(beam_cut_threshold), int(mode), C.byref(out.res)))
    r = BatchResult(out.labels, None, out.out_len, out.status, ambiguous=out.ambiguous)
    r._handle = h
    return r


def crf_beam_search_duplex_batch(network_outputs_1, init_states_1, network_outputs_2, init_states_2, alphabet,
                                 envelopes=None, beam_size=5, beam_cut_threshold=0.0, lengths_1=None, lengths_2=None,
                                 logadd_mode=None):
    """Batched crf_beam_search_duplex on HOST arrays -> list[str] (the compiled layer's function)."""
    mode = int(_DEFAULT_LOGADD[0] if logadd_mode is None else _MODE_CODES[logadd_mode])
    return _compiled().crf_beam_search_duplex_batch(_host_input(network_outputs_1), init_states_1,
                                                    _host_input(network_outputs_2), init_states_2, alphabet, envelopes,
                                                    beam_size, beam_cut_threshold, lengths_1, lengths_2, True, mode)


def crf_beam_search_duplex(network_output_1, init_state_1, network_output_2, init_state_2,
                           alphabet, envelope=None, beam_size=5, beam_cut_threshold=0.0, *,
                           logadd_mode=None):
    """Mirrors src/lib.rs:490-578 -> duplex.rs:652-834.  Returns the consensus sequence (str)."""
    x1 = _as_f32(network_output_1, 3, "network_output_1")
    i1 = _as_f32(init_state_1, 1, "init_state_1")
    x2 = _as_f32(network_output_2, 3, "network_output_2")
    i2 = _as_f32(init_state_2, 1, "init_state_2")
    alpha = _seq_to_vec(alphabet)
    if x1.shape[2] != x2.shape[2]:
        raise ValueError("inner axes of the network outputs do not match")
    # src/lib.rs:509-530 (the message quotes shape()[1], as the reference does)
    n_alpha = len(alpha)
    if n_alpha != x1.shape[2]:
        raise ValueError("alphabet size %d does not match probability matrix inner dimension %d"
                         % (n_alpha, x1.shape[1]))
    _check_beam_args(n_alpha, x1.shape[2], beam_size, beam_cut_threshold)
    _check_envelope(envelope, x1.shape[0])
    if x1.shape[1] != x2.shape[1]:
        raise RuntimeError("state axes of the network outputs do not match (the reference asserts and aborts)")
    if x1.shape[0] == 0 or i1.size == 0 or i2.size == 0:
        raise RuntimeError("empty network_output_1 / init_state (the reference aborts here)")
    co = _coalescer
    if co is not None:
        mode = _DEFAULT_LOGADD[0] if logadd_mode is None else _MODE_CODES[logadd_mode]
        x1, x2, i1, i2 = _dense(x1), _dense(x2), np.ascontiguousarray(i1), np.ascontiguousarray(i2)
        env = (_default_envelope(1, x1.shape[0], x2.shape[0])[0] if envelope is None
               else np.ascontiguousarray(envelope, np.uint64))
        r = _HostOut(1, x1.shape[0], want_path=False)
        b1, b2 = _host_batch(x1[None], True), _host_batch(x2[None], True)
        with co:
            co.check(co.lib.fcd_coalescer_crf_beam_search_duplex(
                co.ptr, C.byref(b1), i1.ctypes.data, i1.shape[0], C.byref(b2), i2.ctypes.data, i2.shape[0],
                env.ctypes.data, int(beam_size), float(beam_cut_threshold), int(mode), C.byref(r.res)))
    else:
        env = None if envelope is None else np.ascontiguousarray(envelope)[None]
        r = crf_beam_search_duplex_batch_raw(_dense(x1)[None], np.ascontiguousarray(i1)[None],
                                             _dense(x2)[None], np.ascontiguousarray(i2)[None], env,
                                             beam_size, beam_cut_threshold, logadd_mode=logadd_mode)
    _raise_status(int(r.status[0]))
    n = int(r.out_len[0])
    # src/duplex.rs:825-833: labels appended leaf -> root, then the CHARACTERS reversed
    return "".join(alpha[l] for l in r.labels[0, :n][::-1])[::-1]


# ---------------------------------------------------------------------------------------------
# additive batch API
# ---------------------------------------------------------------------------------------------
def _is_torch_cuda(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and bool(x.is_cuda)


class BatchResult:
    """Raw outcome of a batched search: label indices, paths, lengths and per-read status.
    Arrays are numpy for host inputs, torch tensors (same device) for device inputs."""

    def __init__(self, labels, path, out_len, status, qual=None, ambiguous=None):
        self.labels, self.path, self.out_len, self.status, self.qual = labels, path, out_len, status, qual
        # beam searches with count_ambiguous=True: (n_reads, 2) tie counters per read -- [:, 0] steps with
        # > 20 candidates and an exact tie involving a kept one, [:, 1] steps with an exact tie at ranks
        # 0 / 1 or across the truncation boundary (include/fcd.h, fcd_result.ambiguous)
        self.ambiguous = ambiguous

    def cpu(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.overlap and not isinstance(self.labels, np.ndarray):
            # the search may have run on one of the handle's internal streams (Handle.set_overlap): the copies below
            # are issued on torch's current stream, which waits for every overlapping call first
            import torch
            h.set_stream(torch.cuda.current_stream(self.labels.device).cuda_stream)
            h.overlap_join()

        def c(a):
            return a if a is None or isinstance(a, np.ndarray) else a.cpu().numpy()
        return BatchResult(c(self.labels), c(self.path), c(self.out_len), c(self.status), c(self.qual),
                           c(self.ambiguous))

    def sequences(self, alphabet, raise_on_error=True, paths="list"):
        """-> list of (str, path) per read, exactly what the single-read functions return.

        paths="list" (default) gives list[int] like the reference; building millions of Python ints
        dominates large batches (4096 reads x ~2000 labels: ~150 ms against a 5 ms kernel), so
        paths="array" returns int32 numpy views into the result instead, and paths=None skips them."""
        if paths not in ("list", "array", None):
            raise ValueError("paths must be 'list', 'array' or None")
        r = self.cpu()
        alpha = _seq_to_vec(alphabet)
        ok = np.ones(len(r.out_len), bool) if r.status is None else (np.asarray(r.status) == nat.ST_OK)
        if raise_on_error and not ok.all():
            i = int(np.flatnonzero(~ok)[0])
            raise RuntimeError("read %d: %s" % (i, nat.status_string(int(r.status[i]))))
        lens = np.asarray(r.out_len).astype(np.int64)
        labels = np.asarray(r.labels)
        single = all(len(a) == 1 and ord(a) < 128 for a in alpha)
        if single:
            # one vectorised table lookup for the whole batch, then a slice + decode per read
            lut = np.zeros(256, np.uint8)
            lut[:len(alpha)] = np.frombuffer("".join(alpha).encode("ascii"), np.uint8)
            chars = lut[labels]
        else:
            table = np.array(alpha, dtype=object)
        out = []
        for i in range(len(lens)):
            if not ok[i]:
                out.append(None)
                continue
            n = int(lens[i])
            if single:
                seq = chars[i, :n].tobytes().decode("ascii")
            else:
                seq = "".join(table[labels[i, :n]]) if n else ""
            if r.path is None or paths is None:
                pth = None
            elif paths == "array":
                pth = r.path[i, :n]
            else:
                pth = r.path[i, :n].tolist()
            out.append((seq, pth))
        return out


def _torch_dtype_code(x):
    import torch
    code = {torch.float32: nat.DTYPE_F32, torch.float16: nat.DTYPE_F16, torch.bfloat16: nat.DTYPE_BF16}.get(x.dtype)
    if code is None:
        raise TypeError("device posteriors must be float32, float16 or bfloat16")
    return code


def _torch_call(fn_name, x, crf, lengths, extra_args, want_qual=False, want_path=True,
                need_status=True, handle=None, want_amb=False):
    import torch

    # half-precision posteriors (what basecaller networks emit) are read as they are: the kernels convert in
    # registers while loading -- exactly, so the result is the reference's on the upcast matrix -- and compute in f32
    dcode = _torch_dtype_code(x)
    dev = x.device.index or 0
    h = handle if handle is not None else nat.default_handle(dev)
    if crf:
        B, T, S, N = x.shape
        st = x.stride()
        b = nat.Batch(x.data_ptr(), B, T, S, N, st[0], st[1], st[2], st[3], None, dcode)
    else:
        B, T, N = x.shape
        st = x.stride()
        b = nat.Batch(x.data_ptr(), B, T, 1, N, st[0], st[1], 0, st[2], None, dcode)
    if lengths is not None:
        lengths = torch.as_tensor(lengths, dtype=torch.int64, device=x.device).contiguous()
        b.lengths = lengths.data_ptr()
    w = max(int(T), 1)
    labels = torch.empty((B, w), dtype=torch.uint8, device=x.device)
    path = torch.empty((B, w), dtype=torch.int32, device=x.device) if want_path else None
    qual = torch.empty((B, w), dtype=torch.float32, device=x.device) if want_qual else None
    # (the beam kernels write out_len and status of EVERY read on every way out -- no fill kernels in front of them, which
    # under Handle.set_overlap would wait for a free wavefront slot behind the internal streams' high-priority launches)
    beam = fn_name in ("fcd_beam_search_dev", "fcd_crf_beam_search_dev", "fcd_crf_beam_search_dev_k")
    meta = (torch.empty if beam else torch.zeros)((2, B), dtype=torch.int32, device=x.device)
    out_len, status = meta[0], meta[1]
    amb = torch.zeros((B, 2), dtype=torch.int32, device=x.device) if want_amb else None
    res = nat.Result(labels.data_ptr(), path.data_ptr() if want_path else None,
                     qual.data_ptr() if want_qual else None, out_len.data_ptr(),
                     status.data_ptr(), w, amb.data_ptr() if want_amb else None)
    h.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
    fn = getattr(h.lib, fn_name)
    h.check(fn(h.ptr, C.byref(b), *extra_args, C.byref(res)))
    r = BatchResult(labels, path, out_len, status, qual, amb)
    r._handle = h
    r._keep = (x, lengths)
    if h.overlap:
        # Handle.set_overlap: the call may still run on an internal stream when these tensors lose their last
        # reference -- the handle keeps them until the next overlap_join() (BatchResult.cpu() joins by itself)
        h._inflight.append((r._keep, labels, path, qual, out_len, status, amb))
        if len(h._inflight) > 256:  # (a caller that never joins: bound what is kept -- the current stream waits, they go)
            h.overlap_join()
    return r


def _stack_host(x, ndim):
    if isinstance(x, np.ndarray):
        a = x
    else:
        a = np.stack([np.asarray(v) for v in x])
    if a.dtype not in (np.float32, np.float16, np.uint16) or a.ndim != ndim:
        raise TypeError("expected a float32 (or float16 / bfloat16-bits uint16) array of rank %d" % ndim)
    return _dense(a)


def _ragged(network_outputs, lengths, ndim):
    """A list/tuple of per-read arrays of different lengths -> (padded batch, lengths).
    Reads are padded with zeros to the longest one; the kernels never look past lengths[i]."""
    if isinstance(network_outputs, (list, tuple)) and len(network_outputs) > 0 and lengths is None \
            and all(isinstance(v, np.ndarray) for v in network_outputs):
        Ts = [v.shape[0] for v in network_outputs]
        if len(set(Ts)) > 1:
            tail = network_outputs[0].shape[1:]
            if any(v.dtype != np.float32 or v.ndim != ndim - 1 or v.shape[1:] != tail for v in network_outputs):
                raise TypeError("expected float32 arrays of rank %d with equal inner shapes" % (ndim - 1))
            out = np.zeros((len(Ts), max(Ts)) + tail, np.float32)
            for i, v in enumerate(network_outputs):
                out[i, :Ts[i]] = v
            return out, np.asarray(Ts, np.int64)
    return network_outputs, lengths


def _host_input(x):
    """What the compiled batch functions take: one ndarray, or a list of per-read ndarrays."""
    if isinstance(x, np.ndarray):
        return x
    return [np.asarray(v) for v in x]


def _device_tensor(x):
    """torch ROCm tensors pass through; any other device array speaking DLPack (e.g. CuPy) is
    wrapped zero-copy.  Host objects return None."""
    if _is_torch_cuda(x):
        return x
    if hasattr(x, "__dlpack__") and not isinstance(x, np.ndarray) and hasattr(x, "__dlpack_device__"):
        try:
            dev_type = int(x.__dlpack_device__()[0])
        except Exception:
            return None
        if dev_type in (2, 10):  # kDLCUDA, kDLROCM
            import torch
            return torch.from_dlpack(x)
    return None


def beam_search_batch_raw(network_outputs, beam_size=5, beam_cut_threshold=0.0,
                          collapse_repeats=True, lengths=None, kernel=nat.KERNEL_AUTO, handle=None,
                          count_ambiguous=False, input_dtype=None):
    """Decode a (B,T,N) batch with search::beam_search semantics; returns a BatchResult.
    `handle` (device tensors only): an explicit fast_ctc_decode_amd._native.Handle -- one per
    concurrent torch stream, since a handle owns the tree-arena workspace its kernels use.
    `count_ambiguous`: also fill BatchResult.ambiguous (instrumented kernels; include/fcd.h)."""
    dev_x = _device_tensor(network_outputs)
    if dev_x is not None:
        network_outputs = dev_x
        return _torch_call("fcd_beam_search_dev", network_outputs, False, lengths,
                           (int(beam_size), float(beam_cut_threshold),
                            int(bool(collapse_repeats)), int(kernel)), handle=handle,
                           want_amb=count_ambiguous)
    network_outputs, lengths = _ragged(network_outputs, lengths, 3)
    x = _stack_host(network_outputs, 3)
    B, T, N = x.shape
    h = nat.default_handle()
    out = _HostOut(B, T, want_amb=count_ambiguous)
    l = _np_lengths(lengths, B)
    b = _host_batch(x, False, l, input_dtype)
    h.check(h.lib.fcd_beam_search_host(h.ptr, C.byref(b), int(beam_size), float(beam_cut_threshold),
                                       int(bool(collapse_repeats)), int(kernel), C.byref(out.res)))
    return BatchResult(out.labels, out.path, out.out_len, out.status, ambiguous=out.ambiguous)


def beam_search_batch(network_outputs, alphabet, beam_size=5, beam_cut_threshold=0.0,
                      collapse_repeats=True, lengths=None, kernel=nat.KERNEL_AUTO, paths="list"):
    """Batched beam_search: element i equals beam_search(network_outputs[i][:lengths[i]], ...).
    paths="array" returns the paths as numpy arrays instead of list[int] (see BatchResult.sequences)."""
    if _device_tensor(network_outputs) is None:  # host input: the compiled layer streams result chunks
        return _compiled().beam_search_batch(_host_input(network_outputs), alphabet, beam_size, beam_cut_threshold,
                                             collapse_repeats, lengths, paths, int(kernel))
    alpha = _seq_to_vec(alphabet)
    _check_beam_args(len(alpha), network_outputs.shape[-1], beam_size, beam_cut_threshold)
    r = beam_search_batch_raw(network_outputs, beam_size, beam_cut_threshold, collapse_repeats,
                              lengths, kernel)
    return r.sequences(alpha, paths=paths)


def viterbi_search_batch_raw(network_outputs, collapse_repeats=True, lengths=None, qual=False, input_dtype=None):
    dev_x = _device_tensor(network_outputs)
    if dev_x is not None:
        return _torch_call("fcd_viterbi_search_dev", dev_x, False, lengths,
                           (int(bool(collapse_repeats)),), want_qual=qual)
    network_outputs, lengths = _ragged(network_outputs, lengths, 3)
    x = _stack_host(network_outputs, 3)
    B, T, N = x.shape
    h = nat.default_handle()
    out = _HostOut(B, T, want_qual=qual)
    l = _np_lengths(lengths, B)
    b = _host_batch(x, False, l, input_dtype)
    h.check(h.lib.fcd_viterbi_search_host(h.ptr, C.byref(b), int(bool(collapse_repeats)),
                                          C.byref(out.res)))
    return BatchResult(out.labels, out.path, out.out_len, out.status, out.qual)


def viterbi_search_batch(network_outputs, alphabet, qstring=False, qscale=1.0, qbias=0.0,
                         collapse_repeats=True, lengths=None, paths="list"):
    if _device_tensor(network_outputs) is None:
        return _compiled().viterbi_search_batch(_host_input(network_outputs), alphabet, bool(qstring), qscale, qbias,
                                                collapse_repeats, lengths, paths)
    alpha = _seq_to_vec(alphabet)
    _check_greedy_alphabet(len(alpha), network_outputs.shape[-1])
    r = viterbi_search_batch_raw(network_outputs, collapse_repeats, lengths, qual=qstring).cpu()
    res = r.sequences(alpha, paths=paths if paths is not None else "array")
    if qstring:
        res = [(s + _qual_chars(r.qual[i, :len(p)], qscale, qbias), p) for i, (s, p) in enumerate(res)]
    return res


def crf_beam_search_batch_raw(network_outputs, init_states, beam_size=5, beam_cut_threshold=0.0,
                              lengths=None, kernel=nat.KERNEL_AUTO, count_ambiguous=False):
    """(B,T,S,N) posteriors + (B,n_init) initial state scores -> BatchResult."""
    if _is_torch_cuda(network_outputs):
        import torch
        init = torch.as_tensor(init_states, dtype=torch.float32,
                               device=network_outputs.device).contiguous()
        r = _torch_call("fcd_crf_beam_search_dev_k", network_outputs, True, lengths,
                        (C.c_void_p(init.data_ptr()), int(init.shape[1]), int(init.shape[1]),
                         int(beam_size), float(beam_cut_threshold), int(kernel)),
                        want_amb=count_ambiguous)
        r._keep = r._keep + (init,)
        return r
    x = _stack_host(network_outputs, 4)
    init = np.ascontiguousarray(np.asarray(init_states, np.float32))
    B, T, S, N = x.shape
    if init.shape[0] != B or init.ndim != 2:
        raise ValueError("init_states must have shape (n_reads, n_init)")
    h = nat.default_handle()
    out = _HostOut(B, T, want_amb=count_ambiguous)
    l = _np_lengths(lengths, B)
    b = _host_batch(x, True, l)
    h.check(h.lib.fcd_crf_beam_search_host_k(h.ptr, C.byref(b), init.ctypes.data, init.shape[1],
                                             init.shape[1], int(beam_size), float(beam_cut_threshold),
                                             int(kernel), C.byref(out.res)))
    return BatchResult(out.labels, out.path, out.out_len, out.status, ambiguous=out.ambiguous)


def crf_beam_search_batch(network_outputs, init_states, alphabet, beam_size=5,
                          beam_cut_threshold=0.0, lengths=None):
    if _device_tensor(network_outputs) is None:
        return _compiled().crf_beam_search_batch(_host_input(network_outputs), np.asarray(init_states, np.float32),
                                                 alphabet, beam_size, beam_cut_threshold, lengths)
    alpha = _seq_to_vec(alphabet)
    _check_greedy_alphabet(len(alpha), network_outputs.shape[-1])
    r = crf_beam_search_batch_raw(network_outputs, init_states, beam_size, beam_cut_threshold,
                                  lengths).cpu()
    out = []
    for i, item in enumerate(r.sequences(alpha)):
        n = int(r.out_len[i])
        labels = r.labels[i, :n]
        out.append(("".join(alpha[l] for l in labels[::-1])[::-1], item[1]))
    return out
