"""ctypes binding of the C ABI (include/fcd.h) exported by libfcd_hip.so.

There is NO CPU fallback: if the library is missing it is built with hipcc (cross-compiles
without a GPU); if it cannot be loaded, or no gfx950 device is visible when a search is
requested, the call fails loudly.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfcd_hip.so")

OK = 0
E_INVALID, E_HIP, E_NOMEM, E_UNSUPPORTED, E_NODEVICE = -1, -2, -3, -4, -5
ST_OK, ST_RAN_OUT_OF_BEAM, ST_INCOMPARABLE, ST_INVALID_ENVELOPE, ST_BAD_STATE, ST_INTERNAL = range(6)
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_WAVE, KERNEL_WAVE1, KERNEL_LANE = 0, 1, 2, 3, 4
LOGADD_LOGSUMEXP, LOGADD_MAX, LOGADD_LOGSUMEXP_GLIBC235 = 0, 1, 2
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
TIE_DEFAULT, TIE_PDQ178, TIE_STABLE = -1, 0, 1

# every symbol include/fcd.h declares (tests/test_capi_symbols.py checks the .so against this
# list AND against the header text)
SYMBOLS = [
    "fcd_version", "fcd_device_count", "fcd_create", "fcd_destroy", "fcd_set_stream", "fcd_reset_stream",
    "fcd_synchronize", "fcd_set_overlap", "fcd_overlap_join", "fcd_overlap_join_stream", "fcd_overlap_last_slot", "fcd_overlap_join_slot", "fcd_last_error", "fcd_status_string", "fcd_set_workspace_limit", "fcd_release_workspace",
    "fcd_set_tie_order", "fcd_get_tie_order", "fcd_set_default_tie_order", "fcd_debug_pdq178_sort_dev", "fcd_debug_pdq178_coop_sort_dev", "fcd_debug_pdq178_coop_profile",
    "fcd_debug_set_pdq178_std_form", "fcd_debug_get_pdq178_std_form",
    "fcd_last_kernel_ms", "fcd_timing_reset", "fcd_timing_mean_ms", "fcd_debug_set_first_pass_divisor", "fcd_debug_set_duplex_profile", "fcd_debug_set_duplex_kernel",
    "fcd_viterbi_search_dev", "fcd_viterbi_search_host",
    "fcd_beam_search_dev", "fcd_beam_search_host", "fcd_beam_search_profile_dev",
    "fcd_crf_beam_search_dev", "fcd_crf_beam_search_dev_k", "fcd_crf_beam_search_host",
    "fcd_crf_beam_search_host_k",
    "fcd_crf_greedy_search_dev", "fcd_crf_greedy_search_host",
    "fcd_beam_search_duplex_dev", "fcd_beam_search_duplex_host",
    "fcd_crf_beam_search_duplex_dev", "fcd_crf_beam_search_duplex_host",
    "fcd_duplex_envelope_dev", "fcd_duplex_envelope_host",
    "fcd_logspace_probe_dev", "fcd_debug_glibc235_dev", "fcd_logadd_latency_probe_dev", "fcd_logadd_sweep_dev", "fcd_phred",
    "fcd_packed_result_bytes", "fcd_result_offsets_dev", "fcd_pack_results_dev", "fcd_unpack_results_dev",
    "fcd_coalescer_create", "fcd_coalescer_destroy", "fcd_coalescer_beam_search", "fcd_coalescer_viterbi_search",
    "fcd_coalescer_crf_beam_search", "fcd_coalescer_crf_greedy_search",
    "fcd_coalescer_beam_search_duplex", "fcd_coalescer_crf_beam_search_duplex",
    "fcd_coalescer_stats", "fcd_coalescer_last_error",
    "fcd_comm_unique_id", "fcd_comm_create", "fcd_comm_wrap", "fcd_comm_destroy", "fcd_gather_results_dev",
    "fcd_comm_synchronize", "fcd_unpack_gathered_dev",
    "fcd_viterbi_search_host_begin", "fcd_beam_search_host_begin", "fcd_crf_beam_search_host_begin",
    "fcd_crf_greedy_search_host_begin", "fcd_viterbi_search_host_ptrs_begin", "fcd_beam_search_host_ptrs_begin", "fcd_set_host_pipeline", "fcd_job_chunks", "fcd_job_next", "fcd_job_end",
]
JOB_PATH, JOB_QUAL, JOB_AMBIGUOUS, JOB_DONE = 1, 2, 4, 1


class Batch(C.Structure):
    _fields_ = [
        ("post", C.c_void_p), ("n_reads", C.c_int64), ("T", C.c_int64), ("S", C.c_int64),
        ("N", C.c_int64), ("stride_read", C.c_int64), ("stride_t", C.c_int64),
        ("stride_s", C.c_int64), ("stride_n", C.c_int64), ("lengths", C.c_void_p), ("dtype", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("labels", C.c_void_p), ("path", C.c_void_p), ("qual", C.c_void_p),
        ("out_len", C.c_void_p), ("status", C.c_void_p), ("out_stride", C.c_int64),
        ("ambiguous", C.c_void_p),
    ]


class Chunk(C.Structure):
    _fields_ = [
        ("read_begin", C.c_int64), ("n_reads", C.c_int64), ("out_len", C.c_void_p), ("status", C.c_void_p),
        ("offsets", C.c_void_p), ("labels", C.c_void_p), ("path", C.c_void_p), ("path_bytes", C.c_int),
        ("qual", C.c_void_p), ("ambiguous", C.c_void_p),
    ]


class NativeError(RuntimeError):
    pass


_lib = None
_lib_lock = threading.Lock()


def load():
    """Load (building first if needed) libfcd_hip.so.  Raises if that is impossible."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            from . import build as _build
            _build.build()
        # PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME
        # libamdhip64.so.7).  Two HIP runtimes in one process cannot both own the GPU, and device
        # pointers are only meaningful inside the runtime that made them, so when torch is
        # installed it must be loaded FIRST: our NEEDED libamdhip64.so.7 then binds to torch's.
        if os.environ.get("FCD_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # no silent fallback
            raise NativeError("cannot load %s: %s" % (LIB_PATH, e))
        _lib = bind(lib)
        return _lib


def bind(lib):
    """Declare the argument / result types of every C-ABI entry point on a loaded library."""
    P, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    BP, RP = C.POINTER(Batch), C.POINTER(Result)
    lib.fcd_version.restype = i32
    lib.fcd_device_count.restype = i32
    lib.fcd_create.argtypes = [i32, C.POINTER(P)]
    lib.fcd_destroy.argtypes = [P]
    lib.fcd_set_stream.argtypes = [P, P]
    lib.fcd_reset_stream.argtypes = [P]
    lib.fcd_set_overlap.argtypes = [P, i32]
    lib.fcd_overlap_join.argtypes = [P]
    lib.fcd_overlap_join_stream.argtypes = [P, P]
    lib.fcd_overlap_last_slot.argtypes = [P]
    lib.fcd_overlap_join_slot.argtypes = [P, i32, P]
    lib.fcd_synchronize.argtypes = [P]
    lib.fcd_last_error.argtypes = [P]
    lib.fcd_last_error.restype = C.c_char_p
    lib.fcd_status_string.argtypes = [i32]
    lib.fcd_status_string.restype = C.c_char_p
    lib.fcd_set_workspace_limit.argtypes = [P, i64]
    lib.fcd_release_workspace.argtypes = [P]
    lib.fcd_set_tie_order.argtypes = [P, i32]
    lib.fcd_get_tie_order.argtypes = [P]
    lib.fcd_set_default_tie_order.argtypes = [i32]
    lib.fcd_debug_set_pdq178_std_form.argtypes = [P, i32]
    lib.fcd_debug_get_pdq178_std_form.argtypes = []
    lib.fcd_debug_pdq178_sort_dev.argtypes = [P, P, i64, i64, P]
    lib.fcd_debug_pdq178_coop_sort_dev.argtypes = [P, P, i64, i64, P, i32, i32]
    lib.fcd_debug_pdq178_coop_profile.argtypes = [P, P, i32]
    lib.fcd_debug_set_first_pass_divisor.argtypes = [P, i32]
    lib.fcd_debug_set_duplex_profile.argtypes = [P, P]
    lib.fcd_debug_set_duplex_kernel.argtypes = [P, C.c_int]
    lib.fcd_last_kernel_ms.argtypes = [P]
    lib.fcd_last_kernel_ms.restype = C.c_double
    lib.fcd_timing_reset.argtypes = [P]
    lib.fcd_timing_mean_ms.argtypes = [P, C.POINTER(i64)]
    lib.fcd_timing_mean_ms.restype = C.c_double
    for sfx in ("dev", "host"):
        getattr(lib, "fcd_viterbi_search_" + sfx).argtypes = [P, BP, i32, RP]
        getattr(lib, "fcd_beam_search_" + sfx).argtypes = [P, BP, i64, f32, i32, i32, RP]
        getattr(lib, "fcd_crf_beam_search_" + sfx).argtypes = [P, BP, P, i64, i64, i64, f32, RP]
        getattr(lib, "fcd_crf_greedy_search_" + sfx).argtypes = [P, BP, P, i64, i64, RP]
        getattr(lib, "fcd_beam_search_duplex_" + sfx).argtypes = [
            P, BP, BP, P, i64, i64, f32, i32, i32, RP]
    for sfx in ("dev", "host"):
        getattr(lib, "fcd_crf_beam_search_duplex_" + sfx).argtypes = [
            P, BP, P, i64, i64, BP, P, i64, i64, P, i64, i64, f32, i32, RP]
    lib.fcd_beam_search_profile_dev.argtypes = [P, BP, i64, f32, i32, RP, P]
    lib.fcd_crf_beam_search_dev_k.argtypes = [P, BP, P, i64, i64, i64, f32, i32, RP]
    lib.fcd_crf_beam_search_host_k.argtypes = [P, BP, P, i64, i64, i64, f32, i32, RP]
    for sfx in ("dev", "host"):
        getattr(lib, "fcd_duplex_envelope_" + sfx).argtypes = [
            P, i64, P, P, P, i64, P, i64, P, P, P, i64, P, i64, i64, P, i64]
    lib.fcd_logspace_probe_dev.argtypes = [P, P, P, P, P, i64, i32]
    lib.fcd_debug_glibc235_dev.argtypes = [P, i32, P, P, i64]
    lib.fcd_logadd_latency_probe_dev.argtypes = [P, i32, i32, P, P]
    lib.fcd_logadd_sweep_dev.argtypes = [P, i32, C.c_uint32, C.c_uint32, P]
    lib.fcd_packed_result_bytes.argtypes = [i64, i64, i32]
    lib.fcd_packed_result_bytes.restype = i64
    lib.fcd_result_offsets_dev.argtypes = [P, P, i64, i64, P]
    lib.fcd_pack_results_dev.argtypes = [P, RP, i64, i32, P, P]
    lib.fcd_unpack_results_dev.argtypes = [P, P, i64, P, RP]
    lib.fcd_phred.argtypes = [f32, f32, f32]
    lib.fcd_phred.restype = C.c_uint32
    lib.fcd_coalescer_create.argtypes = [i32, i32, i32, C.POINTER(P)]
    lib.fcd_coalescer_destroy.argtypes = [P]
    lib.fcd_coalescer_beam_search.argtypes = [P, BP, i64, f32, i32, RP]
    lib.fcd_coalescer_viterbi_search.argtypes = [P, BP, i32, RP]
    lib.fcd_coalescer_crf_beam_search.argtypes = [P, BP, P, i64, i64, f32, RP]
    lib.fcd_coalescer_crf_greedy_search.argtypes = [P, BP, P, i64, RP]
    lib.fcd_coalescer_beam_search_duplex.argtypes = [P, BP, BP, P, i64, f32, i32, i32, RP]
    lib.fcd_coalescer_crf_beam_search_duplex.argtypes = [P, BP, P, i64, BP, P, i64, P, i64, f32, i32, RP]
    lib.fcd_coalescer_stats.argtypes = [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    lib.fcd_coalescer_last_error.restype = C.c_char_p
    PP = C.POINTER(P)
    lib.fcd_comm_unique_id.argtypes = [P]
    lib.fcd_comm_create.argtypes = [P, i32, i32, P, PP]
    lib.fcd_comm_wrap.argtypes = [P, P, i32, i32, PP]
    lib.fcd_comm_destroy.argtypes = [P]
    lib.fcd_gather_results_dev.argtypes = [P, RP, i64, P, i32, RP]
    lib.fcd_comm_synchronize.argtypes = [P]
    lib.fcd_unpack_gathered_dev.argtypes = [P, P, i64, i32, P, i64, P, RP, P]
    lib.fcd_viterbi_search_host_begin.argtypes = [P, BP, i32, i32, PP]
    lib.fcd_beam_search_host_begin.argtypes = [P, BP, i64, f32, i32, i32, i32, PP]
    lib.fcd_crf_beam_search_host_begin.argtypes = [P, BP, P, i64, i64, i64, f32, i32, i32, PP]
    lib.fcd_crf_greedy_search_host_begin.argtypes = [P, BP, P, i64, i64, i32, PP]
    lib.fcd_viterbi_search_host_ptrs_begin.argtypes = [P, P, P, i64, i64, i32, i32, i32, PP]
    lib.fcd_beam_search_host_ptrs_begin.argtypes = [P, P, P, i64, i64, i32, i64, f32, i32, i32, i32, PP]
    lib.fcd_set_host_pipeline.argtypes = [P, i32, i64, i64]
    lib.fcd_job_chunks.argtypes = [P, C.POINTER(i64), C.POINTER(i32)]
    lib.fcd_job_next.argtypes = [P, C.POINTER(Chunk)]
    lib.fcd_job_end.argtypes = [P]
    return lib


class Handle:
    """Owns one fcd_handle (device binding, stream, workspace)."""

    def __init__(self, device=0):
        self.lib = load()
        self.ptr = C.c_void_p()
        rc = self.lib.fcd_create(int(device), C.byref(self.ptr))
        if rc != OK:
            raise NativeError(
                "fcd_create(device=%d) failed with %d: no usable gfx950 device -- this library has "
                "no CPU fallback" % (device, rc))
        self.device = int(device)
        self.overlap = 0       # set_overlap
        self._inflight = []    # tensors of overlapping calls not joined yet (api._torch_call)

    def check(self, rc):
        if rc != OK:
            msg = self.lib.fcd_last_error(self.ptr)
            raise NativeError("libfcd_hip error %d: %s" % (rc, msg.decode() if msg else ""))

    def set_stream(self, stream_ptr):
        """Launch on this hipStream_t handle; 0/None is the HIP null stream (torch's default)."""
        self.check(self.lib.fcd_set_stream(self.ptr, C.c_void_p(stream_ptr or None)))

    def reset_stream(self):
        self.check(self.lib.fcd_reset_stream(self.ptr))

    def synchronize(self):
        self.check(self.lib.fcd_synchronize(self.ptr))
        self._inflight = []

    def set_overlap(self, streams):
        """include/fcd.h, fcd_set_overlap: wide-beam device calls go round-robin to `streams` internal streams (2 .. 8),
        each behind the handle's stream as it stood at the call and not behind one another -- the stragglers of a batch
        run under the next batches.  Results are complete after overlap_join() / synchronize(); 0 = stream order."""
        self.check(self.lib.fcd_set_overlap(self.ptr, int(streams)))
        self.overlap = int(streams) if int(streams) >= 2 else 0
        self._inflight = []

    def overlap_join(self, stream_ptr=None):
        """The handle's stream (stream_ptr=None) or the given hipStream_t waits for every overlapping call made so far."""
        if stream_ptr is None:
            self.check(self.lib.fcd_overlap_join(self.ptr))
            self._inflight = []  # (later work on that stream is ordered behind them: the caching allocator may reuse them)
        else:
            self.check(self.lib.fcd_overlap_join_stream(self.ptr, C.c_void_p(stream_ptr)))

    def overlap_last_slot(self):
        """The internal stream the latest overlapping call went to (-1: none); see overlap_join_slot."""
        return int(self.lib.fcd_overlap_last_slot(self.ptr))

    def overlap_join_slot(self, slot, stream_ptr):
        """`stream_ptr` waits for what internal stream `slot` has been given so far (include/fcd.h)."""
        self.check(self.lib.fcd_overlap_join_slot(self.ptr, int(slot), C.c_void_p(stream_ptr)))

    def last_kernel_ms(self):
        return float(self.lib.fcd_last_kernel_ms(self.ptr))

    def timing_reset(self):
        self.check(self.lib.fcd_timing_reset(self.ptr))

    def timing_mean_ms(self):
        """-> (mean kernel ms per search call since timing_reset, number of calls)"""
        n = C.c_int64(0)
        ms = float(self.lib.fcd_timing_mean_ms(self.ptr, C.byref(n)))
        return ms, int(n.value)

    def set_workspace_limit(self, nbytes):
        self.check(self.lib.fcd_set_workspace_limit(self.ptr, int(nbytes)))

    def set_host_pipeline(self, lanes=0, chunk_reads=0, min_bytes=-1):
        """Tuning of the chunked host path (include/fcd.h: fcd_set_host_pipeline)."""
        self.check(self.lib.fcd_set_host_pipeline(self.ptr, int(lanes), int(chunk_reads), int(min_bytes)))

    def set_tie_order(self, order):
        """TIE_PDQ178 / TIE_STABLE for this handle, TIE_DEFAULT to follow the process default (include/fcd.h)."""
        self.check(self.lib.fcd_set_tie_order(self.ptr, int(order)))

    def tie_order(self):
        return int(self.lib.fcd_get_tie_order(self.ptr))

    def set_pdq178_std_form(self, bits):
        """(include/fcd_debug.h) which form of the two routines std changed in 2023 the quicksort replay follows:
        process-wide, written to this handle's device at once; 0 = Rust 1.78 as recalled (default), 3 = rustc 1.65"""
        self.check(self.lib.fcd_debug_set_pdq178_std_form(self.ptr, int(bits)))

    def release_workspace(self):
        """Give the tree arena / staging memory back to the device (the next call allocates afresh)."""
        self.check(self.lib.fcd_release_workspace(self.ptr))

    def close(self):
        if self.ptr:
            self.lib.fcd_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Coalescer:
    """fcd_coalescer (csrc/coalesce.hip): concurrent per-read calls -> batched launches."""

    def __init__(self, device=0, max_batch=256, max_wait_us=0):
        self.lib = load()
        self.ptr = C.c_void_p()
        rc = self.lib.fcd_coalescer_create(int(device), int(max_batch), int(max_wait_us), C.byref(self.ptr))
        if rc != OK:
            raise NativeError("fcd_coalescer_create(device=%d) failed with %d: no usable gfx950 device -- this "
                              "library has no CPU fallback" % (device, rc))
        self._users = 0
        self._users_lock = threading.Lock()

    def check(self, rc):
        if rc != OK:
            msg = self.lib.fcd_coalescer_last_error()
            raise NativeError("libfcd_hip error %d: %s" % (rc, msg.decode() if msg else ""))

    def __enter__(self):  # a call's hold on the coalescer: close() waits for the holders
        with self._users_lock:
            self._users += 1
        return self

    def __exit__(self, *exc):
        with self._users_lock:
            self._users -= 1
        return False

    def stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.check(self.lib.fcd_coalescer_stats(self.ptr, C.byref(a), C.byref(b), C.byref(c)))
        return {"calls": int(a.value), "launches": int(b.value), "largest_batch": int(c.value)}

    def close(self):
        import time
        if self.ptr:
            while True:
                with self._users_lock:
                    if self._users == 0:
                        break
                time.sleep(0.001)
            rc = self.lib.fcd_coalescer_destroy(self.ptr)
            if rc != OK:
                raise NativeError("fcd_coalescer_destroy: calls are still in flight")
            self.ptr = C.c_void_p()


_tls = threading.local()


def default_handle(device=0):
    """One handle per (thread, device): the reference's functions are re-entrant and release the
    GIL (src/lib.rs:199), so concurrent callers must not share a stream/workspace."""
    cache = getattr(_tls, "handles", None)
    if cache is None:
        cache = _tls.handles = {}
    h = cache.get(device)
    if h is None:
        h = cache[device] = Handle(device)
    return h


_TIE_NAMES = {"pdq178": TIE_PDQ178, "stable": TIE_STABLE}


def set_default_tie_order(order):
    """Process-wide order of equal probabilities in the beam searches' prune: "pdq178" (Rust 1.78's
    sort_unstable_by, the default) or "stable" (ascending node index); include/fcd.h, FCD_TIE_*."""
    order = _TIE_NAMES.get(order, order)
    if load().fcd_set_default_tie_order(int(order)) != OK:
        raise ValueError("tie order must be 'pdq178' or 'stable'")


def default_tie_order():
    return "stable" if int(load().fcd_get_tie_order(None)) == TIE_STABLE else "pdq178"


def status_string(st):
    return load().fcd_status_string(int(st)).decode()
