// pymodule.cpp -- the compiled Python module `fast_ctc_decode`, host side above the C ABI.
//
// This is the C++ counterpart of the reference's PyO3 layer (/root/reference/src/lib.rs:142-628;
// no Rust toolchain exists in the build image): same module name, function names, argument names,
// order and defaults, the same validation order and messages, ValueError for argument errors,
// RuntimeError carrying the SearchError text for search failures, TypeError for wrong array
// types, GIL released around the search.  Every search runs on the GPU through include/fcd.h;
// there is no CPU path in here.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <atomic>
#include <charconv>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fcd.h"

namespace py = pybind11;

namespace {

// one handle per host thread: the reference's functions are re-entrant (lib.rs:199 releases the GIL), so
// concurrent callers must not share a stream / workspace.  The handle (stream + device workspace) goes away
// with its thread; at interpreter shutdown the HIP runtime may already be gone, so nothing is torn down then.
struct ThreadHandle {
    fcd_handle *h = nullptr;
    ~ThreadHandle() {
        if (h && Py_IsInitialized()) fcd_destroy(h);
    }
};

fcd_handle *thread_handle() {
    thread_local ThreadHandle th;
    if (!th.h) {
        int rc = fcd_create(0, &th.h);
        if (rc != FCD_OK || !th.h)
            throw std::runtime_error("fast_ctc_decode: no usable gfx950 device (fcd_create failed with " +
                                     std::to_string(rc) + "); this module has no CPU fallback");
    }
    return th.h;
}

// set_coalescing(): when set, per-read viterbi_search / beam_search / crf_beam_search / crf_greedy_search calls (and
// the per-PAIR beam_search_duplex / crf_beam_search_duplex calls) of
// every thread go through it
// (include/fcd.h: concurrent calls share batched launches).  Every call holds a reference to the coalescer that
// was current when it started; a replaced coalescer is destroyed when its LAST call returns -- nobody waits for
// a global count to reach zero, so set_coalescing() cannot starve under a steady stream of calls.
struct CoalescerBox {
    fcd_coalescer *co = nullptr;
    ~CoalescerBox() {
        if (co && Py_IsInitialized()) fcd_coalescer_destroy(co);  // (at interpreter exit HIP may be gone)
    }
};
std::mutex g_coalescer_mu;
std::shared_ptr<CoalescerBox> g_coalescer;
struct CoalescerUse {  // a call's hold on whichever coalescer was current when it started
    std::shared_ptr<CoalescerBox> box;
    fcd_coalescer *co;
    CoalescerUse() {
        std::lock_guard<std::mutex> g(g_coalescer_mu);
        box = g_coalescer;
        co = box ? box->co : nullptr;
    }
};

void check_rc_coalescer(int rc) {
    if (rc != FCD_OK)
        throw std::runtime_error(std::string("libfcd_hip error ") + std::to_string(rc) + ": " +
                                 fcd_coalescer_last_error());
}

void check_rc(fcd_handle *h, int rc) {
    if (rc != FCD_OK)
        throw std::runtime_error(std::string("libfcd_hip error ") + std::to_string(rc) + ": " +
                                 fcd_last_error(h));
}

// lib.rs:143-146: PySequence -> tuple -> str() of every element
std::vector<std::string> seq_to_vec(const py::object &alphabet) {
    py::tuple t;
    try {
        t = py::tuple(alphabet);
    } catch (py::error_already_set &) {
        throw py::type_error("argument 'alphabet': expected a sequence");
    }
    std::vector<std::string> out;
    out.reserve(t.size());
    for (auto item : t) out.push_back(py::str(item).cast<std::string>());
    return out;
}

// PyO3 extracts &PyArrayN<f32>: wrong type / dtype / rank is a TypeError, never a cast
py::array as_f32(const py::object &o, int ndim, const char *name) {
    if (!py::isinstance<py::array>(o))
        throw py::type_error(std::string("argument '") + name + "': expected numpy.ndarray");
    py::array a = py::reinterpret_borrow<py::array>(o);
    if (!a.dtype().is(py::dtype::of<float>()) || a.ndim() != ndim)
        throw py::type_error(std::string("argument '") + name + "': expected a " +
                             std::to_string(ndim) + "-dimensional float32 array");
    for (int d = 0; d < ndim; ++d)
        if (a.strides(d) < 0) return py::array::ensure(a, py::array::c_style);  // staging copies one span
    return a;
}

std::string f32_display(float v) {  // Rust `format!("{}", f32)`: shortest round-trip, never scientific
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

// lib.rs:331-349, in this order
void check_beam_args(size_t n_alpha, py::ssize_t inner, py::ssize_t beam_size, float thr) {
    const float max_beam_cut = 1.0f / (float)n_alpha;
    if ((py::ssize_t)n_alpha != inner)
        throw py::value_error("alphabet size " + std::to_string(n_alpha) +
                              " does not match probability matrix inner dimension " +
                              std::to_string(inner));
    if (beam_size == 0) throw py::value_error("beam_size cannot be 0");
    if (thr < -0.0f) throw py::value_error("beam_cut_threshold must be at least 0.0");
    if (thr >= max_beam_cut)
        throw py::value_error("beam_cut_threshold cannot be more than " + f32_display(max_beam_cut));
}

void check_greedy_alphabet(size_t n_alpha, py::ssize_t inner) {  // lib.rs:190-195
    if (n_alpha == 0) throw py::value_error("Empty alphabet given");
    if ((py::ssize_t)n_alpha != inner)
        throw py::value_error("alphabet size does not match probability matrix dimensions");
}

size_t to_usize(const py::object &o, const char *name) {  // PyO3 usize extraction
    if (py::isinstance<py::bool_>(o) || !py::isinstance<py::int_>(o))
        throw py::type_error(std::string("argument '") + name + "': expected an integer");
    if (o.cast<py::int_>() < py::int_(0)) {  // PyO3: OverflowError, before any ValueError check of lib.rs:331-349
        PyErr_SetString(PyExc_OverflowError, "can't convert negative int to unsigned");
        throw py::error_already_set();
    }
    return o.cast<size_t>();
}

void raise_status(int st) {  // lib.rs:363 map_err -> PyRuntimeError(format!("{}", e))
    if (st != FCD_ST_OK) throw std::runtime_error(fcd_status_string(st));
}

void append_utf8(std::string &s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

std::string reverse_chars(const std::string &s) {  // Rust `.chars().rev().collect()`
    std::string out;
    out.reserve(s.size());
    size_t end = s.size();
    while (end > 0) {
        size_t start = end - 1;
        while (start > 0 && ((unsigned char)s[start] & 0xC0) == 0x80) --start;
        out.append(s, start, end - start);
        end = start;
    }
    return out;
}

struct Out {
    std::vector<uint8_t> labels;
    std::vector<uint32_t> path;
    std::vector<float> qual;
    uint32_t len = 0;
    int32_t status = 0;
    fcd_result res{};
    Out(py::ssize_t T, bool want_path, bool want_qual) {
        const size_t w = T > 0 ? (size_t)T : 1;
        labels.resize(w);
        if (want_path) path.resize(w);
        if (want_qual) qual.resize(w);
        res.labels = labels.data();
        res.path = want_path ? path.data() : nullptr;
        res.qual = want_qual ? qual.data() : nullptr;
        res.out_len = &len;
        res.status = &status;
        res.out_stride = (int64_t)w;
    }
};

fcd_batch batch2(const py::array &a) {  // (T, N) view as a one-read batch, element strides
    fcd_batch b{};
    b.post = static_cast<const float *>(a.data());
    b.n_reads = 1;
    b.T = a.shape(0);
    b.S = 1;
    b.N = a.shape(1);
    b.stride_read = 0;
    b.stride_t = a.strides(0) / 4;
    b.stride_n = a.strides(1) / 4;
    return b;
}

fcd_batch batch3(const py::array &a) {  // (T, S, N)
    fcd_batch b{};
    b.post = static_cast<const float *>(a.data());
    b.n_reads = 1;
    b.T = a.shape(0);
    b.S = a.shape(1);
    b.N = a.shape(2);
    b.stride_t = a.strides(0) / 4;
    b.stride_s = a.strides(1) / 4;
    b.stride_n = a.strides(2) / 4;
    return b;
}

py::list path_list(const Out &o) {
    py::list l;
    for (uint32_t i = 0; i < o.len; ++i) l.append(py::int_((size_t)o.path[i]));
    return l;
}

// ---- viterbi_search: lib.rs:170-212 ----
py::tuple viterbi_search(const py::object &network_output, const py::object &alphabet, bool qstring,
                         float qscale, float qbias, bool collapse_repeats) {
    py::array x = as_f32(network_output, 2, "network_output");
    auto alpha = seq_to_vec(alphabet);
    check_greedy_alphabet(alpha.size(), x.shape(1));
    if (x.shape(0) == 0)
        throw std::runtime_error("network_output is empty (the reference asserts and aborts here)");
    Out o(x.shape(0), true, qstring);
    fcd_batch b = batch2(x);
    int rc;
    CoalescerUse use;
    if (fcd_coalescer *co = use.co) {
        {
            py::gil_scoped_release nogil;
            rc = fcd_coalescer_viterbi_search(co, &b, collapse_repeats ? 1 : 0, &o.res);
        }
        check_rc_coalescer(rc);
    } else {
        fcd_handle *h = thread_handle();
        {
            py::gil_scoped_release nogil;
            rc = fcd_viterbi_search_host(h, &b, collapse_repeats ? 1 : 0, &o.res);
        }
        check_rc(h, rc);
    }
    std::string seq;
    for (uint32_t i = 0; i < o.len; ++i) seq += alpha[o.labels[i]];
    if (qstring)
        for (uint32_t i = 0; i < o.len; ++i) append_utf8(seq, fcd_phred(o.qual[i], qscale, qbias));
    return py::make_tuple(py::str(seq), path_list(o));
}

// ---- beam_search: lib.rs:318-365 ----
py::tuple beam_search(const py::object &network_output, const py::object &alphabet,
                      const py::object &beam_size_o, float beam_cut_threshold, bool collapse_repeats) {
    py::array x = as_f32(network_output, 2, "network_output");
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    check_beam_args(alpha.size(), x.shape(1), (py::ssize_t)beam_size, beam_cut_threshold);
    Out o(x.shape(0), true, false);
    fcd_batch b = batch2(x);
    int rc;
    CoalescerUse use;
    if (fcd_coalescer *co = use.co) {
        {
            py::gil_scoped_release nogil;
            rc = fcd_coalescer_beam_search(co, &b, (int64_t)beam_size, beam_cut_threshold,
                                           collapse_repeats ? 1 : 0, &o.res);
        }
        check_rc_coalescer(rc);
    } else {
        fcd_handle *h = thread_handle();
        {
            py::gil_scoped_release nogil;
            rc = fcd_beam_search_host(h, &b, (int64_t)beam_size, beam_cut_threshold,
                                      collapse_repeats ? 1 : 0, FCD_KERNEL_AUTO, &o.res);
        }
        check_rc(h, rc);
    }
    raise_status(o.status);
    std::string seq;
    for (uint32_t i = 0; i < o.len; ++i) seq += alpha[o.labels[i]];
    return py::make_tuple(py::str(seq), path_list(o));
}

// ---- crf_beam_search: lib.rs:252-286 (validates only the alphabet) ----
py::tuple crf_beam_search(const py::object &network_output, const py::object &init_state,
                          const py::object &alphabet, const py::object &beam_size_o,
                          float beam_cut_threshold) {
    py::array x = as_f32(network_output, 3, "network_output");
    py::array init = py::array::ensure(as_f32(init_state, 1, "init_state"), py::array::c_style);
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    check_greedy_alphabet(alpha.size(), x.shape(2));
    if (x.size() == 0 || init.size() == 0)
        throw std::runtime_error("network_output/init_state is empty (the reference asserts and aborts here)");
    if (beam_size == 0) raise_status(FCD_ST_RAN_OUT_OF_BEAM);  // truncate(0) empties the beam (search.rs:133-137)
    Out o(x.shape(0), true, false);
    fcd_batch b = batch3(x);
    int rc;
    CoalescerUse use;
    if (fcd_coalescer *co = use.co) {
        {
            py::gil_scoped_release nogil;
            rc = fcd_coalescer_crf_beam_search(co, &b, static_cast<const float *>(init.data()), init.shape(0),
                                               (int64_t)beam_size, beam_cut_threshold, &o.res);
        }
        check_rc_coalescer(rc);
    } else {
        fcd_handle *h = thread_handle();
        {
            py::gil_scoped_release nogil;
            rc = fcd_crf_beam_search_host(h, &b, static_cast<const float *>(init.data()), init.shape(0),
                                          init.shape(0), (int64_t)beam_size, beam_cut_threshold, &o.res);
        }
        check_rc(h, rc);
    }
    raise_status(o.status);
    // search.rs:146-156: labels appended leaf -> root, then the CHARACTERS are reversed
    std::string rev;
    for (uint32_t i = o.len; i > 0; --i) rev += alpha[o.labels[i - 1]];
    return py::make_tuple(py::str(reverse_chars(rev)), path_list(o));
}

// ---- crf_greedy_search: lib.rs:214-250 ----
py::tuple crf_greedy_search(const py::object &network_output, const py::object &init_state,
                            const py::object &alphabet, bool qstring, float qscale, float qbias) {
    py::array x = as_f32(network_output, 3, "network_output");
    py::array init = py::array::ensure(as_f32(init_state, 1, "init_state"), py::array::c_style);
    auto alpha = seq_to_vec(alphabet);
    check_greedy_alphabet(alpha.size(), x.shape(2));
    if (x.size() == 0 || init.size() == 0)
        throw std::runtime_error("network_output/init_state is empty (the reference asserts and aborts here)");
    Out o(x.shape(0), true, true);
    fcd_batch b = batch3(x);
    int rc;
    CoalescerUse use;
    if (fcd_coalescer *co = use.co) {
        {
            py::gil_scoped_release nogil;
            rc = fcd_coalescer_crf_greedy_search(co, &b, static_cast<const float *>(init.data()), init.shape(0), &o.res);
        }
        check_rc_coalescer(rc);
    } else {
        fcd_handle *h = thread_handle();
        {
            py::gil_scoped_release nogil;
            rc = fcd_crf_greedy_search_host(h, &b, static_cast<const float *>(init.data()), init.shape(0),
                                            init.shape(0), &o.res);
        }
        check_rc(h, rc);
    }
    raise_status(o.status);
    std::string seq;
    for (uint32_t i = 0; i < o.len; ++i) seq += alpha[o.labels[i]];
    if (qstring)
        for (uint32_t i = 0; i < o.len; ++i) append_utf8(seq, fcd_phred(o.qual[i], qscale, qbias));
    return py::make_tuple(py::str(seq), path_list(o));
}

int g_logadd_mode = FCD_LOGADD_LOGSUMEXP;

// ---- beam_search_duplex: lib.rs:401-488 ----
py::str beam_search_duplex(const py::object &network_output_1, const py::object &network_output_2,
                           const py::object &alphabet, const py::object &envelope,
                           const py::object &beam_size_o, float beam_cut_threshold,
                           bool collapse_repeats) {
    py::array x1 = as_f32(network_output_1, 2, "network_output_1");
    py::array x2 = as_f32(network_output_2, 2, "network_output_2");
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    if (x1.shape(1) != x2.shape(1)) throw py::value_error("inner axes of the network outputs do not match");
    check_beam_args(alpha.size(), x1.shape(1), (py::ssize_t)beam_size, beam_cut_threshold);
    const py::ssize_t T1 = x1.shape(0), T2 = x2.shape(0);
    std::vector<uint64_t> env_default;
    py::array env_arr;
    const uint64_t *env = nullptr;
    if (!envelope.is_none()) {  // lib.rs:445-456
        if (!py::isinstance<py::array>(envelope))
            throw py::type_error("argument 'envelope': expected numpy.ndarray");
        py::array e = py::reinterpret_borrow<py::array>(envelope);
        if (!e.dtype().is(py::dtype::of<uint64_t>()) || e.ndim() != 2)
            throw py::type_error("argument 'envelope': expected a 2-dimensional uint64 array");
        if (e.shape(0) != T1) throw py::value_error("the lengths of network_output_1 and envelope do not match");
        if (e.shape(1) != 2) throw py::value_error("the inner axis of envelope must have size 2");
        env_arr = py::array::ensure(e, py::array::c_style);
        env = static_cast<const uint64_t *>(env_arr.data());
    } else {  // lib.rs:459-468: every row searches the whole of read 2
        env_default.resize((size_t)(T1 > 0 ? T1 : 1) * 2);
        for (py::ssize_t t = 0; t < T1; ++t) {
            env_default[2 * t] = 0;
            env_default[2 * t + 1] = (uint64_t)T2;
        }
        env = env_default.data();
    }
    if (T1 == 0)
        throw std::runtime_error("network_output_1 is empty (the reference indexes envelope[(0,1)] and aborts)");
    Out o(T1, false, false);
    fcd_batch b1 = batch2(x1), b2 = batch2(x2);
    int rc;
    CoalescerUse use;
    if (fcd_coalescer *co = use.co) {  // concurrent per-pair calls share a launch (set_coalescing)
        {
            py::gil_scoped_release nogil;
            rc = fcd_coalescer_beam_search_duplex(co, &b1, &b2, env, (int64_t)beam_size, beam_cut_threshold,
                                                  collapse_repeats ? 1 : 0, g_logadd_mode, &o.res);
        }
        check_rc_coalescer(rc);
    } else {
        fcd_handle *h = thread_handle();
        {
            py::gil_scoped_release nogil;
            rc = fcd_beam_search_duplex_host(h, &b1, &b2, env, T1, (int64_t)beam_size, beam_cut_threshold,
                                             collapse_repeats ? 1 : 0, g_logadd_mode, &o.res);
        }
        check_rc(h, rc);
    }
    raise_status(o.status);
    std::string seq;
    for (uint32_t i = 0; i < o.len; ++i) seq += alpha[o.labels[i]];
    return py::str(seq);
}

// ---- crf_beam_search_duplex: lib.rs:490-578 ----
py::str crf_beam_search_duplex(const py::object &network_output_1, const py::object &init_state_1,
                               const py::object &network_output_2, const py::object &init_state_2,
                               const py::object &alphabet, const py::object &envelope,
                               const py::object &beam_size_o, float beam_cut_threshold) {
    py::array x1 = as_f32(network_output_1, 3, "network_output_1");
    py::array i1 = py::array::ensure(as_f32(init_state_1, 1, "init_state_1"), py::array::c_style);
    py::array x2 = as_f32(network_output_2, 3, "network_output_2");
    py::array i2 = py::array::ensure(as_f32(init_state_2, 1, "init_state_2"), py::array::c_style);
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    if (x1.shape(2) != x2.shape(2)) throw py::value_error("inner axes of the network outputs do not match");
    if ((py::ssize_t)alpha.size() != x1.shape(2))  // the reference's message quotes shape()[1] (lib.rs:512-516)
        throw py::value_error("alphabet size " + std::to_string(alpha.size()) +
                              " does not match probability matrix inner dimension " +
                              std::to_string(x1.shape(1)));
    check_beam_args(alpha.size(), x1.shape(2), (py::ssize_t)beam_size, beam_cut_threshold);
    const py::ssize_t T1 = x1.shape(0), T2 = x2.shape(0);
    std::vector<uint64_t> env_default;
    py::array env_arr;
    const uint64_t *env = nullptr;
    if (!envelope.is_none()) {
        if (!py::isinstance<py::array>(envelope))
            throw py::type_error("argument 'envelope': expected numpy.ndarray");
        py::array e = py::reinterpret_borrow<py::array>(envelope);
        if (!e.dtype().is(py::dtype::of<uint64_t>()) || e.ndim() != 2)
            throw py::type_error("argument 'envelope': expected a 2-dimensional uint64 array");
        if (e.shape(0) != T1) throw py::value_error("the lengths of network_output_1 and envelope do not match");
        if (e.shape(1) != 2) throw py::value_error("the inner axis of envelope must have size 2");
        env_arr = py::array::ensure(e, py::array::c_style);
        env = static_cast<const uint64_t *>(env_arr.data());
    } else {
        env_default.resize((size_t)(T1 > 0 ? T1 : 1) * 2);
        for (py::ssize_t t = 0; t < T1; ++t) {
            env_default[2 * t] = 0;
            env_default[2 * t + 1] = (uint64_t)T2;
        }
        env = env_default.data();
    }
    if (x1.shape(1) != x2.shape(1))
        throw std::runtime_error("state axes of the network outputs do not match (the reference asserts and aborts)");
    if (T1 == 0 || i1.size() == 0 || i2.size() == 0)
        throw std::runtime_error("empty network_output_1 / init_state (the reference aborts here)");
    Out o(T1, false, false);
    fcd_batch b1 = batch3(x1), b2 = batch3(x2);
    int rc;
    CoalescerUse use;
    if (fcd_coalescer *co = use.co) {
        {
            py::gil_scoped_release nogil;
            rc = fcd_coalescer_crf_beam_search_duplex(co, &b1, static_cast<const float *>(i1.data()), i1.shape(0), &b2,
                                                      static_cast<const float *>(i2.data()), i2.shape(0), env,
                                                      (int64_t)beam_size, beam_cut_threshold, g_logadd_mode, &o.res);
        }
        check_rc_coalescer(rc);
    } else {
        fcd_handle *h = thread_handle();
        {
            py::gil_scoped_release nogil;
            rc = fcd_crf_beam_search_duplex_host(h, &b1, static_cast<const float *>(i1.data()), i1.shape(0),
                                                 i1.shape(0), &b2, static_cast<const float *>(i2.data()),
                                                 i2.shape(0), i2.shape(0), env, T1, (int64_t)beam_size,
                                                 beam_cut_threshold, g_logadd_mode, &o.res);
        }
        check_rc(h, rc);
    }
    raise_status(o.status);
    std::string rev;  // duplex.rs:825-833: appended leaf -> root, characters reversed
    for (uint32_t i = o.len; i > 0; --i) rev += alpha[o.labels[i - 1]];
    return py::str(reverse_chars(rev));
}


// =============================================================================================================
// Batch functions (additive: the reference decodes one read per call).  Element i of the returned list is what
// the per-read function returns for read i.  A host batch runs as a job of result chunks (include/fcd.h,
// csrc/hostjob.hip): uploads, searches and packed downloads of different chunks overlap, and this thread turns
// chunk c into Python objects -- under the GIL -- while the later chunks are still in flight.
// =============================================================================================================
// batch inputs may be float32 or float16 (read by the kernels as they are, include/fcd.h FCD_DTYPE_*); the per-read
// functions keep the reference's float32-only rule
py::array as_post(const py::object &o, int ndim, const char *name, int &dtype, py::ssize_t &esz) {
    if (!py::isinstance<py::array>(o))
        throw py::type_error(std::string("argument '") + name + "': expected numpy.ndarray");
    py::array a = py::reinterpret_borrow<py::array>(o);
    const bool f32 = a.dtype().is(py::dtype::of<float>());
    const bool f16 = !f32 && a.dtype().kind() == 'f' && a.dtype().itemsize() == 2;
    if ((!f32 && !f16) || a.ndim() != ndim)
        throw py::type_error(std::string("argument '") + name + "': expected a " + std::to_string(ndim) +
                             "-dimensional float32 (or float16) array");
    dtype = f32 ? FCD_DTYPE_F32 : FCD_DTYPE_F16;
    esz = f32 ? 4 : 2;
    for (int d = 0; d < ndim; ++d)
        if (a.strides(d) < 0) return py::array::ensure(a, py::array::c_style);
    return a;
}

struct BatchInput {
    fcd_batch b{};
    py::array keep;                   // the caller's array (a view: zero-copy)
    std::unique_ptr<float[]> padded;  // or a padded copy of a sequence of per-read arrays
    std::vector<int64_t> lengths;
    py::ssize_t inner = 0;
    // ... or (the 1D searches' host jobs) the per-read arrays as they are: fcd_*_host_ptrs_begin gathers them chunk by
    // chunk on its lanes, no padded copy here
    std::vector<py::array> reads;
    py::object reads_owner;
    std::vector<const void *> ptrs;
    bool use_ptrs = false;
};

// network_outputs: ONE float32 array of rank `ndim` ((B,T,N) or (B,T,S,N), any non-negative strides: not copied),
// or a sequence of per-read float32 arrays of rank ndim-1 with equal inner shapes (copied into a padded batch;
// their row counts become the lengths).
void make_batch(BatchInput &in, const py::object &x, int ndim, const py::object &lengths_o, bool allow_ptrs = false) {
    if (py::isinstance<py::array>(x)) {
        int dtype = FCD_DTYPE_F32;
        py::ssize_t esz = 4;
        py::array a = as_post(x, ndim, "network_outputs", dtype, esz);
        in.keep = a;
        in.b.post = static_cast<const float *>(a.data());
        in.b.dtype = dtype;
        in.b.n_reads = a.shape(0);
        in.b.T = a.shape(1);
        in.b.stride_read = a.strides(0) / esz;
        in.b.stride_t = a.strides(1) / esz;
        if (ndim == 4) {
            in.b.S = a.shape(2);
            in.b.N = a.shape(3);
            in.b.stride_s = a.strides(2) / esz;
            in.b.stride_n = a.strides(3) / esz;
        } else {
            in.b.S = 1;
            in.b.N = a.shape(2);
            in.b.stride_n = a.strides(2) / esz;
        }
        in.inner = in.b.N;
    } else {
        py::sequence seq;
        try {
            seq = py::reinterpret_borrow<py::sequence>(x);
            if (!PySequence_Check(x.ptr())) throw py::type_error("");
        } catch (...) {
            throw py::type_error("argument 'network_outputs': expected a float32 numpy.ndarray or a sequence of them");
        }
        const py::ssize_t B = (py::ssize_t)py::len(seq);
        if (allow_ptrs && ndim == 3 && B > 0 && lengths_o.is_none()) {
            // the usual case -- a list of C-contiguous float32 (T_r, N) arrays -- costs a few fields per read, not a
            // dtype object and a PyArray_FromAny each (4096 reads: milliseconds, on a path that takes twelve)
            py::object fast = py::reinterpret_steal<py::object>(PySequence_Fast(x.ptr(), "network_outputs"));
            auto &npy = py::detail::npy_api::get();
            bool ok = (bool)fast;
            py::ssize_t Tm = 0, N0 = -1;
            in.ptrs.resize((size_t)B);
            in.lengths.resize((size_t)B);
            for (py::ssize_t i = 0; ok && i < B; ++i) {
                PyObject *it = PySequence_Fast_GET_ITEM(fast.ptr(), i);
                if (!npy.PyArray_Check_(it)) {
                    ok = false;
                    break;
                }
                auto *pa = py::detail::array_proxy(it);
                const int tn = py::detail::array_descriptor_proxy(pa->descr)->type_num;
                // native byte order ('=' or '|'; a '>f4' array has the same type number and other bytes) and aligned
                // data: anything else takes the converting path below.  The arrays are read by the lane threads with
                // the GIL released: they must not be written to until the batch call returns (README).
                const char bo = py::detail::array_descriptor_proxy(pa->descr)->byteorder;
                if (tn != py::detail::npy_api::NPY_FLOAT_ || pa->nd != 2 || (bo != '=' && bo != '|') ||
                    !(pa->flags & py::detail::npy_api::NPY_ARRAY_C_CONTIGUOUS_) ||
                    !(pa->flags & py::detail::npy_api::NPY_ARRAY_ALIGNED_) || (N0 >= 0 && pa->dimensions[1] != N0)) {
                    ok = false;
                    break;
                }
                N0 = pa->dimensions[1];
                Tm = std::max<py::ssize_t>(Tm, pa->dimensions[0]);
                in.ptrs[(size_t)i] = pa->data;
                in.lengths[(size_t)i] = pa->dimensions[0];
                in.reads.push_back(py::reinterpret_borrow<py::array>(it));  // (a reference of our own: the list may change)
            }
            if (ok) {
                in.reads_owner = fast;  // (holds the items: a tuple, or the caller's list)
                in.use_ptrs = true;
                in.b.n_reads = B;
                in.b.T = Tm;
                in.b.S = 1;
                in.b.N = N0;
                in.b.dtype = FCD_DTYPE_F32;
                in.inner = N0;
                return;
            }
            in.ptrs.clear();
            in.lengths.clear();
            in.reads.clear();
        }
        std::vector<py::array> reads;
        reads.reserve((size_t)B);
        py::ssize_t Tmax = 0, S = 1, N = 0;
        for (py::ssize_t i = 0; i < B; ++i) {
            py::array a = as_f32(seq[i], ndim - 1, "network_outputs[i]");
            const py::ssize_t s = ndim == 4 ? a.shape(1) : 1, n = a.shape(ndim - 2);
            if (i == 0) {
                S = s;
                N = n;
            } else if (s != S || n != N) {
                throw py::type_error("argument 'network_outputs': the reads' inner shapes differ");
            }
            Tmax = std::max(Tmax, a.shape(0));
            reads.push_back(a);
        }
        if (!lengths_o.is_none()) throw py::value_error("lengths cannot be given with a sequence of per-read arrays");
        if (allow_ptrs && ndim == 3 && B > 0) {
            in.lengths.resize((size_t)B);
            in.ptrs.resize((size_t)B);
            for (py::ssize_t i = 0; i < B; ++i) {
                in.reads.push_back(py::array::ensure(reads[(size_t)i], py::array::c_style));  // a no-op for contiguous reads
                in.lengths[(size_t)i] = in.reads.back().shape(0);
                in.ptrs[(size_t)i] = in.reads.back().data();
            }
            in.use_ptrs = true;
            in.b.n_reads = B;
            in.b.T = Tmax;
            in.b.S = 1;
            in.b.N = N;
            in.b.dtype = FCD_DTYPE_F32;
            in.inner = N;
            return;
        }
        const size_t row = (size_t)S * (size_t)N;
        in.padded.reset(new float[std::max<size_t>((size_t)B * (size_t)Tmax * row, 1)]);
        in.lengths.resize((size_t)B);
        for (py::ssize_t i = 0; i < B; ++i) {
            const py::array &a = reads[(size_t)i];
            in.lengths[(size_t)i] = a.shape(0);
            float *dst = in.padded.get() + (size_t)i * (size_t)Tmax * row;
            py::array c = py::array::ensure(a, py::array::c_style);  // a no-op for contiguous reads
            if (a.shape(0) > 0) std::memcpy(dst, c.data(), (size_t)a.shape(0) * row * sizeof(float));
        }
        in.b.post = in.padded.get();
        in.b.n_reads = B;
        in.b.T = Tmax;
        in.b.S = S;
        in.b.N = N;
        in.b.stride_read = (int64_t)((size_t)Tmax * row);
        in.b.stride_t = (int64_t)row;
        in.b.stride_s = ndim == 4 ? N : 0;
        in.b.stride_n = 1;
        in.b.lengths = in.lengths.data();
        in.inner = N;
        return;
    }
    if (!lengths_o.is_none()) {
        py::array_t<int64_t, py::array::c_style | py::array::forcecast> l(lengths_o);
        if (l.ndim() != 1 || l.shape(0) != in.b.n_reads) throw py::value_error("lengths must have shape (n_reads,)");
        in.lengths.assign(l.data(), l.data() + l.shape(0));
        for (int64_t v : in.lengths)
            if (v < 0 || v > in.b.T) throw py::value_error("lengths must lie in [0, T]");
        in.b.lengths = in.lengths.data();
    }
}

enum class Paths { List, Array, None };

Paths parse_paths(const py::object &o) {
    if (o.is_none()) return Paths::None;
    if (py::isinstance<py::str>(o)) {
        const std::string v = o.cast<std::string>();
        if (v == "list") return Paths::List;
        if (v == "array") return Paths::Array;
    }
    throw py::value_error("paths must be 'list', 'array' or None");
}

struct JobGuard {  // fcd_job_end on every way out
    fcd_job *job = nullptr;
    ~JobGuard() {
        if (job) {
            py::gil_scoped_release nogil;
            fcd_job_end(job);
        }
    }
};

// Python ints for path entries: every entry is a row index < T, so the lists share ONE int object per value
// (ints are immutable; creating eight million of them is what made list paths 10x slower than array paths).
struct IntTable {
    std::vector<PyObject *> v;
    ~IntTable() {
        for (PyObject *o : v) Py_XDECREF(o);
    }
    // every int below n exists afterwards: list paths are filled from v.data() without further checks
    void fill_below(size_t n) {
        if (n > v.size()) v.resize(n, nullptr);
        for (size_t i = 0; i < n; ++i)
            if (!v[i]) {
                v[i] = PyLong_FromSize_t(i);
                if (!v[i]) throw py::error_already_set();
            }
    }
    PyObject *get(size_t i) {
        if (i >= v.size()) v.resize(std::max(i + 1, v.size() * 2), nullptr);
        PyObject *&o = v[i];
        if (!o) {
            o = PyLong_FromSize_t(i);
            if (!o) throw py::error_already_set();
        }
        return o;
    }
};

struct BatchCall {
    enum Kind { Viterbi, Beam, CrfBeam, CrfGreedy } kind;
    bool qstring = false;
    float qscale = 1.0f, qbias = 0.0f;
    bool reverse_chars_join = false;  // the CRF beam search joins leaf -> root and reverses CHARACTERS
    int64_t T = 0;                    // rows per read: every path entry is a row index below it
};

// list[int] paths of one chunk.  The lists exist (created under the GIL, items NULL); their item arrays are plain memory,
// so a few threads fill them with pointers out of the shared int table -- every path entry is a row index below T, ints
// are immutable, so one object per index serves every list -- counting per thread how often each index was used; the
// reference counts are then settled in ONE pass over the table (T additions instead of one increment per entry: eight
// million at config 2).  Nothing in between can raise or touch a Python object.
struct ListFill {
    PyObject *list;
    uint64_t off;
    size_t len;
};

// The bulk path below writes PyListObject::ob_item from threads that do not hold the GIL and settles reference counts
// with Py_SET_REFCNT -- fine on a GIL build of CPython (tested: test_list_paths_reference_counts), wrong wherever a
// reference count is not a plain field one thread may add to: free-threaded CPython (Py_GIL_DISABLED: biased counts,
// per-object locks), the limited API, other interpreters.  There every entry is stored the documented way, under the
// GIL, with its own Py_INCREF.  -DFCD_PORTABLE_LISTS=1 forces that path (tests/test_host_layers.py builds it so).
#if !defined(FCD_PORTABLE_LISTS) && (defined(Py_GIL_DISABLED) || defined(Py_LIMITED_API) || defined(PYPY_VERSION) || defined(GRAALVM_PYTHON))
#define FCD_PORTABLE_LISTS 1
#endif

void fill_list_paths(const fcd_chunk &ch, const std::vector<ListFill> &fills, IntTable &ints, int64_t T) {
    if (fills.empty()) return;
    size_t n_idx = T > 0 ? (size_t)T : 0;
    if (n_idx == 0 || n_idx > (1u << 24)) {  // no bound known (or an absurd one): find the largest entry first
        n_idx = 0;
        for (const ListFill &f : fills)
            for (size_t k = 0; k < f.len; ++k) {
                const size_t v = ch.path_bytes == 2 ? static_cast<const uint16_t *>(ch.path)[f.off + k]
                                                    : static_cast<const uint32_t *>(ch.path)[f.off + k];
                n_idx = std::max(n_idx, v + 1);
            }
    }
    ints.fill_below(n_idx);
    PyObject *const *tab = ints.v.data();
#ifdef FCD_PORTABLE_LISTS
    for (const ListFill &f : fills)
        for (size_t k = 0; k < f.len; ++k) {
            size_t v = ch.path_bytes == 2 ? static_cast<const uint16_t *>(ch.path)[f.off + k]
                                          : static_cast<const uint32_t *>(ch.path)[f.off + k];
            if (v >= n_idx) v = 0;
            Py_INCREF(tab[v]);
            PyList_SET_ITEM(f.list, (Py_ssize_t)k, tab[v]);
        }
    return;
#else
    size_t total = 0;
    for (const ListFill &f : fills) total += f.len;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(4u, hw), std::max<size_t>(1, total / 200000));
    std::vector<std::vector<uint32_t>> counts(nt, std::vector<uint32_t>(n_idx, 0u));
    auto work = [&](unsigned t) {
        uint32_t *cnt = counts[t].data();
        for (size_t i = t; i < fills.size(); i += nt) {
            const ListFill &f = fills[i];
            PyObject **items = reinterpret_cast<PyListObject *>(f.list)->ob_item;
            if (ch.path_bytes == 2) {
                const uint16_t *src = static_cast<const uint16_t *>(ch.path) + f.off;
                for (size_t k = 0; k < f.len; ++k) {
                    const uint32_t v = src[k] < n_idx ? src[k] : 0u;
                    items[k] = tab[v];
                    ++cnt[v];
                }
            } else {
                const uint32_t *src = static_cast<const uint32_t *>(ch.path) + f.off;
                for (size_t k = 0; k < f.len; ++k) {
                    const uint32_t v = src[k] < n_idx ? src[k] : 0u;
                    items[k] = tab[v];
                    ++cnt[v];
                }
            }
        }
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (size_t i = 0; i < n_idx; ++i) {
        Py_ssize_t add = 0;
        for (unsigned t = 0; t < nt; ++t) add += (Py_ssize_t)counts[t][i];
        if (!add) continue;
#if PY_VERSION_HEX >= 0x030C0000
        if (_Py_IsImmortal(tab[i])) continue;  // (3.12+: the small ints are immortal, their count is not to be touched)
#endif
        Py_SET_REFCNT(tab[i], Py_REFCNT(tab[i]) + add);
    }
#endif
}

// Turns one result chunk into Python objects and stores them in result[read_begin ..].
void emit_chunk(const fcd_chunk &ch, const std::vector<std::string> &alpha, const BatchCall &call, Paths paths,
                bool raise_on_error, IntTable &ints, py::list &result) {
    // single-byte alphabets (the usual "NACGT"): labels -> characters through a table, straight into the str
    bool ascii1 = !call.qstring;
    uint8_t lut[256] = {0};
    for (size_t k = 0; k < alpha.size() && ascii1; ++k) {
        if (alpha[k].size() == 1 && (unsigned char)alpha[k][0] < 128 && k < 256) lut[k] = (uint8_t)alpha[k][0];
        else ascii1 = false;
    }
    // path entries of the whole chunk as ONE uint32 array; every read gets a view of it
    py::array_t<uint32_t> all_paths;
    uint32_t *ap = nullptr;
    const uint64_t total = ch.offsets[ch.n_reads];
    if (paths == Paths::Array && ch.path) {
        all_paths = py::array_t<uint32_t>((py::ssize_t)total);
        ap = all_paths.mutable_data();
        if (ch.path_bytes == 2) {
            const uint16_t *src = static_cast<const uint16_t *>(ch.path);
            for (uint64_t k = 0; k < total; ++k) ap[k] = src[k];
        } else {
            std::memcpy(ap, ch.path, (size_t)total * 4);
        }
    }
    std::string buf;
    std::vector<ListFill> fills;
    for (int64_t i = 0; i < ch.n_reads; ++i) {
        const int64_t r = ch.read_begin + i;
        const int32_t st = ch.status[i];
        if (st != FCD_ST_OK) {
            if (raise_on_error)
                throw std::runtime_error("read " + std::to_string(r) + ": " + fcd_status_string(st));
            Py_INCREF(Py_None);
            PyList_SET_ITEM(result.ptr(), r, Py_None);
            continue;
        }
        const uint64_t off = ch.offsets[i];
        const size_t len = (size_t)(ch.offsets[i + 1] - off);
        const uint8_t *lab = ch.labels + off;
        PyObject *seq;
        if (ascii1) {
            seq = PyUnicode_New((Py_ssize_t)len, 127);
            if (!seq) throw py::error_already_set();
            Py_UCS1 *d = PyUnicode_1BYTE_DATA(seq);
            for (size_t k = 0; k < len; ++k) d[k] = lut[lab[k]];
        } else {
            buf.clear();
            if (call.reverse_chars_join) {  // search.rs:146-156
                std::string rev;
                for (size_t k = len; k > 0; --k) rev += alpha[lab[k - 1]];
                buf = reverse_chars(rev);
            } else {
                for (size_t k = 0; k < len; ++k) buf += alpha[lab[k]];
            }
            if (call.qstring)
                for (size_t k = 0; k < len; ++k) append_utf8(buf, fcd_phred(ch.qual[off + k], call.qscale, call.qbias));
            seq = PyUnicode_DecodeUTF8(buf.data(), (Py_ssize_t)buf.size(), "strict");
            if (!seq) throw py::error_already_set();
        }
        PyObject *pth;
        if (paths == Paths::None || !ch.path) {
            Py_INCREF(Py_None);
            pth = Py_None;
        } else if (paths == Paths::Array) {
            py::array_t<uint32_t> view({(py::ssize_t)len}, {(py::ssize_t)4}, ap + off, all_paths);
            pth = view.release().ptr();
        } else {
            pth = PyList_New((Py_ssize_t)len);  // (items NULL: filled below, all reads of the chunk at once)
            if (!pth) {
                Py_DECREF(seq);
                throw py::error_already_set();
            }
            if (len) fills.push_back(ListFill{pth, off, len});
        }
        PyObject *tup = PyTuple_New(2);
        if (!tup) {
            Py_DECREF(seq);
            Py_DECREF(pth);
            throw py::error_already_set();
        }
        PyTuple_SET_ITEM(tup, 0, seq);
        PyTuple_SET_ITEM(tup, 1, pth);
        PyList_SET_ITEM(result.ptr(), r, tup);
    }
    fill_list_paths(ch, fills, ints, call.T);
}

// begin (already done by the caller) -> next / emit ... -> end
py::list run_job(fcd_handle *h, JobGuard &jg, int64_t B, const std::vector<std::string> &alpha, const BatchCall &call,
                 Paths paths, bool raise_on_error) {
    py::list result((py::ssize_t)B);  // slots are NULL until emit_chunk fills them
    IntTable ints;
    for (;;) {
        fcd_chunk ch;
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = fcd_job_next(jg.job, &ch);
        }
        if (rc == FCD_JOB_DONE) break;
        if (rc != FCD_OK) {
            // the list still has empty slots: fill them before it can be released
            for (int64_t r = 0; r < B; ++r)
                if (!PyList_GET_ITEM(result.ptr(), r)) {
                    Py_INCREF(Py_None);
                    PyList_SET_ITEM(result.ptr(), r, Py_None);
                }
            check_rc(h, rc);
        }
        try {
            emit_chunk(ch, alpha, call, paths, raise_on_error, ints, result);
        } catch (...) {
            for (int64_t r = 0; r < B; ++r)
                if (!PyList_GET_ITEM(result.ptr(), r)) {
                    Py_INCREF(Py_None);
                    PyList_SET_ITEM(result.ptr(), r, Py_None);
                }
            throw;
        }
    }
    return result;
}

int want_flags(Paths paths, bool qual) {
    return (paths != Paths::None ? FCD_JOB_PATH : 0) | (qual ? FCD_JOB_QUAL : 0);
}

py::list beam_search_batch(const py::object &network_outputs, const py::object &alphabet,
                           const py::object &beam_size_o, float beam_cut_threshold, bool collapse_repeats,
                           const py::object &lengths, const py::object &paths_o, int kernel, bool raise_on_error) {
    BatchInput in;
    make_batch(in, network_outputs, 3, lengths, true);
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    check_beam_args(alpha.size(), in.inner, (py::ssize_t)beam_size, beam_cut_threshold);
    const Paths paths = parse_paths(paths_o);
    fcd_handle *h = thread_handle();
    JobGuard jg;
    int rc;
    {
        py::gil_scoped_release nogil;
        if (in.use_ptrs)
            rc = fcd_beam_search_host_ptrs_begin(h, in.ptrs.data(), in.lengths.data(), in.b.n_reads, in.b.N, in.b.dtype,
                                                 (int64_t)beam_size, beam_cut_threshold, collapse_repeats ? 1 : 0, kernel,
                                                 want_flags(paths, false), &jg.job);
        else
            rc = fcd_beam_search_host_begin(h, &in.b, (int64_t)beam_size, beam_cut_threshold, collapse_repeats ? 1 : 0,
                                            kernel, want_flags(paths, false), &jg.job);
    }
    check_rc(h, rc);
    BatchCall call{BatchCall::Beam};
    call.T = in.b.T;
    return run_job(h, jg, in.b.n_reads, alpha, call, paths, raise_on_error);
}

py::list viterbi_search_batch(const py::object &network_outputs, const py::object &alphabet, bool qstring, float qscale,
                              float qbias, bool collapse_repeats, const py::object &lengths, const py::object &paths_o,
                              bool raise_on_error) {
    BatchInput in;
    make_batch(in, network_outputs, 3, lengths, true);
    auto alpha = seq_to_vec(alphabet);
    check_greedy_alphabet(alpha.size(), in.inner);
    if (in.b.T == 0 && in.b.n_reads > 0)
        throw std::runtime_error("network_output is empty (the reference asserts and aborts here)");
    const Paths paths = parse_paths(paths_o);
    fcd_handle *h = thread_handle();
    JobGuard jg;
    int rc;
    {
        py::gil_scoped_release nogil;
        if (in.use_ptrs)
            rc = fcd_viterbi_search_host_ptrs_begin(h, in.ptrs.data(), in.lengths.data(), in.b.n_reads, in.b.N, in.b.dtype,
                                                    collapse_repeats ? 1 : 0, want_flags(paths, qstring), &jg.job);
        else
            rc = fcd_viterbi_search_host_begin(h, &in.b, collapse_repeats ? 1 : 0, want_flags(paths, qstring), &jg.job);
    }
    check_rc(h, rc);
    BatchCall call{BatchCall::Viterbi};
    call.T = in.b.T;
    call.qstring = qstring;
    call.qscale = qscale;
    call.qbias = qbias;
    return run_job(h, jg, in.b.n_reads, alpha, call, paths, raise_on_error);
}

py::array init_states_array(const py::object &init_states, int64_t B) {
    py::array init = py::array::ensure(as_f32(init_states, 2, "init_states"), py::array::c_style);
    if (init.shape(0) != B) throw py::value_error("init_states must have shape (n_reads, n_init)");
    if (init.shape(1) == 0 && B > 0)
        throw std::runtime_error("network_output/init_state is empty (the reference asserts and aborts here)");
    return init;
}

py::list crf_beam_search_batch(const py::object &network_outputs, const py::object &init_states,
                               const py::object &alphabet, const py::object &beam_size_o, float beam_cut_threshold,
                               const py::object &lengths, const py::object &paths_o, int kernel, bool raise_on_error) {
    BatchInput in;
    make_batch(in, network_outputs, 4, lengths);
    py::array init = init_states_array(init_states, in.b.n_reads);
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    check_greedy_alphabet(alpha.size(), in.inner);
    if (in.b.n_reads > 0 && (in.b.T == 0 || in.b.S == 0))
        throw std::runtime_error("network_output/init_state is empty (the reference asserts and aborts here)");
    if (beam_size == 0) raise_status(FCD_ST_RAN_OUT_OF_BEAM);
    const Paths paths = parse_paths(paths_o);
    fcd_handle *h = thread_handle();
    JobGuard jg;
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = fcd_crf_beam_search_host_begin(h, &in.b, static_cast<const float *>(init.data()), init.shape(1),
                                            init.shape(1), (int64_t)beam_size, beam_cut_threshold, kernel,
                                            want_flags(paths, false), &jg.job);
    }
    check_rc(h, rc);
    BatchCall call{BatchCall::CrfBeam};
    call.T = in.b.T;
    call.reverse_chars_join = true;
    return run_job(h, jg, in.b.n_reads, alpha, call, paths, raise_on_error);
}

py::list crf_greedy_search_batch(const py::object &network_outputs, const py::object &init_states,
                                 const py::object &alphabet, bool qstring, float qscale, float qbias,
                                 const py::object &lengths, const py::object &paths_o, bool raise_on_error) {
    BatchInput in;
    make_batch(in, network_outputs, 4, lengths);
    py::array init = init_states_array(init_states, in.b.n_reads);
    auto alpha = seq_to_vec(alphabet);
    check_greedy_alphabet(alpha.size(), in.inner);
    if (in.b.n_reads > 0 && (in.b.T == 0 || in.b.S == 0))
        throw std::runtime_error("network_output/init_state is empty (the reference asserts and aborts here)");
    const Paths paths = parse_paths(paths_o);
    fcd_handle *h = thread_handle();
    JobGuard jg;
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = fcd_crf_greedy_search_host_begin(h, &in.b, static_cast<const float *>(init.data()), init.shape(1),
                                              init.shape(1), want_flags(paths, qstring), &jg.job);
    }
    check_rc(h, rc);
    BatchCall call{BatchCall::CrfGreedy};
    call.T = in.b.T;
    call.qstring = qstring;
    call.qscale = qscale;
    call.qbias = qbias;
    return run_job(h, jg, in.b.n_reads, alpha, call, paths, raise_on_error);
}

// ---- duplex batches: n pairs per call (one launch), sequences only (the duplex searches report no path) ----
// envelopes: None (every row of read 1 searches the whole of read 2, lib.rs:459-468) or ONE uint64 array (B, T1, 2).
struct DuplexOut {
    std::vector<uint8_t> labels;
    std::vector<uint32_t> len;
    std::vector<int32_t> status;
    fcd_result res{};
    DuplexOut(int64_t B, int64_t T1) {
        const size_t w = T1 > 0 ? (size_t)T1 : 1, n = B > 0 ? (size_t)B : 1;
        labels.resize(n * w);
        len.assign(n, 0);
        status.assign(n, 0);
        res.labels = labels.data();
        res.out_len = len.data();
        res.status = status.data();
        res.out_stride = (int64_t)w;
    }
};

const uint64_t *duplex_envelopes(const py::object &envelopes, int64_t B, int64_t T1, int64_t T2, const std::vector<int64_t> &len2,
                                 std::vector<uint64_t> &fallback, py::array &keep) {
    if (!envelopes.is_none()) {
        if (!py::isinstance<py::array>(envelopes)) throw py::type_error("argument 'envelopes': expected numpy.ndarray");
        py::array e = py::reinterpret_borrow<py::array>(envelopes);
        if (!e.dtype().is(py::dtype::of<uint64_t>()) || e.ndim() != 3)
            throw py::type_error("argument 'envelopes': expected a 3-dimensional uint64 array");
        if (e.shape(0) != B || e.shape(1) != T1 || e.shape(2) != 2)
            throw py::value_error("envelopes must have shape (n_pairs, T1, 2)");
        keep = py::array::ensure(e, py::array::c_style);
        return static_cast<const uint64_t *>(keep.data());
    }
    fallback.resize((size_t)std::max<int64_t>(B * T1, 1) * 2);
    for (int64_t r = 0; r < B; ++r) {
        const uint64_t hi = (uint64_t)(len2.empty() ? T2 : std::min<int64_t>(std::max<int64_t>(len2[(size_t)r], 0), T2));
        for (int64_t t = 0; t < T1; ++t) {
            fallback[(size_t)(r * T1 + t) * 2] = 0;
            fallback[(size_t)(r * T1 + t) * 2 + 1] = hi;
        }
    }
    return fallback.data();
}

py::list duplex_sequences(const DuplexOut &o, int64_t B, const std::vector<std::string> &alpha, bool reversed,
                          bool raise_on_error) {
    py::list out((py::ssize_t)B);
    for (int64_t r = 0; r < B; ++r) {
        PyObject *item;
        if (o.status[(size_t)r] != FCD_ST_OK) {
            if (raise_on_error) {
                for (int64_t q = r; q < B; ++q) {  // (the list must not be released with empty slots)
                    Py_INCREF(Py_None);
                    PyList_SET_ITEM(out.ptr(), q, Py_None);
                }
                throw std::runtime_error("pair " + std::to_string(r) + ": " + fcd_status_string(o.status[(size_t)r]));
            }
            Py_INCREF(Py_None);
            item = Py_None;
        } else {
            const uint8_t *lab = o.labels.data() + (size_t)r * (size_t)o.res.out_stride;
            const uint32_t n = o.len[(size_t)r];
            std::string seq;
            if (reversed) {  // duplex.rs:825-833: appended leaf -> root, then the CHARACTERS are reversed
                for (uint32_t i = n; i > 0; --i) seq += alpha[lab[i - 1]];
                seq = reverse_chars(seq);
            } else {
                for (uint32_t i = 0; i < n; ++i) seq += alpha[lab[i]];
            }
            item = PyUnicode_FromStringAndSize(seq.data(), (Py_ssize_t)seq.size());
            if (!item) throw py::error_already_set();
        }
        PyList_SET_ITEM(out.ptr(), r, item);
    }
    return out;
}

int duplex_mode_of(const py::object &m) {  // None: the module's setting; "logsumexp" / "max"; or the C ABI's integer
    if (m.is_none()) return g_logadd_mode;
    if (py::isinstance<py::str>(m)) {
        const std::string v = m.cast<std::string>();
        if (v == "logsumexp") return FCD_LOGADD_LOGSUMEXP;
        if (v == "max") return FCD_LOGADD_MAX;
        throw py::value_error("logadd_mode must be 'logsumexp' or 'max'");
    }
    const int v = m.cast<int>();
    if (v != FCD_LOGADD_LOGSUMEXP && v != FCD_LOGADD_MAX) throw py::value_error("logadd_mode must be 'logsumexp' or 'max'");
    return v;
}

py::list beam_search_duplex_batch(const py::object &network_outputs_1, const py::object &network_outputs_2,
                                  const py::object &alphabet, const py::object &envelopes, const py::object &beam_size_o,
                                  float beam_cut_threshold, bool collapse_repeats, const py::object &lengths_1,
                                  const py::object &lengths_2, bool raise_on_error, const py::object &logadd_mode) {
    const int mode = duplex_mode_of(logadd_mode);
    BatchInput in1, in2;
    make_batch(in1, network_outputs_1, 3, lengths_1);
    make_batch(in2, network_outputs_2, 3, lengths_2);
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    if (in1.inner != in2.inner) throw py::value_error("inner axes of the network outputs do not match");
    check_beam_args(alpha.size(), in1.inner, (py::ssize_t)beam_size, beam_cut_threshold);
    const int64_t B = in1.b.n_reads, T1 = in1.b.T, T2 = in2.b.T;
    if (in2.b.n_reads != B) throw py::value_error("both batches must hold the same number of reads");
    if (B > 0 && T1 == 0)
        throw std::runtime_error("network_output_1 is empty (the reference indexes envelope[(0,1)] and aborts)");
    std::vector<uint64_t> env_default;
    py::array env_keep;
    const uint64_t *env = duplex_envelopes(envelopes, B, T1, T2, in2.lengths, env_default, env_keep);
    DuplexOut o(B, T1);
    if (B == 0) return py::list();
    int rc;
    fcd_handle *h = thread_handle();
    {
        py::gil_scoped_release nogil;
        rc = fcd_beam_search_duplex_host(h, &in1.b, &in2.b, env, T1, (int64_t)beam_size, beam_cut_threshold,
                                         collapse_repeats ? 1 : 0, mode, &o.res);
    }
    check_rc(h, rc);
    return duplex_sequences(o, B, alpha, false, raise_on_error);
}

py::list crf_beam_search_duplex_batch(const py::object &network_outputs_1, const py::object &init_states_1,
                                      const py::object &network_outputs_2, const py::object &init_states_2,
                                      const py::object &alphabet, const py::object &envelopes,
                                      const py::object &beam_size_o, float beam_cut_threshold,
                                      const py::object &lengths_1, const py::object &lengths_2, bool raise_on_error,
                                      const py::object &logadd_mode) {
    const int mode = duplex_mode_of(logadd_mode);
    BatchInput in1, in2;
    make_batch(in1, network_outputs_1, 4, lengths_1);
    make_batch(in2, network_outputs_2, 4, lengths_2);
    const int64_t B = in1.b.n_reads, T1 = in1.b.T, T2 = in2.b.T;
    if (in2.b.n_reads != B) throw py::value_error("both batches must hold the same number of reads");
    py::array i1 = init_states_array(init_states_1, B), i2 = init_states_array(init_states_2, B);
    auto alpha = seq_to_vec(alphabet);
    const size_t beam_size = to_usize(beam_size_o, "beam_size");
    if (in1.inner != in2.inner) throw py::value_error("inner axes of the network outputs do not match");
    check_beam_args(alpha.size(), in1.inner, (py::ssize_t)beam_size, beam_cut_threshold);
    if (in1.b.S != in2.b.S)
        throw std::runtime_error("state axes of the network outputs do not match (the reference asserts and aborts)");
    if (B > 0 && T1 == 0) throw std::runtime_error("empty network_output_1 / init_state (the reference aborts here)");
    std::vector<uint64_t> env_default;
    py::array env_keep;
    const uint64_t *env = duplex_envelopes(envelopes, B, T1, T2, in2.lengths, env_default, env_keep);
    DuplexOut o(B, T1);
    if (B == 0) return py::list();
    int rc;
    fcd_handle *h = thread_handle();
    {
        py::gil_scoped_release nogil;
        rc = fcd_crf_beam_search_duplex_host(h, &in1.b, static_cast<const float *>(i1.data()), i1.shape(1), i1.shape(1),
                                             &in2.b, static_cast<const float *>(i2.data()), i2.shape(1), i2.shape(1), env,
                                             T1, (int64_t)beam_size, beam_cut_threshold, mode, &o.res);
    }
    check_rc(h, rc);
    return duplex_sequences(o, B, alpha, true, raise_on_error);
}

}  // namespace

PYBIND11_MODULE(fast_ctc_decode, m) {
    m.doc() = "Methods for labelling RNN results using CTC decoding (MI355X / HIP build of fast_ctc_decode).";
    using namespace pybind11::literals;
    m.def("viterbi_search", &viterbi_search, "network_output"_a, "alphabet"_a, "qstring"_a = false,
          "qscale"_a = 1.0f, "qbias"_a = 0.0f, "collapse_repeats"_a = true,
          "viterbi_search(network_output, alphabet, qstring=False, qscale=1.0, qbias=0.0, collapse_repeats=True)");
    m.def("beam_search", &beam_search, "network_output"_a, "alphabet"_a, "beam_size"_a = 5,
          "beam_cut_threshold"_a = 0.0f, "collapse_repeats"_a = true,
          "beam_search(network_output, alphabet, beam_size=5, beam_cut_threshold=0.0, collapse_repeats=True)");
    m.def("crf_beam_search", &crf_beam_search, "network_output"_a, "init_state"_a, "alphabet"_a,
          "beam_size"_a = 5, "beam_cut_threshold"_a = 0.0f,
          "crf_beam_search(network_output, init_state, alphabet, beam_size, beam_cut_threshold)");
    m.def("crf_greedy_search", &crf_greedy_search, "network_output"_a, "init_state"_a, "alphabet"_a,
          "qstring"_a = false, "qscale"_a = 1.0f, "qbias"_a = 0.0f,
          "crf_greedy_search(network_output, init_state, alphabet)");
    m.def("beam_search_duplex", &beam_search_duplex, "network_output_1"_a, "network_output_2"_a,
          "alphabet"_a, "envelope"_a = py::none(), "beam_size"_a = 5, "beam_cut_threshold"_a = 0.0f,
          "collapse_repeats"_a = true,
          "beam_search_duplex(network_output_1, network_output_2, alphabet, envelope=None, beam_size=5, "
          "beam_cut_threshold=0.0, collapse_repeats=True)");
    m.def("crf_beam_search_duplex", &crf_beam_search_duplex, "network_output_1"_a, "init_state_1"_a,
          "network_output_2"_a, "init_state_2"_a, "alphabet"_a, "envelope"_a = py::none(),
          "beam_size"_a = 5, "beam_cut_threshold"_a = 0.0f,
          "crf_beam_search_duplex(network_output_1, init_state_1, network_output_2, init_state_2, alphabet, "
          "envelope=None, beam_size=5, beam_cut_threshold=0.0)");
    // not part of the reference surface: selects what the reference fixes at build time
    // (`fastexp` feature on = "max", off = "logsumexp"; SURVEY.md finding 3).  The default is "logsumexp"
    // (the north star's "fastexp disabled"); the published PyPI wheels are built WITH fastexp and compute
    // "max" -- callers comparing against them must switch (also: environment variable
    // FCD_DUPLEX_LOGADD=max, read once at import).
    auto set_mode = [](const std::string &mode) {
        if (mode == "logsumexp") g_logadd_mode = FCD_LOGADD_LOGSUMEXP;
        else if (mode == "max") g_logadd_mode = FCD_LOGADD_MAX;
        else if (mode == "logsumexp_glibc235") g_logadd_mode = FCD_LOGADD_LOGSUMEXP_GLIBC235;
        else throw py::value_error("mode must be 'logsumexp', 'max' or 'logsumexp_glibc235'");
    };
    m.def("set_duplex_logadd_mode", set_mode, "mode"_a,
          "set_duplex_logadd_mode(mode): 'logsumexp' (default; the reference built with --no-default-features) or "
          "'max' (the reference's default `fastexp` feature, i.e. what the PyPI wheels compute) or 'logsumexp_glibc235' "
          "(logsumexp on glibc 2.35's expf / logf / log1pf, bit for bit)");
    m.def("_set_duplex_logadd_mode", set_mode);  // earlier name, kept for the tests
    m.def(
        "set_tie_order",
        [](const std::string &order) {
            int o;
            if (order == "pdq178") o = FCD_TIE_PDQ178;
            else if (order == "stable") o = FCD_TIE_STABLE;
            else throw py::value_error("order must be 'pdq178' or 'stable'");
            fcd_set_default_tie_order(o);
        },
        "order"_a,
        "set_tie_order(order): how the beam searches order EQUAL probabilities among more than 20 candidates -- "
        "'pdq178' (default: the order Rust 1.78's sort_unstable_by, i.e. the reference wheels, leaves them in) or "
        "'stable' (ascending node index).  Process-wide (include/fcd.h, FCD_TIE_*).");
#ifdef FCD_PORTABLE_LISTS
    m.attr("_portable_lists") = true;
#else
    m.attr("_portable_lists") = false;
#endif
    m.def("tie_order", []() { return std::string(fcd_get_tie_order(nullptr) == FCD_TIE_STABLE ? "stable" : "pdq178"); });
    m.def(
        "set_coalescing",
        [](int max_batch, int max_wait_us, int device) {
            fcd_coalescer *fresh = nullptr;
            if (max_batch > 0) {
                const int rc = fcd_coalescer_create(device, max_batch, max_wait_us, &fresh);
                if (rc != FCD_OK || !fresh)
                    throw std::runtime_error("fast_ctc_decode: no usable gfx950 device (fcd_coalescer_create failed with " +
                                             std::to_string(rc) + "); this module has no CPU fallback");
            }
            std::shared_ptr<CoalescerBox> box;
            if (fresh) {
                box = std::make_shared<CoalescerBox>();
                box->co = fresh;
            }
            std::shared_ptr<CoalescerBox> old;
            {
                std::lock_guard<std::mutex> g(g_coalescer_mu);
                old = std::move(g_coalescer);
                g_coalescer = std::move(box);
            }
            py::gil_scoped_release nogil;  // dropping the last reference destroys the old coalescer
            old.reset();
        },
        "max_batch"_a = 256, "max_wait_us"_a = 0, "device"_a = 0,
        "Decode concurrent per-read viterbi_search / beam_search calls (any threads) with shared batched launches; "
        "max_batch=0 switches it off.  Results do not change.  (Not in the reference.)");
    m.def("coalescing_stats", []() -> py::object {
        CoalescerUse use;
        fcd_coalescer *co = use.co;
        if (!co) return py::none();
        int64_t a = 0, b = 0, c = 0;
        fcd_coalescer_stats(co, &a, &b, &c);
        py::dict d;
        d["calls"] = a;
        d["launches"] = b;
        d["largest_batch"] = c;
        return std::move(d);
    });
    // ---- additive batch functions (not in the reference): element i == the per-read call on read i ----
    m.def("beam_search_batch", &beam_search_batch, "network_outputs"_a, "alphabet"_a, "beam_size"_a = 5,
          "beam_cut_threshold"_a = 0.0f, "collapse_repeats"_a = true, "lengths"_a = py::none(), "paths"_a = "list",
          "kernel"_a = 0, "raise_on_error"_a = true,
          "beam_search_batch(network_outputs, alphabet, beam_size=5, beam_cut_threshold=0.0, collapse_repeats=True, "
          "lengths=None, paths='list') -> list of (str, path).  network_outputs: one (B,T,N) float32 array (zero-copy) "
          "or a sequence of (T_i,N) arrays.  paths: 'list' (list[int], as the reference), 'array' (uint32 ndarray "
          "views) or None.");
    m.def("viterbi_search_batch", &viterbi_search_batch, "network_outputs"_a, "alphabet"_a, "qstring"_a = false,
          "qscale"_a = 1.0f, "qbias"_a = 0.0f, "collapse_repeats"_a = true, "lengths"_a = py::none(),
          "paths"_a = "list", "raise_on_error"_a = true);
    m.def("crf_beam_search_batch", &crf_beam_search_batch, "network_outputs"_a, "init_states"_a, "alphabet"_a,
          "beam_size"_a = 5, "beam_cut_threshold"_a = 0.0f, "lengths"_a = py::none(), "paths"_a = "list",
          "kernel"_a = 0, "raise_on_error"_a = true);
    m.def("crf_greedy_search_batch", &crf_greedy_search_batch, "network_outputs"_a, "init_states"_a, "alphabet"_a,
          "qstring"_a = false, "qscale"_a = 1.0f, "qbias"_a = 0.0f, "lengths"_a = py::none(), "paths"_a = "list",
          "raise_on_error"_a = true);
    m.def("beam_search_duplex_batch", &beam_search_duplex_batch, "network_outputs_1"_a, "network_outputs_2"_a, "alphabet"_a,
          "envelopes"_a = py::none(), "beam_size"_a = 5, "beam_cut_threshold"_a = 0.0f, "collapse_repeats"_a = true,
          "lengths_1"_a = py::none(), "lengths_2"_a = py::none(), "raise_on_error"_a = true, "logadd_mode"_a = py::none(),
          "beam_search_duplex for n pairs in one launch -> list[str] (None for a failed pair when raise_on_error is "
          "false).  envelopes: None or ONE uint64 array (n_pairs, T1, 2).");
    m.def("crf_beam_search_duplex_batch", &crf_beam_search_duplex_batch, "network_outputs_1"_a, "init_states_1"_a,
          "network_outputs_2"_a, "init_states_2"_a, "alphabet"_a, "envelopes"_a = py::none(), "beam_size"_a = 5,
          "beam_cut_threshold"_a = 0.0f, "lengths_1"_a = py::none(), "lengths_2"_a = py::none(),
          "raise_on_error"_a = true, "logadd_mode"_a = py::none(),
          "crf_beam_search_duplex for n pairs in one launch -> list[str].");
    m.def("_set_host_pipeline", [](int lanes, int64_t chunk_reads, int64_t min_bytes) {
        fcd_handle *h = thread_handle();
        check_rc(h, fcd_set_host_pipeline(h, lanes, chunk_reads, min_bytes));
    }, "lanes"_a = 0, "chunk_reads"_a = 0, "min_bytes"_a = -1, "tuning / tests: include/fcd.h fcd_set_host_pipeline");
    if (const char *env = std::getenv("FCD_DUPLEX_LOGADD")) set_mode(env);
    m.attr("__version__") = "0.3.7";  // src/lib.rs:626 (CARGO_PKG_VERSION of the mirrored reference)
}
