// capi.hip -- the C ABI of libfcd_hip.so (include/fcd.h): argument checking, workspace and
// chunking, stream/event plumbing, host staging for the *_host entry points.
#include <stdio.h>
#include <math.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "fcd_internal.h"
#include "glibc235_math.h"
#include "slab_pool.h"

using namespace fcd;

namespace {

// process default of the prune's tie order (include/fcd.h, FCD_TIE_*): the environment at load time, then
// fcd_set_default_tie_order
int tie_order_from_env() {
    const char *e = getenv("FCD_TIE_ORDER");
    if (!e || !*e) return FCD_TIE_PDQ178;
    if (!strcmp(e, "stable") || !strcmp(e, "STABLE") || !strcmp(e, "1")) return FCD_TIE_STABLE;
    if (!strcmp(e, "pdq178") || !strcmp(e, "PDQ178") || !strcmp(e, "2")) return FCD_TIE_PDQ178;
    // a typo must not silently select the other order
    fprintf(stderr, "fast_ctc_decode (fcd): FCD_TIE_ORDER=\"%s\" is neither \"pdq178\" nor \"stable\"; using pdq178\n", e);
    return FCD_TIE_PDQ178;
}
std::atomic<int> g_tie_order{tie_order_from_env()};

// FCD_PDQ178_STD_FORM (include/fcd.h): which form of the two routines std changed in 2023 the quicksort replay follows --
// a process-wide word, copied to every device a handle is created on (csrc/pdq178.h g_std_form)
int pdq178_std_form_from_env() {
    const char *e = getenv("FCD_PDQ178_STD_FORM");
    if (!e || !*e) return 0;
    if (e[0] >= '0' && e[0] <= '3' && !e[1]) {
        // (said once, so that nobody compares a 1-D search under form 3 with a duplex search and wonders)
        if (e[0] != '0')
            fprintf(stderr, "fast_ctc_decode (fcd): FCD_PDQ178_STD_FORM=%s applies to the 1-D searches; the duplex searches replay "
                            "form 0 only (csrc/pdq178.h)\n", e);
        return e[0] - '0';
    }
    fprintf(stderr, "fast_ctc_decode (fcd): FCD_PDQ178_STD_FORM=\"%s\" is not 0, 1, 2 or 3; using 0\n", e);
    return 0;
}
std::atomic<int> g_pdq178_std_form{pdq178_std_form_from_env()};

// (on the current device)
hipError_t apply_pdq178_std_form(int bits) {
    hipError_t e = beam_wave_set_pdq178_std_form(bits);
    if (e == hipSuccess) e = beam_lane_set_pdq178_std_form(bits);
    if (e == hipSuccess) e = beam_generic_set_pdq178_std_form(bits);
    if (e == hipSuccess) e = tieorder_set_pdq178_std_form(bits);
    return e;
}

constexpr int kRetrySlabs = 256;  // lane kernel, two-pass sizing: worst-case slabs of the retry pass (its wavefronts wait for one)

#define FCD_HIP(h, expr)                                                          \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);        \
            return FCD_E_HIP;                                                     \
        }                                                                         \
    } while (0)

// Binds the calling thread to the handle's device for the duration of an entry point and puts the previous
// device back afterwards: a caller that works on several GPUs (torch) must not find its current device changed.
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        err = prev == dev ? hipSuccess : hipSetDevice(dev);
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define FCD_DEVICE(h)                \
    DeviceGuard dev_guard__((h)->device); \
    FCD_HIP(h, dev_guard__.err)

int fail(fcd_handle *h, int code, const char *msg) {
    if (h) h->err = msg;
    return code;
}

int ensure(fcd_handle *h, void **buf, size_t *have, size_t need) {
    if (*have >= need) return FCD_OK;
    if (*buf) {
        // work enqueued earlier -- on the current stream or on one set before the last fcd_set_stream -- may
        // still use the buffer: hipFree waits for the whole device, but be explicit about both streams
        FCD_HIP(h, hipStreamSynchronize(h->stream));
        if (h->own_stream && h->own_stream != h->stream) FCD_HIP(h, hipStreamSynchronize(h->own_stream));
        FCD_HIP(h, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    hipError_t e = hipMalloc(buf, need);
    if (e != hipSuccess) {
        *buf = nullptr;
        h->err = std::string("hipMalloc of workspace failed: ") + hipGetErrorString(e);
        return FCD_E_NOMEM;
    }
    *have = need;
    return FCD_OK;
}

int check_batch(fcd_handle *h, const fcd_batch *in, bool crf) {
    if (!h) return FCD_E_INVALID;
    if (!in) return fail(h, FCD_E_INVALID, "null batch");
    if (in->n_reads < 0 || in->T < 0 || in->N < 1) return fail(h, FCD_E_INVALID, "negative size");
    if (in->n_reads > 0 && in->T > 0 && !in->post) return fail(h, FCD_E_INVALID, "null post");
    if (in->N > 256) return fail(h, FCD_E_UNSUPPORTED, "alphabets above 256 labels are unsupported (u8 labels)");
    if (in->T >= (1ll << 28)) return fail(h, FCD_E_UNSUPPORTED, "T must be < 2^28");
    if (in->dtype != FCD_DTYPE_F32 && in->dtype != FCD_DTYPE_F16 && in->dtype != FCD_DTYPE_BF16)
        return fail(h, FCD_E_INVALID, "unknown posterior dtype");
    if (crf && in->S < 1) return fail(h, FCD_E_INVALID, "S must be >= 1");
    return FCD_OK;
}

int check_result(fcd_handle *h, const fcd_batch *in, const fcd_result *out, bool need_status) {
    if (!out) return fail(h, FCD_E_INVALID, "null result");
    if (in->n_reads == 0) return FCD_OK;
    if (!out->labels || !out->out_len) return fail(h, FCD_E_INVALID, "null labels/out_len");
    if (need_status && !out->status) return fail(h, FCD_E_INVALID, "null status");
    if (out->out_stride < in->T) return fail(h, FCD_E_INVALID, "out_stride must be >= T");
    return FCD_OK;
}

BatchDesc to_desc(const fcd_batch *in, bool crf) {
    BatchDesc d;
    d.post = static_cast<const float *>(in->post);  // (typed by `dtype`; the kernels convert on load)
    d.lengths = in->lengths;
    d.n_reads = in->n_reads;
    d.T = in->T;
    d.stride_read = in->stride_read;
    d.stride_t = in->stride_t;
    d.stride_s = crf ? in->stride_s : 0;
    d.stride_n = in->stride_n;
    d.S = crf ? (int)in->S : 1;
    d.N = (int)in->N;
    d.dtype = in->dtype;
    return d;
}

ResultDesc to_desc(const fcd_result *o) {
    return ResultDesc{o->labels, o->path, o->qual, o->out_len, o->status, o->out_stride, o->ambiguous};
}

// Brackets the kernel launches of one search call with HIP events on the launch stream.
struct Timer {
    fcd_handle *h;
    int slot;
    hipStream_t st;
    explicit Timer(fcd_handle *hh, hipStream_t on = nullptr, bool given = false) : h(hh), st(given ? on : hh->stream) {
        slot = (int)(h->n_timed % fcd_handle::kTimingRing);
        if ((int)h->ev0.size() <= slot) {
            hipEvent_t a = nullptr, b = nullptr;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            h->ev0.push_back(a);
            h->ev1.push_back(b);
        }
        (void)hipEventRecord(h->ev0[slot], st);
    }
    void stop() {
        (void)hipEventRecord(h->ev1[slot], st);
        h->n_timed++;
    }
};

int64_t workspace_budget(fcd_handle *h) {
    if (h->ws_limit > 0) return h->ws_limit;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 8ll << 30;
    return (int64_t)((free_b + h->arena_bytes + h->pool_arena_bytes) / 2);
}

// ---- fcd_set_overlap: wide-beam calls on internal streams ----
// every internal stream's work is finished (host wait)
int overlap_drain(fcd_handle *h) {
    for (int s = 0; s < fcd_handle::kMaxOverlap; ++s) {
        if (!h->ov_stream[s]) continue;
        FCD_HIP(h, hipStreamSynchronize(h->ov_stream[s]));
        for (fcd_handle::Flight &f : h->ov_flights[s]) h->ov_event_pool.push_back(f.done);
        h->ov_flights[s].clear();
        h->ov_used[s] = false;
    }
    return FCD_OK;
}

// `stream` waits for every call enqueued on the internal streams so far
int overlap_join(fcd_handle *h, hipStream_t stream) {
    for (int s = 0; s < fcd_handle::kMaxOverlap; ++s)
        if (h->ov_stream[s] && h->ov_used[s]) FCD_HIP(h, hipStreamWaitEvent(stream, h->ov_last[s], 0));
    return FCD_OK;
}

// `S` (internal stream `own_slot`, or the handle's stream: -1) waits for every overlapping call in flight that writes any
// of the output arrays of a call about to be enqueued on it; `mine` receives those arrays' address ranges.
int overlap_order_behind(fcd_handle *h, hipStream_t S, int own_slot, const ResultDesc &o, int64_t n_reads,
                         fcd_handle::Range mine[6], int *n_mine_out) {
    int n_mine = 0;
    auto add = [&](const void *ptr, size_t bytes) {
        if (ptr && bytes) mine[n_mine++] = {reinterpret_cast<uintptr_t>(ptr), reinterpret_cast<uintptr_t>(ptr) + bytes};
    };
    const size_t rows = (size_t)n_reads * (size_t)o.out_stride;
    add(o.labels, rows);
    add(o.path, rows * 4);
    add(o.qual, rows * 4);
    add(o.out_len, (size_t)n_reads * 4);
    add(o.status, (size_t)n_reads * 4);
    add(o.ambiguous, (size_t)n_reads * 8);
    *n_mine_out = n_mine;
    for (int s = 0; s < fcd_handle::kMaxOverlap; ++s) {
        if (!h->ov_stream[s] || !h->ov_used[s]) continue;
        std::deque<fcd_handle::Flight> &fl = h->ov_flights[s];
        while (!fl.empty() && hipEventQuery(fl.front().done) == hipSuccess) {  // finished calls go
            h->ov_event_pool.push_back(fl.front().done);
            fl.pop_front();
        }
        if (fl.empty()) {  // nothing in flight there any more
            h->ov_used[s] = false;
            continue;
        }
        if (s == own_slot) continue;  // (stream order)
        // the YOUNGEST call in flight there that writes any of these arrays: the stream runs them in order
        hipEvent_t wait_for = nullptr;
        for (const fcd_handle::Flight &f : fl) {
            bool clash = false;
            for (int i = 0; i < f.n && !clash; ++i)
                for (int k = 0; k < n_mine && !clash; ++k) clash = f.r[i].lo < mine[k].hi && mine[k].lo < f.r[i].hi;
            if (clash) wait_for = f.done;
        }
        if (wait_for) FCD_HIP(h, hipStreamWaitEvent(S, wait_for, 0));
    }
    return FCD_OK;
}

// every entry point that writes an fcd_result in the handle's stream: behind the overlapping calls that write it too
int overlap_order_writer(fcd_handle *h, const fcd_result *out, int64_t n_reads) {
    bool any = false;
    for (int s = 0; s < fcd_handle::kMaxOverlap; ++s) any = any || h->ov_used[s];
    if (!any) return FCD_OK;
    fcd_handle::Range mine[6];
    int n_mine = 0;
    return overlap_order_behind(h, h->stream, -1, to_desc(out), n_reads, mine, &n_mine);
}

// an entry point about to use the handle's arena from its start, in the handle's stream: behind every overlapping call in
// flight (they hold regions of it) -- which also covers the calls that write its output arrays
int arena_exclusive(fcd_handle *h, const fcd_result *, int64_t) { return overlap_join(h, h->stream); }

// The stream of the next overlapping call: behind the handle's stream as it stands now, and behind every call in flight
// that writes any of this call's output arrays.
int overlap_begin(fcd_handle *h, const ResultDesc &o, int64_t n_reads, int *slot_out) {
    const int n = std::min(h->overlap_n, (int)fcd_handle::kMaxOverlap);
    if (!h->ov_fork) FCD_HIP(h, hipEventCreateWithFlags(&h->ov_fork, hipEventDisableTiming));
    for (int s = 0; s < n; ++s) {
        if (h->ov_stream[s]) continue;
        // The internal streams are created in the HIGH priority class: the runtime hands out the hardware queues of
        // a class separately, so they get queues of their own whatever the process's other streams occupy (normal
        // streams beyond GPU_MAX_HW_QUEUES = 4 share queues, and kernels that share a queue run one after the other:
        // 262 k reads/s instead of 320 k on BASELINE config 3, profiles/r06v_*).  FCD_OVERLAP_PRIORITY=normal|low: A/B.
        const char *pr = getenv("FCD_OVERLAP_PRIORITY");
        int least = 0, greatest = 0;
        const bool ranged = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        if (ranged && (!pr || !strcmp(pr, "high") || !strcmp(pr, "low")))
            FCD_HIP(h, hipStreamCreateWithPriority(&h->ov_stream[s], hipStreamNonBlocking, (pr && !strcmp(pr, "low")) ? least : greatest));
        else
            FCD_HIP(h, hipStreamCreateWithFlags(&h->ov_stream[s], hipStreamNonBlocking));
        FCD_HIP(h, hipEventCreateWithFlags(&h->ov_last[s], hipEventDisableTiming));
    }
    const int slot = (int)(h->ov_seq % (uint64_t)n);
    hipStream_t S = h->ov_stream[slot];
    FCD_HIP(h, hipEventRecord(h->ov_fork, h->stream));
    FCD_HIP(h, hipStreamWaitEvent(S, h->ov_fork, 0));
    fcd_handle::Range mine[6];
    int n_mine = 0;
    int rc = overlap_order_behind(h, S, slot, o, n_reads, mine, &n_mine);
    if (rc) return rc;
    *slot_out = slot;
    return FCD_OK;
}

// ... and the call has been enqueued on internal stream `slot`: from here on it counts as in flight (not earlier: a
// call that has to grow a buffer waits for the others in between, which forgets what was in flight)
int overlap_end(fcd_handle *h, int slot, const ResultDesc &o, int64_t n_reads) {
    FCD_HIP(h, hipEventRecord(h->ov_last[slot], h->ov_stream[slot]));
    fcd_handle::Flight f;
    if (!h->ov_event_pool.empty()) {
        f.done = h->ov_event_pool.back();
        h->ov_event_pool.pop_back();
    } else {
        FCD_HIP(h, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    }
    FCD_HIP(h, hipEventRecord(f.done, h->ov_stream[slot]));
    auto add = [&](const void *ptr, size_t bytes) {
        if (ptr && bytes && f.n < 6) f.r[f.n++] = {reinterpret_cast<uintptr_t>(ptr), reinterpret_cast<uintptr_t>(ptr) + bytes};
    };
    const size_t rows = (size_t)n_reads * (size_t)o.out_stride;
    add(o.labels, rows);
    add(o.path, rows * 4);
    add(o.qual, rows * 4);
    add(o.out_len, (size_t)n_reads * 4);
    add(o.status, (size_t)n_reads * 4);
    add(o.ambiguous, (size_t)n_reads * 8);
    h->ov_flights[slot].push_back(f);
    h->ov_used[slot] = true;
    h->ov_last_slot = slot;
    h->ov_seq++;
    return FCD_OK;
}

// Shared driver of search::beam_search and search::crf_beam_search on device buffers.
int beam_dev(fcd_handle *h, const fcd_batch *in, const BeamArgs &a, int kernel,
             const fcd_result *out) {
    const bool crf = a.crf != 0;
    int rc = check_batch(h, in, crf);
    if (rc) return rc;
    rc = check_result(h, in, out, true);
    if (rc) return rc;
    if (a.beam_size < 1) return fail(h, FCD_E_INVALID, "beam_size cannot be 0");
    if (in->N < 2) return fail(h, FCD_E_UNSUPPORTED, "alphabet needs at least one label besides the blank");
    if (in->n_reads == 0) return FCD_OK;
    FCD_DEVICE(h);

    const BatchDesc d = to_desc(in, crf);
    const ResultDesc o = to_desc(out);
    const int N = d.N, NL = N - 1;
    int64_t beam = a.beam_size;
    BeamArgs args = a;
    args.tie_order = effective_tie_order(h);

    bool use_wave = false;
    if (kernel == FCD_KERNEL_WAVE || kernel == FCD_KERNEL_WAVE1) {
        if (!beam_wave_supported((int)std::min<int64_t>(beam, 1 << 20), N, a.crf, d.S))
            return fail(h, FCD_E_UNSUPPORTED, "wave kernel: needs beam_size <= 8 and N <= 7, or beam_size <= 12 and N <= 5 (CRF: N = 5, S a power of two >= 4)");
        use_wave = true;
    } else if (kernel == FCD_KERNEL_AUTO) {
        use_wave = beam_wave_supported((int)std::min<int64_t>(beam, 1 << 20), N, a.crf, d.S);
    }
    // the wave kernel packs node ids -- (time step << shift) | index among the step's new nodes -- into 25 bits
    const int wave_shift = use_wave ? beam_wave_id_shift((int)std::min<int64_t>(beam, 1 << 20), N, kernel == FCD_KERNEL_WAVE1) : 0;
    if (use_wave && ((d.T << wave_shift) + 16 >= (1ll << 25))) {
        if (kernel != FCD_KERNEL_AUTO) return fail(h, FCD_E_UNSUPPORTED, "wave kernel: T too large");
        use_wave = false;
    }
    args.force_one_read_per_wave = kernel == FCD_KERNEL_WAVE1 ? 1 : 0;
    // wide beams: one beam entry per lane (node ids in 23 bits)
    bool use_lane = false;
    if (kernel == FCD_KERNEL_LANE || (kernel == FCD_KERNEL_AUTO && !use_wave)) {
        const bool ok = beam_lane_supported((int)std::min<int64_t>(beam, 1 << 20), N, a.crf, d.S) &&
                        d.T < (1ll << 26) && d.T * beam * NL + 16 < (1ll << 23);
        if (kernel == FCD_KERNEL_LANE && !ok)
            return fail(h, FCD_E_UNSUPPORTED, "lane kernel: needs beam_size <= 64, N <= 8 (CRF: N = 5, S a power of two >= 4), T * beam_size * (N-1) < 2^23");
        use_lane = ok;
    }

    // Worst-case tree size per read: every step every beam entry creates NL nodes
    // (tree.rs:125 add_node is only called from the expansion loop, search.rs:200-239).
    const int64_t T = std::max<int64_t>(d.T, 1);
    int64_t cap_nodes;
    size_t per_read;
    if (use_wave || use_lane) {
        // (wave kernel: one block of 1 << shift ids per time step, see beam_wave.hip; its records are 4 bytes)
        cap_nodes = use_wave ? ((T << wave_shift) + 8 + 3) & ~3ll
                             : (T * std::min<int64_t>(beam, 64) * NL + 8 + 3) & ~3ll;  // rows stay 16-B aligned
        const int row_words = NL <= 4 ? 4 : 8;
        per_read = (size_t)cap_nodes * (4 + 4 + row_words * 4);
    } else {
        if (beam > (1 << 16)) return fail(h, FCD_E_UNSUPPORTED, "beam_size above 65536");
        if (beam_generic_lds_bytes((int)beam, N, args.tie_order) > 64 * 1024)
            return fail(h, FCD_E_UNSUPPORTED, beam_generic_lds_bytes((int)beam, N, FCD_TIE_STABLE) > 64 * 1024
                                                  ? "beam_size * alphabet too large for the LDS-resident kernel"
                                                  : "beam_size * alphabet too large for the LDS-resident kernel under FCD_TIE_PDQ178 (the "
                                                    "quicksort's list needs LDS too: it fits under FCD_TIE_STABLE)");
        cap_nodes = T * beam * NL + 8;
        if (cap_nodes >= (1ll << 30)) return fail(h, FCD_E_UNSUPPORTED, "tree arena above 2^30 nodes per read");
        per_read = (size_t)cap_nodes * (sizeof(int4) + (size_t)NL * 4);
    }
    args.beam_size = (int)beam;
    const int64_t budget = workspace_budget(h);
    // Wide beams, large jobs: the worst case (every entry creates every child at every step) is about three
    // times what trees actually grow to (172 k of 512 k nodes per read at beam 32) and
    // would reserve 117 GB for BASELINE config 3's 8192 reads.  The lane kernel therefore runs in slabs of HALF
    // the worst case; a read that outgrows its slab is stopped (FCD_ST_INTERNAL) and decoded again by a retry
    // pass in worst-case slabs (stream order: the first pass has finished, results are already traced back).  The
    // retry pass is enqueued unconditionally (see below): the entry point never waits for the device.  Used only
    // when the worst-case arena would exceed 8 GiB (or the workspace limit) -- and then the slabs of both passes are
    // handed out on the device (slab_pool.h), as many as the chip holds wavefronts, whatever the size of the job.
    // A job in which more than a quarter of the reads overflow (dense posteriors: nearly every extension passes
    // the cut) makes this handle size later jobs for the worst case straight away -- one job late.
    // both register kernels keep 4-byte records (parent, label); the creation time is the upper part of the id
    // (wave kernel) or comes from the lane kernel's per-step table of first ids: T + 1 more words per slab
    const size_t rec_bytes = 4;
    const size_t node_bytes = (use_wave || use_lane) ? rec_bytes + 4 + (NL <= 4 ? 4 : 8) * 4 : 0;
    const int64_t first_stride = use_lane ? ((T + 1 + 3) & ~3ll) : 0;
    const size_t first_bytes = (size_t)first_stride * 4;
    if (use_lane) per_read += first_bytes;
    const size_t worst_total = ((size_t)cap_nodes * node_bytes + first_bytes) * (size_t)d.n_reads;
    const bool two_pass = use_lane && (worst_total > ((size_t)8 << 30) || (int64_t)worst_total > budget);
    const int64_t cap_worst = cap_nodes;
    const size_t worst_read = (size_t)cap_worst * node_bytes + first_bytes;
    auto wave_arena = [&](char *base, int64_t slabs, int64_t cap) {
        WaveArena ar{};
        ar.cap_nodes = cap;
        ar.row_words = NL <= 4 ? 4 : 8;
        ar.rec = reinterpret_cast<int32_t *>(base);
        ar.jmp = reinterpret_cast<int32_t *>(base + (size_t)slabs * cap * rec_bytes);
        ar.rows = reinterpret_cast<int32_t *>(base + (size_t)slabs * cap * (rec_bytes + 4));
        ar.first = reinterpret_cast<int32_t *>(base + (size_t)slabs * cap * node_bytes);
        ar.first_stride = first_stride;
        return ar;
    };
    if (two_pass) {
        // ---- slabs from the device-side pool (slab_pool.h): the arena is sized by what the chip holds at once, one
        // launch takes the whole job, and calls on the overlap streams share it ----
        if (h->retry_pending && hipEventQuery(h->retry_ev) == hipSuccess) {
            // the overflow count of an earlier job has arrived: a job in which more than a quarter of the reads outgrew
            // their first-pass slabs makes this handle size later jobs for the worst case straight away
            const int64_t overflowed = *reinterpret_cast<volatile int32_t *>(h->retry_host);
            if (!h->first_pass_div_pinned && overflowed * 4 > h->retry_n) h->first_pass_div = 1;
            h->retry_pending = false;
        }
        cap_nodes = (cap_worst / std::max(h->first_pass_div, 1) + 63) & ~63ll;
        if (cap_nodes > cap_worst) cap_nodes = cap_worst;
        const bool retry_needed = cap_nodes < cap_worst;  // (divisor 1: the first pass IS the worst case)
        const int rpw = beam_lane_reads_per_wave((int)beam);
        const size_t wave_bytes = (size_t)rpw * ((size_t)cap_nodes * node_bytes + first_bytes);
        const int64_t waves = (d.n_reads + rpw - 1) / rpw;
        const bool amb = o.ambiguous != nullptr;
        int64_t p1_want = std::min<int64_t>(slab_pool::kMaxSlabs, beam_lane_resident_waves((int)beam, N, a.crf, true, amb, args.tie_order));
        if (h->overlap_n < 2) p1_want = std::min<int64_t>(p1_want, waves);  // (overlapping calls: whatever the chip holds)
        // the retry pass: a few worst-case slabs however small the job -- never less than one: a read that overflowed
        // must be decodable -- and no more than a quarter of the workspace limit
        int64_t p2 = 0;
        if (retry_needed) {
            p2 = std::min<int64_t>(std::min<int64_t>(d.n_reads, kRetrySlabs),
                                   beam_lane_resident_waves((int)beam, N, a.crf, false, amb, args.tie_order));
            p2 = std::max<int64_t>(1, std::min<int64_t>(p2, budget / 4 / (int64_t)worst_read));
        }
        const int64_t left = budget - p2 * (int64_t)worst_read;
        const int64_t p1 = std::max<int64_t>(1, std::min<int64_t>(p1_want, left / (int64_t)wave_bytes));
        fcd_handle::PoolGeom &g = h->pool_geom;
        const bool fits = h->pool_arena && g.cap_nodes == cap_nodes && g.cap_worst == cap_worst && g.first_stride == first_stride &&
                          g.rpw == rpw && g.row_words == (NL <= 4 ? 4 : 8) && g.p1 >= p1 && g.p2 >= p2 && (g.p2 > 0) == (p2 > 0);
        if (!fits) {
            rc = overlap_drain(h);  // calls in flight still use the old slabs
            if (rc) return rc;
            FCD_HIP(h, hipStreamSynchronize(h->stream));
            if (h->pool_arena) FCD_HIP(h, hipFree(h->pool_arena));
            if (h->pool_ctl) FCD_HIP(h, hipFree(h->pool_ctl));
            h->pool_arena = h->pool_ctl = nullptr;
            h->pool_arena_bytes = h->pool_ctl_bytes = 0;
            g = fcd_handle::PoolGeom();
            const size_t first_region = ((size_t)p1 * wave_bytes + 255) & ~(size_t)255;
            size_t dummy = 0;
            rc = ensure(h, &h->pool_arena, &h->pool_arena_bytes, first_region + (size_t)p2 * worst_read);
            if (rc) return rc;
            rc = ensure(h, &h->pool_ctl, &dummy, slab_pool::bytes((int)p1) + slab_pool::bytes((int)std::max<int64_t>(p2, 1)));
            if (rc) return rc;
            h->pool_ctl_bytes = dummy;
            unsigned long long *ring1 = reinterpret_cast<unsigned long long *>(h->pool_ctl);
            FCD_HIP(h, slab_pool_init(ring1, (int)p1, h->stream));
            if (p2 > 0) FCD_HIP(h, slab_pool_init(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->pool_ctl) + slab_pool::bytes((int)p1)), (int)p2, h->stream));
            g.cap_nodes = cap_nodes;
            g.cap_worst = cap_worst;
            g.first_stride = first_stride;
            g.rpw = rpw;
            g.row_words = NL <= 4 ? 4 : 8;
            g.p1 = (int)p1;
            g.p2 = (int)p2;
        }
        int32_t *d_counter = nullptr;
        if (retry_needed) {  // the overflow counters have their own small allocation: one per call, 64 calls round
            rc = ensure(h, &h->retry_counter, &h->retry_counter_bytes, 256);
            if (rc) return rc;
            d_counter = reinterpret_cast<int32_t *>(h->retry_counter) + (h->ov_seq & 63);
            if (!h->retry_host) {  // (page-locked: the copy back is a DMA nobody waits for)
                if (hipHostMalloc(&h->retry_host, 64, hipHostMallocDefault) != hipSuccess) h->retry_host = nullptr;
                if (h->retry_host && hipEventCreateWithFlags(&h->retry_ev, hipEventDisableTiming) != hipSuccess) {
                    (void)hipHostFree(h->retry_host);
                    h->retry_host = nullptr;
                }
                if (!h->retry_host) {  // (not silently)
                    static std::atomic<bool> said{false};
                    if (!said.exchange(true))
                        fprintf(stderr, "fast_ctc_decode (fcd): no page-locked word for the overflow count of the wide-beam kernel's first "
                                        "pass: the first-pass slab size will not adapt (results are unaffected)\n");
                }
            }
        }
        char *const base = reinterpret_cast<char *>(h->pool_arena);
        const size_t first_region = ((size_t)g.p1 * wave_bytes + 255) & ~(size_t)255;
        unsigned long long *const ring1 = reinterpret_cast<unsigned long long *>(h->pool_ctl);
        unsigned long long *const ring2 = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(h->pool_ctl) + slab_pool::bytes(g.p1));
        // which stream: the handle's, or -- fcd_set_overlap -- the next internal one, behind the handle's stream as it
        // stands now and behind whatever still writes this call's outputs
        hipStream_t S = h->stream;
        int slot = -1;
        if (h->overlap_n >= 2) {
            rc = overlap_begin(h, o, d.n_reads, &slot);
            if (rc) return rc;
            S = h->ov_stream[slot];
        }
        Timer tm(h, S, true);
        WaveArena ar = wave_arena(base, (int64_t)g.p1 * rpw, cap_nodes);
        ar.pool = ring1;
        FCD_HIP(h, launch_beam_lane(d, 0, d.n_reads, args, ar, o, S));
        if (retry_needed) {
            // every read that overflowed is decoded again in a worst-case slab, where it cannot overflow; a wavefront of
            // this pass that finds no slab free waits for one (the pool is a queue), so ONE launch finishes the job
            // whatever happened in the first pass -- with nothing to do it is n wavefronts that read one status word and
            // leave.  Nobody waits for it: this entry point stays enqueue-only.
            WaveArena rr = wave_arena(base + first_region, g.p2, cap_worst);
            rr.pool = ring2;
            rr.retry_counter = d_counter;
            FCD_HIP(h, hipMemsetAsync(d_counter, 0, sizeof(int32_t), S));
            FCD_HIP(h, launch_beam_lane(d, 0, d.n_reads, args, rr, o, S));
            // how many reads overflowed is read back ONE CALL LATE: it only steers the sizing of later jobs
            if (h->retry_host && !h->retry_pending) {
                FCD_HIP(h, hipMemcpyAsync(h->retry_host, d_counter, sizeof(int32_t), hipMemcpyDeviceToHost, S));
                FCD_HIP(h, hipEventRecord(h->retry_ev, S));
                h->retry_pending = true;
                h->retry_n = d.n_reads;
            }
        }
        tm.stop();
        if (slot >= 0) return overlap_end(h, slot, o, d.n_reads);
        h->ov_seq++;
        return FCD_OK;
    }
    // fcd_set_overlap: the call goes to the next internal stream and has a region of the arena to itself (calls on the
    // same internal stream follow one another, so do their uses of its region)
    const int regions = h->overlap_n >= 2 ? std::min(h->overlap_n, (int)fcd_handle::kMaxOverlap) : 1;
    int64_t chunk = std::max<int64_t>(1, budget / regions / (int64_t)per_read);
    chunk = std::min<int64_t>(chunk, d.n_reads);
    size_t region = (size_t)chunk * per_read;
    hipStream_t S = h->stream;
    int slot = -1;
    if (regions > 1) {
        region = (region + 255) & ~(size_t)255;
        if (region > h->arena_region || h->arena_bytes < (size_t)regions * h->arena_region) {
            rc = overlap_drain(h);  // (calls in flight count on the old regions)
            if (rc) return rc;
            rc = ensure(h, &h->arena, &h->arena_bytes, (size_t)regions * region);
            if (rc) return rc;
            h->arena_region = region;
        }
        rc = overlap_begin(h, o, d.n_reads, &slot);
        if (rc) return rc;
        S = h->ov_stream[slot];
    } else {
        rc = arena_exclusive(h, out, d.n_reads);
        if (rc) return rc;
        rc = ensure(h, &h->arena, &h->arena_bytes, region);
        if (rc) return rc;
    }
    char *const abase = reinterpret_cast<char *>(h->arena) + (slot >= 0 ? (size_t)slot * h->arena_region : 0);

    Timer tm(h, S, true);
    for (int64_t begin = 0; begin < d.n_reads; begin += chunk) {
        const int64_t n = std::min<int64_t>(chunk, d.n_reads - begin);
        hipError_t e;
        if (use_wave || use_lane) {
            const WaveArena ar = wave_arena(abase, chunk, cap_nodes);
            e = use_lane ? launch_beam_lane(d, begin, n, args, ar, o, S) : launch_beam_wave(d, begin, n, args, ar, o, S);
            FCD_HIP(h, e);
            continue;
        } else {
            GenericArena ar;
            ar.cap_nodes = cap_nodes;
            ar.rec = reinterpret_cast<int4 *>(abase);
            ar.rows = reinterpret_cast<int32_t *>(abase + (size_t)chunk * cap_nodes * sizeof(int4));
            e = launch_beam_generic(d, begin, n, args, ar, o, S);
        }
        FCD_HIP(h, e);
    }
    tm.stop();
    if (slot >= 0) return overlap_end(h, slot, o, d.n_reads);
    return FCD_OK;
}

// ---- host staging -------------------------------------------------------------------------
// Element span of a strided batch, so that an arbitrarily strided host view can be shipped with
// one copy (non-negative strides only).
int64_t span_elems(const fcd_batch *in, bool crf) {
    if (in->n_reads == 0 || in->T == 0) return 0;
    int64_t s = 1;
    s += (in->n_reads - 1) * in->stride_read;
    s += (in->T - 1) * in->stride_t;
    if (crf) s += (in->S - 1) * in->stride_s;
    s += (in->N - 1) * in->stride_n;
    return s;
}

}  // namespace

namespace fcd {
int effective_tie_order(const fcd_handle *h) {
    return (h && h->tie_order != FCD_TIE_DEFAULT) ? h->tie_order : g_tie_order.load();
}
}  // namespace fcd

// =============================================================================================
extern "C" {

int fcd_version(void) { return FCD_VERSION_MAJOR * 1000 + FCD_VERSION_MINOR; }

int fcd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fcd_create(int device, fcd_handle **out) {
    if (!out) return FCD_E_INVALID;
    *out = nullptr;
    int n = fcd_device_count();
    if (n <= 0 || device < 0 || device >= n) return FCD_E_NODEVICE;
    DeviceGuard dev_guard(device);  // the caller's current device is put back on return
    if (dev_guard.err != hipSuccess) return FCD_E_HIP;
    fcd_handle *h = new fcd_handle();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return FCD_E_HIP;
    }
    h->stream = h->own_stream;
    // the replay's std-form word lives on the device: every device a handle is created on gets the process's value
    if (apply_pdq178_std_form(g_pdq178_std_form.load()) != hipSuccess) {
        (void)hipStreamDestroy(h->own_stream);
        delete h;
        return FCD_E_HIP;
    }
    *out = h;
    return FCD_OK;
}

int fcd_destroy(fcd_handle *h) {
    if (!h) return FCD_OK;
    {
        std::lock_guard<std::recursive_mutex> g(h->mu);
        if (h->job_active) {  // its lane threads still write buffers this call would free: fcd_job_end comes first
            h->err = "a host job is running on this handle (fcd_job_end it first)";
            return FCD_E_INVALID;
        }
    }
    DeviceGuard dev_guard(h->device);
    host_job_release_lanes(h, true);
    (void)overlap_drain(h);
    (void)hipStreamSynchronize(h->stream);
    for (int s = 0; s < fcd_handle::kMaxOverlap; ++s) {
        if (h->ov_last[s]) (void)hipEventDestroy(h->ov_last[s]);
        if (h->ov_stream[s]) (void)hipStreamDestroy(h->ov_stream[s]);
    }
    if (h->ov_fork) (void)hipEventDestroy(h->ov_fork);
    for (hipEvent_t e : h->ov_event_pool) (void)hipEventDestroy(e);  // (overlap_drain above returned every flight's)
    if (h->pool_arena) (void)hipFree(h->pool_arena);
    if (h->pool_ctl) (void)hipFree(h->pool_ctl);
    if (h->arena) (void)hipFree(h->arena);
    if (h->stage) (void)hipFree(h->stage);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->lnbuf) (void)hipFree(h->lnbuf);
    if (h->retry_counter) (void)hipFree(h->retry_counter);
    if (h->retry_host) {
        (void)hipEventDestroy(h->retry_ev);
        (void)hipHostFree(h->retry_host);
        h->retry_host = nullptr;
        h->retry_pending = false;
    }
    for (hipEvent_t e : h->ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev1) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return FCD_OK;
}

int fcd_set_stream(fcd_handle *h, void *hip_stream) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);  // nullptr is the HIP null stream
    return FCD_OK;
}

int fcd_reset_stream(fcd_handle *h) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->stream = h->own_stream;
    return FCD_OK;
}

int fcd_synchronize(fcd_handle *h) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    FCD_DEVICE(h);
    int rc = overlap_drain(h);  // (fcd_set_overlap: the internal streams first)
    if (rc) return rc;
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    return FCD_OK;
}

int fcd_set_overlap(fcd_handle *h, int streams) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (streams < 0 || streams > fcd_handle::kMaxOverlap) return fail(h, FCD_E_INVALID, "fcd_set_overlap: 0 .. 8 streams");
    FCD_DEVICE(h);
    // what is in flight joins the handle's stream: from here on the stream order holds again (or a new round starts)
    int rc = overlap_join(h, h->stream);
    if (rc) return rc;
    h->overlap_n = streams < 2 ? 0 : streams;
    return FCD_OK;
}

int fcd_overlap_join(fcd_handle *h) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    FCD_DEVICE(h);
    return overlap_join(h, h->stream);
}

int fcd_overlap_last_slot(fcd_handle *h) {
    if (!h) return -1;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    return h->overlap_n >= 2 ? h->ov_last_slot : -1;
}

int fcd_overlap_join_slot(fcd_handle *h, int slot, void *hip_stream) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (slot < 0 || slot >= fcd_handle::kMaxOverlap) return fail(h, FCD_E_INVALID, "fcd_overlap_join_slot: no such internal stream");
    FCD_DEVICE(h);
    if (h->ov_stream[slot] && h->ov_used[slot]) FCD_HIP(h, hipStreamWaitEvent(reinterpret_cast<hipStream_t>(hip_stream), h->ov_last[slot], 0));
    return FCD_OK;
}

int fcd_overlap_join_stream(fcd_handle *h, void *hip_stream) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    FCD_DEVICE(h);
    return overlap_join(h, reinterpret_cast<hipStream_t>(hip_stream));
}

const char *fcd_last_error(const fcd_handle *h) { return h ? h->err.c_str() : "null handle"; }

const char *fcd_status_string(int status) {
    switch (status) {  // src/lib.rs:46-53, verbatim
        case FCD_ST_OK: return "";
        case FCD_ST_RAN_OUT_OF_BEAM: return "Ran out of search space (beam_cut_threshold too high)";
        case FCD_ST_INCOMPARABLE: return "Failed to compare values (NaNs in input?)";
        case FCD_ST_INVALID_ENVELOPE: return "Invalid envelope values";
        case FCD_ST_BAD_STATE:
            return "the reference would abort on this input (CRF state or init_state out of range, or an envelope whose "
                   "upper bound moves back below a beam entry's window)";
        case FCD_ST_INTERNAL: return "internal error: tree arena exhausted";
        default: return "unknown status";
    }
}

int fcd_set_workspace_limit(fcd_handle *h, int64_t bytes) {
    if (!h || bytes < 0) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->ws_limit = bytes;
    return FCD_OK;
}

int fcd_release_workspace(fcd_handle *h) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (h->job_active) return fail(h, FCD_E_INVALID, "a host job is running on this handle (fcd_job_end it first)");
    FCD_DEVICE(h);
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->own_stream != h->stream) FCD_HIP(h, hipStreamSynchronize(h->own_stream));
    host_job_release_lanes(h, false);
    {
        int rc = overlap_drain(h);
        if (rc) return rc;
    }
    if (h->pool_arena) (void)hipFree(h->pool_arena);
    if (h->pool_ctl) (void)hipFree(h->pool_ctl);
    h->pool_arena = h->pool_ctl = nullptr;
    h->pool_arena_bytes = h->pool_ctl_bytes = 0;
    h->pool_geom = fcd_handle::PoolGeom();
    if (h->arena) (void)hipFree(h->arena);
    if (h->stage) (void)hipFree(h->stage);
    if (h->lnbuf) (void)hipFree(h->lnbuf);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->retry_counter) (void)hipFree(h->retry_counter);
    if (h->retry_host) {
        (void)hipEventDestroy(h->retry_ev);
        (void)hipHostFree(h->retry_host);
        h->retry_host = nullptr;
        h->retry_pending = false;
    }
    h->arena = h->stage = h->lnbuf = h->pin = h->retry_counter = nullptr;
    h->arena_region = h->lnbuf_region = 0;
    h->arena_bytes = h->stage_bytes = h->lnbuf_bytes = h->pin_bytes = h->retry_counter_bytes = 0;
    return FCD_OK;
}

int fcd_set_tie_order(fcd_handle *h, int order) {
    if (!h || (order != FCD_TIE_DEFAULT && order != FCD_TIE_PDQ178 && order != FCD_TIE_STABLE)) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->tie_order = order;
    return FCD_OK;
}

int fcd_get_tie_order(const fcd_handle *h) { return effective_tie_order(h); }

int fcd_set_default_tie_order(int order) {
    if (order != FCD_TIE_PDQ178 && order != FCD_TIE_STABLE) return FCD_E_INVALID;
    g_tie_order.store(order);
    return FCD_OK;
}

int fcd_debug_set_pdq178_std_form(fcd_handle *h, int bits) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (bits < 0 || bits > 3) return fail(h, FCD_E_INVALID, "the std form is 0, 1, 2 or 3");
    FCD_DEVICE(h);
    FCD_HIP(h, hipStreamSynchronize(h->stream));  // (no launch of this handle reads the word while it changes)
    FCD_HIP(h, apply_pdq178_std_form(bits));
    g_pdq178_std_form.store(bits);
    return FCD_OK;
}

int fcd_debug_get_pdq178_std_form(void) { return g_pdq178_std_form.load(); }

int fcd_debug_pdq178_sort_dev(fcd_handle *h, uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (n_lists < 0 || stride < 0 || (n_lists > 0 && (!lists || !lens))) return fail(h, FCD_E_INVALID, "bad argument");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_pdq178_probe(lists, n_lists, stride, lens, h->stream));
    return FCD_OK;
}

int fcd_debug_pdq178_coop_sort_dev(fcd_handle *h, uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens,
                                   int planes, int keep) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (n_lists < 0 || stride < 0 || (n_lists > 0 && (!lists || !lens)) || (planes != 1 && planes != 3 && planes != 5 && planes != 8) || keep < 1)
        return fail(h, FCD_E_INVALID, "bad argument");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_pdq178_coop_probe(lists, n_lists, stride, lens, planes, keep, h->stream));
    return FCD_OK;
}

int fcd_debug_pdq178_coop_profile(fcd_handle *h, uint64_t cycles[16], int reset) {
    if (!h || !cycles) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    FCD_DEVICE(h);
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    // bit 0: reset afterwards; bit 1: the wide-beam kernel's in-place counters (a -DFCD_LANE_TIE_PROF build) instead of the probe kernels'
    if (reset & 2) FCD_HIP(h, lane_tie_prof_read(reinterpret_cast<unsigned long long *>(cycles), (reset & 1) != 0));
    else FCD_HIP(h, coop_prof_read(reinterpret_cast<unsigned long long *>(cycles), (reset & 1) != 0));
    return FCD_OK;
}

int fcd_debug_set_first_pass_divisor(fcd_handle *h, int divisor) {
    if (!h || divisor < 0) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->first_pass_div = divisor ? divisor : 2;
    h->first_pass_div_pinned = divisor != 0;  // 0: back to the default, adaptive sizing
    return FCD_OK;
}

int fcd_debug_set_duplex_profile(fcd_handle *h, uint32_t *cycles) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->duplex_prof = cycles;
    return FCD_OK;
}

int fcd_debug_set_duplex_kernel(fcd_handle *h, int which) {
    if (!h || which < 0 || which > 2) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->duplex_kernel = which;
    return FCD_OK;
}

double fcd_last_kernel_ms(fcd_handle *h) {
    if (!h) return -1.0;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (h->n_timed == 0) return -1.0;
    const int slot = (int)((h->n_timed - 1) % fcd_handle::kTimingRing);
    if (hipEventSynchronize(h->ev1[slot]) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, h->ev0[slot], h->ev1[slot]) != hipSuccess) return -1.0;
    return (double)ms;
}

int fcd_timing_reset(fcd_handle *h) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    h->n_timed = 0;
    return FCD_OK;
}

double fcd_timing_mean_ms(fcd_handle *h, int64_t *n_calls) {
    if (!h) return -1.0;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    const int64_t n = std::min<int64_t>(h->n_timed, fcd_handle::kTimingRing);
    if (n_calls) *n_calls = n;
    if (n == 0) return -1.0;
    double sum = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float ms = 0.0f;
        if (hipEventSynchronize(h->ev1[i]) != hipSuccess) return -1.0;
        if (hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]) != hipSuccess) return -1.0;
        sum += ms;
    }
    return sum / (double)n;
}

// ---- viterbi -------------------------------------------------------------------------------
int fcd_viterbi_search_dev(fcd_handle *h, const fcd_batch *in, int collapse_repeats,
                           const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    int rc = check_batch(h, in, false);
    if (rc) return rc;
    rc = check_result(h, in, out, false);
    if (rc) return rc;
    if (in->n_reads == 0) return FCD_OK;
    FCD_DEVICE(h);
    rc = overlap_order_writer(h, out, in->n_reads);
    if (rc) return rc;
    Timer tm(h);
    FCD_HIP(h, launch_viterbi(to_desc(in, false), collapse_repeats, to_desc(out), h->stream));
    tm.stop();
    return FCD_OK;
}

int fcd_beam_search_dev(fcd_handle *h, const fcd_batch *in, int64_t beam_size,
                        float beam_cut_threshold, int collapse_repeats, int kernel,
                        const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (beam_size < 1) return fail(h, FCD_E_INVALID, "beam_size cannot be 0");
    BeamArgs a{};
    a.beam_size = (int)std::min<int64_t>(beam_size, 1ll << 30);
    a.thr = beam_cut_threshold;
    a.collapse = collapse_repeats ? 1 : 0;
    a.crf = 0;
    return beam_dev(h, in, a, kernel, out);
}

int fcd_beam_search_profile_dev(fcd_handle *h, const fcd_batch *in, int64_t beam_size,
                                float beam_cut_threshold, int collapse_repeats, const fcd_result *out,
                                uint32_t *cycles) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (beam_size < 1 || beam_size > 5 || !in || in->N != 5 || !cycles)
        return fail(h, FCD_E_UNSUPPORTED, "profile: the instrumented instantiation is beam_size <= 5, N = 5");
    BeamArgs a{};
    a.beam_size = (int)beam_size;
    a.thr = beam_cut_threshold;
    a.collapse = collapse_repeats ? 1 : 0;
    a.crf = 0;
    a.prof = cycles;
    return beam_dev(h, in, a, FCD_KERNEL_WAVE, out);
}

int fcd_crf_beam_search_dev(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                            int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                            const fcd_result *out) {
    return fcd_crf_beam_search_dev_k(h, in, init, n_init, init_stride, beam_size, beam_cut_threshold,
                                     FCD_KERNEL_AUTO, out);
}

int fcd_crf_beam_search_dev_k(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                              int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                              int kernel, const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (beam_size < 1) return fail(h, FCD_E_INVALID, "beam_size cannot be 0");
    if (!init || n_init < 1) return fail(h, FCD_E_INVALID, "init_state missing");
    BeamArgs a{};
    a.beam_size = (int)std::min<int64_t>(beam_size, 1ll << 30);
    a.thr = beam_cut_threshold;
    a.collapse = 0;
    a.crf = 1;
    a.init = init;
    a.n_init = n_init;
    a.init_stride = init_stride;
    return beam_dev(h, in, a, kernel, out);
}

int fcd_crf_greedy_search_dev(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                              int64_t init_stride, const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    int rc = check_batch(h, in, true);
    if (rc) return rc;
    rc = check_result(h, in, out, true);
    if (rc) return rc;
    if (!init || n_init < 1) return fail(h, FCD_E_INVALID, "init_state missing");
    if (in->n_reads == 0) return FCD_OK;
    FCD_DEVICE(h);
    rc = overlap_order_writer(h, out, in->n_reads);
    if (rc) return rc;
    Timer tm(h);
    FCD_HIP(h, launch_crf_greedy(to_desc(in, true), init, n_init, init_stride, to_desc(out), h->stream));
    tm.stop();
    return FCD_OK;
}

namespace {
struct CrfInit {
    const float *init1 = nullptr, *init2 = nullptr;
    int64_t n1 = 0, s1 = 0, n2 = 0, s2 = 0;
};

// Shared driver of duplex::beam_search (crf == nullptr) and duplex::crf_beam_search.
int duplex_dev(fcd_handle *h, const fcd_batch *in1, const fcd_batch *in2, const CrfInit *crf,
               const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
               float beam_cut_threshold, int collapse_repeats, int logadd_mode,
               const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    const bool is_crf = crf != nullptr;
    int rc = check_batch(h, in1, is_crf);
    if (rc) return rc;
    rc = check_batch(h, in2, is_crf);
    if (rc) return rc;
    if (is_crf && in1->S != in2->S) return fail(h, FCD_E_INVALID, "state counts of the two network outputs differ");
    if (is_crf && (!crf->init1 || !crf->init2 || crf->n1 < 1 || crf->n2 < 1))
        return fail(h, FCD_E_INVALID, "init_state missing");
    const int S = is_crf ? (int)in1->S : 1;
    if (in1->n_reads != in2->n_reads) return fail(h, FCD_E_INVALID, "pair counts differ");
    if (in1->N != in2->N) return fail(h, FCD_E_INVALID, "inner axes of the network outputs do not match");
    if (in1->N < 2) return fail(h, FCD_E_UNSUPPORTED, "alphabet needs at least one label besides the blank");
    if (beam_size < 1) return fail(h, FCD_E_INVALID, "beam_size cannot be 0");
    if (logadd_mode != FCD_LOGADD_LOGSUMEXP && logadd_mode != FCD_LOGADD_MAX && logadd_mode != FCD_LOGADD_LOGSUMEXP_GLIBC235)
        return fail(h, FCD_E_INVALID, "unknown logadd_mode");
    if (!out) return fail(h, FCD_E_INVALID, "null result");
    const int64_t B = in1->n_reads;
    if (B == 0) return FCD_OK;
    if (!out->labels || !out->out_len || !out->status) return fail(h, FCD_E_INVALID, "null output array");
    if (out->out_stride < in1->T) return fail(h, FCD_E_INVALID, "out_stride must be >= T1");
    if (!envelope || env_stride < in1->T) return fail(h, FCD_E_INVALID, "envelope missing or shorter than read 1");
    if (beam_size > (1 << 12)) return fail(h, FCD_E_UNSUPPORTED, "beam_size above 4096");
    const int N = (int)in1->N, NL = N - 1;
    if (duplex_lds_bytes((int)beam_size, N, 0, S, effective_tie_order(h)) > 64 * 1024)
        return fail(h, FCD_E_UNSUPPORTED, "beam_size * alphabet too large for the LDS-resident kernel");
    FCD_DEVICE(h);

    // log-space copies + one int for the envelope width.  fcd_set_overlap: the call goes to the next internal stream, with
    // a region of the log-space buffer and of the arena to itself (beam_dev)
    const int64_t T1 = std::max<int64_t>(in1->T, 1), T2 = std::max<int64_t>(in2->T, 1);
    const size_t n1 = (size_t)B * T1 * S * N, n2 = (size_t)B * T2 * S * N;
    const int regions = h->overlap_n >= 2 ? std::min(h->overlap_n, (int)fcd_handle::kMaxOverlap) : 1;
    size_t ln_need = (n1 + n2) * 4 + 256;
    hipStream_t St = h->stream;
    int slot = -1;
    if (regions > 1) {
        ln_need = (ln_need + 255) & ~(size_t)255;
        if (ln_need > h->lnbuf_region || h->lnbuf_bytes < (size_t)regions * h->lnbuf_region) {
            rc = overlap_drain(h);
            if (rc) return rc;
            rc = ensure(h, &h->lnbuf, &h->lnbuf_bytes, (size_t)regions * ln_need);
            if (rc) return rc;
            h->lnbuf_region = ln_need;
        }
        rc = overlap_begin(h, to_desc(out), B, &slot);
        if (rc) return rc;
        St = h->ov_stream[slot];
    } else {
        rc = arena_exclusive(h, out, B);  // (behind every overlapping call in flight: they hold regions of both buffers)
        if (rc) return rc;
        rc = ensure(h, &h->lnbuf, &h->lnbuf_bytes, ln_need);
        if (rc) return rc;
    }
    float *ln1 = reinterpret_cast<float *>(reinterpret_cast<char *>(h->lnbuf) + (slot >= 0 ? (size_t)slot * h->lnbuf_region : 0));
    float *ln2 = ln1 + n1;
    int *d_width = reinterpret_cast<int *>(ln2 + n2);
    Timer tm(h, St, true);
    FCD_HIP(h, launch_ln_convert(static_cast<const float *>(in1->post), in1->dtype, B, in1->T, S, N, in1->stride_read, in1->stride_t,
                                 is_crf ? in1->stride_s : 0, in1->stride_n, ln1, logadd_mode == FCD_LOGADD_LOGSUMEXP_GLIBC235, St));
    FCD_HIP(h, launch_ln_convert(static_cast<const float *>(in2->post), in2->dtype, B, in2->T, S, N, in2->stride_read, in2->stride_t,
                                 is_crf ? in2->stride_s : 0, in2->stride_n, ln2, logadd_mode == FCD_LOGADD_LOGSUMEXP_GLIBC235, St));
    FCD_HIP(h, hipMemsetAsync(d_width, 0, sizeof(int), St));
    FCD_HIP(h, launch_env_width(envelope, B, env_stride, in1->T, in2->T, in1->lengths,
                                in2->lengths, d_width, St));
    int width = 0;
    FCD_HIP(h, hipMemcpyAsync(&width, d_width, sizeof(int), hipMemcpyDeviceToHost, St));
    FCD_HIP(h, hipStreamSynchronize(St));  // ring capacity is needed to size the arena
    // Which kernel: the slot-resident one (duplex_slots.hip) wherever it fits -- beam_size * N <= 64 and the live nodes'
    // rings next to the read-2 tile in 64 KiB of LDS -- the any-shape one (duplex.hip) otherwise.
    const int tie = effective_tie_order(h);
    static const int env_kernel = [] {
        const char *e = getenv("FCD_DUPLEX_KERNEL");  // "legacy" / "slots": A/B and test aid
        return e ? (!strcmp(e, "legacy") ? 1 : (!strcmp(e, "slots") ? 2 : 0)) : 0;
    }();
    const int want = h->duplex_kernel ? h->duplex_kernel : env_kernel;
    const bool fits = duplex_slots_supported((int)beam_size, N, S, std::max(width, 1), tie);
    if (want == 2 && !fits) return fail(h, FCD_E_UNSUPPORTED, "duplex kernel: the slot-resident kernel does not cover this shape");
    const int NLp = (NL + 3) & ~3;
    const int64_t cap_nodes = (std::max<int64_t>(in1->T, 1) * beam_size * NL + 8 + 3) & ~3ll;
    if (cap_nodes >= (1ll << 30)) return fail(h, FCD_E_UNSUPPORTED, "tree arena above 2^30 nodes per pair");
    // (the slot-resident kernel addresses a pair's slab and its log-space reads with 32-bit byte offsets)
    const bool small = fits && duplex_slots_pair_bytes(cap_nodes, N, duplex_slots_ring_rows(std::max(width, 1)), in2->T) < (1ull << 32) &&
                       (uint64_t)std::max(in1->T, in2->T) * S * N * 4 < (1ull << 31);
    if (want == 2 && fits && !small) return fail(h, FCD_E_UNSUPPORTED, "duplex kernel: a pair's arena above 4 GiB");
    const bool slots = small && want != 1;
    const int Wcap = slots ? duplex_slots_ring_rows(std::max(width, 1)) : std::max(width, 1) + 2;
    const size_t per_pair = slots ? duplex_slots_pair_bytes(cap_nodes, N, Wcap, in2->T)
                                  : (size_t)cap_nodes * (sizeof(int4) + 8 + (size_t)NL * 4 + (size_t)Wcap * 12) +
                                        (((size_t)(in2->T + 1) * 4 + 64 + 15) & ~(size_t)15);
    const int64_t budget = workspace_budget(h);
    int64_t chunk = std::max<int64_t>(1, budget / regions / (int64_t)per_pair);
    chunk = std::min<int64_t>(chunk, B);
    if (regions > 1) {
        const size_t region = ((size_t)chunk * per_pair + 255) & ~(size_t)255;
        if (region > h->arena_region || h->arena_bytes < (size_t)regions * h->arena_region) {
            rc = overlap_drain(h);
            if (rc) return rc;
            rc = ensure(h, &h->arena, &h->arena_bytes, (size_t)regions * region);
            if (rc) return rc;
            h->arena_region = region;
        }
    } else {
        rc = ensure(h, &h->arena, &h->arena_bytes, (size_t)chunk * per_pair);
        if (rc) return rc;
    }
    char *const abase = reinterpret_cast<char *>(h->arena) + (slot >= 0 ? (size_t)slot * h->arena_region : 0);

    DuplexArgs a;
    a.ln1 = ln1; a.ln2 = ln2; a.T1cap = in1->T; a.T2cap = in2->T;
    a.len1 = in1->lengths; a.len2 = in2->lengths; a.env = envelope; a.env_stride = env_stride;
    a.N = N; a.beam_size = (int)beam_size;
    // ln(threshold) with the kernels' definition of ln: correctly rounded f32 (duplex.rs:454)
    a.thr_ln = logadd_mode == FCD_LOGADD_LOGSUMEXP_GLIBC235 ? g235::logf235(beam_cut_threshold)
                                                           : (float)log((double)beam_cut_threshold);
    a.collapse = collapse_repeats ? 1 : 0; a.mode = logadd_mode;
    a.S = S; a.crf = is_crf ? 1 : 0;
    a.init1 = is_crf ? crf->init1 : nullptr; a.init2 = is_crf ? crf->init2 : nullptr;
    a.n_init1 = is_crf ? crf->n1 : 0; a.n_init2 = is_crf ? crf->n2 : 0;
    a.init1_stride = is_crf ? crf->s1 : 0; a.init2_stride = is_crf ? crf->s2 : 0;
    char *base = abase;
    a.meta = reinterpret_cast<int4 *>(base); base += (size_t)chunk * cap_nodes * sizeof(int4);
    a.aux = nullptr; a.nmax = nullptr; a.rlo = nullptr; a.NLp = NLp; a.pair_stride = (int64_t)per_pair;
    if (slots) {
        a.meta = reinterpret_cast<int4 *>(abase);  // (one slab per pair: duplex_slots.hip)
        a.vec = nullptr; a.rows = nullptr;
    } else {
        a.nmax = reinterpret_cast<float *>(base); base += (size_t)chunk * cap_nodes * 4;
        a.rlo = reinterpret_cast<int32_t *>(base); base += (size_t)chunk * cap_nodes * 4;
        a.rows = reinterpret_cast<int32_t *>(base); base += (size_t)chunk * cap_nodes * NL * 4;
        a.vec = reinterpret_cast<float *>(base); base += (size_t)chunk * cap_nodes * (size_t)Wcap * 12;
    }
    a.rootgap = reinterpret_cast<float *>(base);
    a.cap_nodes = cap_nodes; a.Wcap = Wcap;
    // any-shape kernel: tile the envelope window through LDS when it fits next to the beam (48 KiB budget)
    // (and keep the beam entries' windows resident there: one LDS buffer per beam slot, handed over by lane votes)
    a.staged = !slots && duplex_lds_bytes((int)beam_size, N, Wcap - 2, S, tie) <= 48 * 1024 && beam_size <= 64 ? 1 : 0;
    a.out = to_desc(out);
    a.prof = h->duplex_prof;
    a.tie_order = tie;
    for (int64_t begin = 0; begin < B; begin += chunk) {
        const int64_t n = std::min<int64_t>(chunk, B - begin);
        FCD_HIP(h, slots ? launch_duplex_slots(a, begin, n, St) : launch_duplex(a, begin, n, St));
    }
    tm.stop();
    if (slot >= 0) return overlap_end(h, slot, to_desc(out), B);
    return FCD_OK;
}
}  // namespace

int fcd_beam_search_duplex_dev(fcd_handle *h, const fcd_batch *in1, const fcd_batch *in2,
                               const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                               float beam_cut_threshold, int collapse_repeats, int logadd_mode,
                               const fcd_result *out) {
    return duplex_dev(h, in1, in2, nullptr, envelope, env_stride, beam_size, beam_cut_threshold,
                      collapse_repeats, logadd_mode, out);
}

int fcd_crf_beam_search_duplex_dev(fcd_handle *h, const fcd_batch *in1, const float *init1,
                                   int64_t n_init1, int64_t init1_stride, const fcd_batch *in2,
                                   const float *init2, int64_t n_init2, int64_t init2_stride,
                                   const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                                   float beam_cut_threshold, int logadd_mode, const fcd_result *out) {
    CrfInit c;
    c.init1 = init1; c.n1 = n_init1; c.s1 = init1_stride;
    c.init2 = init2; c.n2 = n_init2; c.s2 = init2_stride;
    return duplex_dev(h, in1, in2, &c, envelope, env_stride, beam_size, beam_cut_threshold, 0,
                      logadd_mode, out);
}

namespace {
// host staging shared by fcd_beam_search_duplex_host / fcd_crf_beam_search_duplex_host
int duplex_host(fcd_handle *h, const fcd_batch *in1, const fcd_batch *in2, const CrfInit *crf,
                const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                float beam_cut_threshold, int collapse_repeats, int logadd_mode,
                const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> whole_call(h->mu);  // staging .. copy-back, see run_host
    const bool is_crf = crf != nullptr;
    {
        std::lock_guard<std::recursive_mutex> g(h->mu);
        int rc = check_batch(h, in1, is_crf);
        if (rc) return rc;
        rc = check_batch(h, in2, is_crf);
        if (rc) return rc;
        if (in1->n_reads != in2->n_reads) return fail(h, FCD_E_INVALID, "pair counts differ");
        if (!out || !envelope) return fail(h, FCD_E_INVALID, "null result/envelope");
        if (in1->stride_read < 0 || in1->stride_t < 0 || in1->stride_n < 0 || in2->stride_read < 0 ||
            in2->stride_t < 0 || in2->stride_n < 0 || (is_crf && (in1->stride_s < 0 || in2->stride_s < 0)))
            return fail(h, FCD_E_UNSUPPORTED, "negative strides: pass a contiguous copy");
        if (is_crf && (!crf->init1 || !crf->init2 || crf->n1 < 1 || crf->n2 < 1))
            return fail(h, FCD_E_INVALID, "init_state missing");
    }
    const int64_t B = in1->n_reads;
    if (B == 0) return FCD_OK;
    if (!out->labels || !out->out_len || !out->status) return fail(h, FCD_E_INVALID, "null output array");
    FCD_DEVICE(h);  // staging, search and copy-back all run on the handle's device
    const size_t e1 = (size_t)span_elems(in1, is_crf), e2 = (size_t)span_elems(in2, is_crf);
    const size_t n_env = (size_t)B * (size_t)env_stride * 2;
    const size_t n_out = (size_t)B * (size_t)out->out_stride;
    const size_t ni1 = is_crf ? (size_t)((B - 1) * crf->s1 + crf->n1) : 0;
    const size_t ni2 = is_crf ? (size_t)((B - 1) * crf->s2 + crf->n2) : 0;
    size_t used = 0;
    auto reserve = [&](size_t bytes) {
        size_t off = (used + 255) & ~(size_t)255;
        used = off + std::max<size_t>(bytes, 8);
        return off;
    };
    const size_t z1 = in1->dtype == FCD_DTYPE_F32 ? 4 : 2, z2 = in2->dtype == FCD_DTYPE_F32 ? 4 : 2;
    const size_t o1 = reserve(e1 * z1), o2 = reserve(e2 * z2), oe = reserve(n_env * 8);
    const size_t ol1 = reserve(in1->lengths ? (size_t)B * 8 : 0);
    const size_t ol2 = reserve(in2->lengths ? (size_t)B * 8 : 0);
    const size_t oi1 = reserve(ni1 * 4), oi2 = reserve(ni2 * 4);
    const size_t olab = reserve(n_out), oolen = reserve((size_t)B * 4), ostat = reserve((size_t)B * 4);
    const size_t oamb = reserve(out->ambiguous ? (size_t)B * 8 : 0);
    {
        std::lock_guard<std::recursive_mutex> g(h->mu);
        int rc = ensure(h, &h->stage, &h->stage_bytes, used);
        if (rc) return rc;
        char *base = reinterpret_cast<char *>(h->stage);
        if (e1) FCD_HIP(h, hipMemcpyAsync(base + o1, in1->post, e1 * z1, hipMemcpyHostToDevice, h->stream));
        if (e2) FCD_HIP(h, hipMemcpyAsync(base + o2, in2->post, e2 * z2, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemcpyAsync(base + oe, envelope, n_env * 8, hipMemcpyHostToDevice, h->stream));
        if (in1->lengths)
            FCD_HIP(h, hipMemcpyAsync(base + ol1, in1->lengths, (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
        if (in2->lengths)
            FCD_HIP(h, hipMemcpyAsync(base + ol2, in2->lengths, (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
        if (is_crf) {
            FCD_HIP(h, hipMemcpyAsync(base + oi1, crf->init1, ni1 * 4, hipMemcpyHostToDevice, h->stream));
            FCD_HIP(h, hipMemcpyAsync(base + oi2, crf->init2, ni2 * 4, hipMemcpyHostToDevice, h->stream));
        }
    }
    char *base = reinterpret_cast<char *>(h->stage);
    fcd_batch d1 = *in1, d2 = *in2;
    d1.post = reinterpret_cast<const float *>(base + o1);
    d2.post = reinterpret_cast<const float *>(base + o2);
    d1.lengths = in1->lengths ? reinterpret_cast<const int64_t *>(base + ol1) : nullptr;
    d2.lengths = in2->lengths ? reinterpret_cast<const int64_t *>(base + ol2) : nullptr;
    fcd_result dout{};
    dout.labels = reinterpret_cast<uint8_t *>(base + olab);
    dout.out_len = reinterpret_cast<uint32_t *>(base + oolen);
    dout.status = reinterpret_cast<int32_t *>(base + ostat);
    dout.out_stride = out->out_stride;
    dout.ambiguous = out->ambiguous ? reinterpret_cast<uint32_t *>(base + oamb) : nullptr;
    CrfInit dc;
    if (is_crf) {
        dc = *crf;
        dc.init1 = reinterpret_cast<const float *>(base + oi1);
        dc.init2 = reinterpret_cast<const float *>(base + oi2);
    }
    int rc = duplex_dev(h, &d1, &d2, is_crf ? &dc : nullptr, reinterpret_cast<const uint64_t *>(base + oe),
                        env_stride, beam_size, beam_cut_threshold, collapse_repeats, logadd_mode, &dout);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (h->overlap_n >= 2) {  // (fcd_set_overlap: the search may sit on an internal stream; the copies below are in the handle's)
        rc = overlap_join(h, h->stream);
        if (rc) return rc;
    }
    FCD_HIP(h, hipMemcpyAsync(out->labels, dout.labels, n_out, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipMemcpyAsync(out->out_len, dout.out_len, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipMemcpyAsync(out->status, dout.status, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (out->ambiguous)
        FCD_HIP(h, hipMemcpyAsync(out->ambiguous, dout.ambiguous, (size_t)B * 8, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    return FCD_OK;
}
}  // namespace

int fcd_beam_search_duplex_host(fcd_handle *h, const fcd_batch *in1, const fcd_batch *in2,
                                const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                                float beam_cut_threshold, int collapse_repeats, int logadd_mode,
                                const fcd_result *out) {
    return duplex_host(h, in1, in2, nullptr, envelope, env_stride, beam_size, beam_cut_threshold,
                       collapse_repeats, logadd_mode, out);
}

int fcd_crf_beam_search_duplex_host(fcd_handle *h, const fcd_batch *in1, const float *init1,
                                    int64_t n_init1, int64_t init1_stride, const fcd_batch *in2,
                                    const float *init2, int64_t n_init2, int64_t init2_stride,
                                    const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                                    float beam_cut_threshold, int logadd_mode, const fcd_result *out) {
    CrfInit c;
    c.init1 = init1; c.n1 = n_init1; c.s1 = init1_stride;
    c.init2 = init2; c.n2 = n_init2; c.s2 = init2_stride;
    return duplex_host(h, in1, in2, &c, envelope, env_stride, beam_size, beam_cut_threshold, 0,
                       logadd_mode, out);
}

// ---- alignment-band estimator (envelope.hip) ----
int fcd_duplex_envelope_dev(fcd_handle *h, int64_t n_pairs,
                            const uint8_t *labels1, const uint32_t *path1, const uint32_t *len1,
                            int64_t stride1, const int64_t *T1, int64_t T1cap,
                            const uint8_t *labels2, const uint32_t *path2, const uint32_t *len2,
                            int64_t stride2, const int64_t *T2, int64_t T2cap,
                            int64_t band, uint64_t *envelope, int64_t env_stride) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (n_pairs < 0 || T1cap < 0 || T2cap < 0 || band < 0) return fail(h, FCD_E_INVALID, "negative size");
    if (n_pairs == 0 || T1cap == 0) return FCD_OK;
    if (!labels1 || !path1 || !len1 || !labels2 || !path2 || !len2 || !envelope)
        return fail(h, FCD_E_INVALID, "null array");
    if (stride1 < T1cap || stride2 < T2cap) return fail(h, FCD_E_INVALID, "label strides shorter than the reads");
    if (env_stride < T1cap) return fail(h, FCD_E_INVALID, "envelope shorter than read 1");
    band = std::min<int64_t>(band, 1 << 20);
    FCD_DEVICE(h);
    // the DP is sized by the longest labellings actually present, not by the time axes
    int rc = ensure(h, &h->lnbuf, &h->lnbuf_bytes, 256);
    if (rc) return rc;
    uint32_t *d_max = reinterpret_cast<uint32_t *>(h->lnbuf);
    uint32_t h_max[2] = {0, 0};
    FCD_HIP(h, hipMemsetAsync(d_max, 0, 8, h->stream));
    FCD_HIP(h, launch_max_u32(len1, len2, n_pairs, d_max, h->stream));
    FCD_HIP(h, hipMemcpyAsync(h_max, d_max, 8, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    const int64_t L1cap = std::max<int64_t>(1, std::min<int64_t>(h_max[0], T1cap));
    const int64_t L2cap = std::max<int64_t>(1, std::min<int64_t>(h_max[1], T2cap));
    if (L1cap + L2cap > 65535) return fail(h, FCD_E_UNSUPPORTED, "envelope estimator: more than 65535 labels in a pair");
    if (envelope_lds_bytes(L2cap) > 64 * 1024)
        return fail(h, FCD_E_UNSUPPORTED, "envelope estimator: read 2 holds too many labels for the LDS-resident rows");
    const int nchunk = (int)((L2cap + 63) / 64);
    const int64_t dirs_stride = std::max<int64_t>(L1cap * std::max(nchunk, 1) * 2, 2);
    const size_t anchor_bytes = ((size_t)(T1cap + 1) * 4 + 15) & ~(size_t)15;
    const size_t per_pair = (size_t)dirs_stride * 8 + anchor_bytes;
    int64_t chunk = std::max<int64_t>(1, workspace_budget(h) / (int64_t)per_pair);
    chunk = std::min<int64_t>(chunk, n_pairs);
    rc = arena_exclusive(h, nullptr, 0);
    if (rc) return rc;
    rc = ensure(h, &h->arena, &h->arena_bytes, (size_t)chunk * per_pair);
    if (rc) return rc;
    EnvelopeArgs a;
    a.labels1 = labels1; a.labels2 = labels2; a.path1 = path1; a.path2 = path2;
    a.len1 = len1; a.len2 = len2; a.stride1 = stride1; a.stride2 = stride2;
    a.T1 = T1; a.T2 = T2; a.T1cap = T1cap; a.T2cap = T2cap; a.L2cap = L2cap; a.band = band;
    a.env = envelope; a.env_stride = env_stride;
    a.dirs = reinterpret_cast<uint64_t *>(h->arena);
    a.dirs_stride = dirs_stride;
    a.nchunk = std::max(nchunk, 1);
    a.anchor = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(h->arena) + (size_t)chunk * dirs_stride * 8);
    // anchor slabs are (T1cap + 1) ints apart inside their region
    Timer tm(h);
    for (int64_t b0 = 0; b0 < n_pairs; b0 += chunk)
        FCD_HIP(h, launch_envelope(a, b0, std::min<int64_t>(chunk, n_pairs - b0), h->stream));
    tm.stop();
    return FCD_OK;
}

int fcd_duplex_envelope_host(fcd_handle *h, int64_t n_pairs,
                             const uint8_t *labels1, const uint32_t *path1, const uint32_t *len1,
                             int64_t stride1, const int64_t *T1, int64_t T1cap,
                             const uint8_t *labels2, const uint32_t *path2, const uint32_t *len2,
                             int64_t stride2, const int64_t *T2, int64_t T2cap,
                             int64_t band, uint64_t *envelope, int64_t env_stride) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> whole_call(h->mu);  // staging .. copy-back, see run_host
    if (n_pairs < 0 || T1cap < 0 || T2cap < 0) return fail(h, FCD_E_INVALID, "negative size");
    if (n_pairs == 0 || T1cap == 0) return FCD_OK;
    if (!labels1 || !path1 || !len1 || !labels2 || !path2 || !len2 || !envelope)
        return fail(h, FCD_E_INVALID, "null array");
    if (stride1 < T1cap || stride2 < T2cap || env_stride < T1cap) return fail(h, FCD_E_INVALID, "strides shorter than the reads");
    FCD_DEVICE(h);
    // one device slab: labels, paths, lengths, row counts of both reads, then the envelope
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_l1 = 0, o_l2 = o_l1 + al((size_t)n_pairs * stride1);
    const size_t o_p1 = o_l2 + al((size_t)n_pairs * stride2), o_p2 = o_p1 + al((size_t)n_pairs * stride1 * 4);
    const size_t o_n1 = o_p2 + al((size_t)n_pairs * stride2 * 4), o_n2 = o_n1 + al((size_t)n_pairs * 4);
    const size_t o_t1 = o_n2 + al((size_t)n_pairs * 4), o_t2 = o_t1 + al((size_t)n_pairs * 8);
    const size_t o_env = o_t2 + al((size_t)n_pairs * 8);
    const size_t total = o_env + (size_t)n_pairs * env_stride * 16;
    char *d = nullptr;
    {
        std::lock_guard<std::recursive_mutex> g(h->mu);
        int rc = ensure(h, &h->stage, &h->stage_bytes, total);
        if (rc) return rc;
        d = reinterpret_cast<char *>(h->stage);
        FCD_HIP(h, hipMemcpyAsync(d + o_l1, labels1, (size_t)n_pairs * stride1, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemcpyAsync(d + o_l2, labels2, (size_t)n_pairs * stride2, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemcpyAsync(d + o_p1, path1, (size_t)n_pairs * stride1 * 4, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemcpyAsync(d + o_p2, path2, (size_t)n_pairs * stride2 * 4, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemcpyAsync(d + o_n1, len1, (size_t)n_pairs * 4, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemcpyAsync(d + o_n2, len2, (size_t)n_pairs * 4, hipMemcpyHostToDevice, h->stream));
        if (T1) FCD_HIP(h, hipMemcpyAsync(d + o_t1, T1, (size_t)n_pairs * 8, hipMemcpyHostToDevice, h->stream));
        if (T2) FCD_HIP(h, hipMemcpyAsync(d + o_t2, T2, (size_t)n_pairs * 8, hipMemcpyHostToDevice, h->stream));
        FCD_HIP(h, hipMemsetAsync(d + o_env, 0, (size_t)n_pairs * env_stride * 16, h->stream));
    }
    int rc = fcd_duplex_envelope_dev(
        h, n_pairs, reinterpret_cast<uint8_t *>(d + o_l1), reinterpret_cast<uint32_t *>(d + o_p1),
        reinterpret_cast<uint32_t *>(d + o_n1), stride1, T1 ? reinterpret_cast<int64_t *>(d + o_t1) : nullptr, T1cap,
        reinterpret_cast<uint8_t *>(d + o_l2), reinterpret_cast<uint32_t *>(d + o_p2),
        reinterpret_cast<uint32_t *>(d + o_n2), stride2, T2 ? reinterpret_cast<int64_t *>(d + o_t2) : nullptr, T2cap,
        band, reinterpret_cast<uint64_t *>(d + o_env), env_stride);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    FCD_HIP(h, hipMemcpyAsync(envelope, d + o_env, (size_t)n_pairs * env_stride * 16, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    return FCD_OK;
}

int fcd_logspace_probe_dev(fcd_handle *h, const float *a, const float *b, float *out_add,
                           float *out_ln, int64_t n, int logadd_mode) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (n < 0 || (n > 0 && (!a || !b || !out_add || !out_ln))) return fail(h, FCD_E_INVALID, "null array");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_logspace_probe(a, b, out_add, out_ln, n, logadd_mode, h->stream));
    return FCD_OK;
}

int fcd_debug_glibc235_dev(fcd_handle *h, int which, const float *x, float *y, int64_t n) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (which < 0 || which > 2 || n < 0 || (n > 0 && (!x || !y))) return fail(h, FCD_E_INVALID, "bad argument");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_glibc235_apply(which, x, y, n, h->stream));
    return FCD_OK;
}

int fcd_logadd_sweep_dev(fcd_handle *h, int which, uint32_t first_bits, uint32_t last_bits, uint64_t *counts) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if ((which != 0 && which != 1) || !counts || first_bits > last_bits) return fail(h, FCD_E_INVALID, "bad argument");
    FCD_DEVICE(h);
    FCD_HIP(h, hipMemsetAsync(counts, 0, 24, h->stream));
    FCD_HIP(h, launch_logadd_sweep(which, first_bits, last_bits, reinterpret_cast<unsigned long long *>(counts), h->stream));
    return FCD_OK;
}

int fcd_logadd_latency_probe_dev(fcd_handle *h, int n_chain, int logadd_mode, uint64_t *cycles, float *sink) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (n_chain < 1 || !cycles || !sink) return fail(h, FCD_E_INVALID, "bad argument");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_logadd_chain(n_chain, logadd_mode, cycles, sink, h->stream));
    return FCD_OK;
}

// ---- compact wire format of a shard's results (pack.hip) ----
int64_t fcd_packed_result_bytes(int64_t n_reads, int64_t total_labels, int path_bytes) {
    if (n_reads < 0 || total_labels < 0 || (path_bytes != 2 && path_bytes != 4)) return -1;
    const int64_t b = 16 + 8 * n_reads + ((total_labels + 3) & ~3ll) + total_labels * path_bytes;
    return (b + 15) & ~15ll;
}

int fcd_result_offsets_dev(fcd_handle *h, const uint32_t *out_len, int64_t n_reads, int64_t out_stride,
                           uint64_t *offsets) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (n_reads < 0 || !offsets || (n_reads > 0 && !out_len) || out_stride < 0) return fail(h, FCD_E_INVALID, "bad argument");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_result_offsets(out_len, n_reads, out_stride, offsets, h->stream));
    return FCD_OK;
}

int fcd_pack_results_dev(fcd_handle *h, const fcd_result *res, int64_t n_reads, int path_bytes,
                         const uint64_t *offsets, uint8_t *buf) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (!res || n_reads < 0 || !offsets || !buf || (path_bytes != 2 && path_bytes != 4))
        return fail(h, FCD_E_INVALID, "bad argument");
    if (n_reads > 0 && (!res->labels || !res->path || !res->out_len)) return fail(h, FCD_E_INVALID, "null labels/path/out_len");
    FCD_DEVICE(h);
    if (n_reads == 0) {
        FCD_HIP(h, hipMemsetAsync(buf, 0, 16, h->stream));
        return FCD_OK;
    }
    ResultDesc wire = to_desc(res);
    wire.qual = nullptr;  // the wire format carries labels, path, out_len, status
    FCD_HIP(h, launch_pack(wire, n_reads, path_bytes, offsets, buf, h->stream));
    return FCD_OK;
}

int fcd_unpack_results_dev(fcd_handle *h, const uint8_t *buf, int64_t n_reads, uint64_t *offsets,
                           const fcd_result *out) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (!buf || n_reads < 0 || !offsets || !out) return fail(h, FCD_E_INVALID, "bad argument");
    if (n_reads == 0) return FCD_OK;
    if (!out->labels || !out->out_len) return fail(h, FCD_E_INVALID, "null labels/out_len");
    FCD_DEVICE(h);
    // the buffer's own out_len array (offset 16) gives the read offsets
    FCD_HIP(h, launch_result_offsets(reinterpret_cast<const uint32_t *>(buf + 16), n_reads, out->out_stride, offsets,
                                     h->stream));
    FCD_HIP(h, launch_unpack(buf, n_reads, offsets, to_desc(out), h->stream));
    return FCD_OK;
}

int fcd_unpack_gathered_dev(fcd_handle *h, const uint8_t *gathered, int64_t stride, int world, const int64_t *first,
                            int64_t n_total, uint64_t *offsets, const fcd_result *out, int32_t *bad) {
    if (!h) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (!gathered || stride < 16 || world < 1 || !first || n_total < 0 || !offsets || !out || !bad)
        return fail(h, FCD_E_INVALID, "bad argument");
    if (n_total == 0) return FCD_OK;
    if (!out->labels || !out->out_len) return fail(h, FCD_E_INVALID, "null labels/out_len");
    FCD_DEVICE(h);
    FCD_HIP(h, launch_unpack_gathered(gathered, stride, world, first, n_total, offsets, to_desc(out), bad, h->stream));
    return FCD_OK;
}

}  // extern "C"

// ---- *_host: stage host buffers through device memory, run the *_dev path, copy back -------
namespace fcd {

// Checks shared by every host-side entry of the four 1D searches.
int host_check(fcd_handle *h, const fcd_batch *in, const fcd_result *out, const HostCall &c) {
    const bool crf = c.op == HostOp::CrfBeam || c.op == HostOp::CrfGreedy;
    int rc = check_batch(h, in, crf);
    if (rc) return rc;
    rc = check_result(h, in, out, c.op != HostOp::Viterbi);
    if (rc) return rc;
    if (in->stride_read < 0 || in->stride_t < 0 || in->stride_n < 0 || (crf && in->stride_s < 0))
        return fail(h, FCD_E_UNSUPPORTED, "negative strides: pass a contiguous copy");
    if (crf && (!c.init || c.n_init < 1)) return fail(h, FCD_E_INVALID, "init_state missing");
    return FCD_OK;
}

// Lays the call out in h->stage (inputs first, then the fixed-stride result arrays) and uploads the inputs on
// h->stream.  `shape` says which result arrays are wanted (non-null members) and their stride; *din / *dout
// receive the device-side batch and result.  Small calls (the per-read drop-in functions: a few hundred KB) go
// through a page-locked mirror of the staging area when allow_mirror: ONE DMA in (and ONE out in host_download)
// instead of a runtime-staged copy per array from pageable memory; large batches are copied straight from / to
// the caller's arrays (an extra host pass would cost more).
int host_upload(fcd_handle *h, const fcd_batch *in, const fcd_result *shape, const HostCall &c, bool allow_mirror,
                HostStage *st, fcd_batch *din, fcd_result *dout) {
    const bool crf = c.op == HostOp::CrfBeam || c.op == HostOp::CrfGreedy;
    const int64_t B = in->n_reads;
    st->B = B;
    st->n_in = (size_t)span_elems(in, crf);
    st->n_out = (size_t)B * (size_t)shape->out_stride;
    st->n_init = crf ? (size_t)((B - 1) * c.init_stride + c.n_init) : 0;
    size_t used = 0;
    auto reserve = [&](size_t bytes) {
        size_t off = (used + 255) & ~(size_t)255;
        used = off + std::max<size_t>(bytes, 4);
        return off;
    };
    const size_t esz = in->dtype == FCD_DTYPE_F32 ? 4 : 2;  // bytes per posterior element
    st->o_in = reserve(st->n_in * esz);
    st->o_len = reserve(in->lengths ? (size_t)B * 8 : 0);
    st->o_init = reserve(st->n_init * 4);
    st->o_lab = reserve(st->n_out);
    st->o_path = reserve(shape->path ? st->n_out * 4 : 0);
    st->o_qual = reserve(shape->qual ? st->n_out * 4 : 0);
    st->o_olen = reserve((size_t)B * 4);
    st->o_stat = reserve((size_t)B * 4);
    st->want_amb = shape->ambiguous && (c.op == HostOp::Beam || c.op == HostOp::CrfBeam);
    st->o_amb = reserve(st->want_amb ? (size_t)B * 8 : 0);
    st->used = used;
    const bool mirror = allow_mirror && used <= ((size_t)4 << 20);
    st->mirror = mirror;

    int rc = ensure(h, &h->stage, &h->stage_bytes, used);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(h->stage);
    if (mirror) {
        if (h->pin_bytes < used) {
            if (h->pin) (void)hipHostFree(h->pin);
            h->pin = nullptr;
            h->pin_bytes = 0;
            const size_t want = std::max<size_t>(used + used / 2, (size_t)256 << 10);
            if (hipHostMalloc(&h->pin, want, hipHostMallocDefault) != hipSuccess) {
                h->pin = nullptr;
                return fail(h, FCD_E_NOMEM, "hipHostMalloc failed (staging mirror)");
            }
            h->pin_bytes = want;
        }
        char *pin = reinterpret_cast<char *>(h->pin);
        if (st->n_in) memcpy(pin + st->o_in, in->post, st->n_in * esz);
        if (in->lengths) memcpy(pin + st->o_len, in->lengths, (size_t)B * 8);
        if (crf) memcpy(pin + st->o_init, c.init, st->n_init * 4);
        FCD_HIP(h, hipMemcpyAsync(base, pin, st->o_lab, hipMemcpyHostToDevice, h->stream));  // inputs lie before o_lab
    } else {
        if (st->n_in)
            FCD_HIP(h, hipMemcpyAsync(base + st->o_in, in->post, st->n_in * esz, hipMemcpyHostToDevice, h->stream));
        if (in->lengths)
            FCD_HIP(h, hipMemcpyAsync(base + st->o_len, in->lengths, (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
        if (crf)
            FCD_HIP(h, hipMemcpyAsync(base + st->o_init, c.init, st->n_init * 4, hipMemcpyHostToDevice, h->stream));
    }
    *din = *in;
    din->post = reinterpret_cast<const float *>(base + st->o_in);
    din->lengths = in->lengths ? reinterpret_cast<const int64_t *>(base + st->o_len) : nullptr;
    dout->labels = reinterpret_cast<uint8_t *>(base + st->o_lab);
    dout->path = shape->path ? reinterpret_cast<uint32_t *>(base + st->o_path) : nullptr;
    dout->qual = shape->qual ? reinterpret_cast<float *>(base + st->o_qual) : nullptr;
    dout->out_len = reinterpret_cast<uint32_t *>(base + st->o_olen);
    dout->status = reinterpret_cast<int32_t *>(base + st->o_stat);
    dout->out_stride = shape->out_stride;
    dout->ambiguous = st->want_amb ? reinterpret_cast<uint32_t *>(base + st->o_amb) : nullptr;
    return FCD_OK;
}

// Runs the *_dev search of a staged call on h->stream.
int host_search(fcd_handle *h, const HostStage &st, const fcd_batch *din, const HostCall &c, const fcd_result *dout) {
    const float *dinit = reinterpret_cast<const float *>(reinterpret_cast<char *>(h->stage) + st.o_init);
    int rc = FCD_E_INVALID;
    switch (c.op) {
        case HostOp::Viterbi: return fcd_viterbi_search_dev(h, din, c.collapse, dout);
        case HostOp::Beam: rc = fcd_beam_search_dev(h, din, c.beam_size, c.thr, c.collapse, c.kernel, dout); break;
        case HostOp::CrfBeam:
            rc = fcd_crf_beam_search_dev_k(h, din, dinit, c.n_init, c.init_stride, c.beam_size, c.thr, c.kernel, dout);
            break;
        case HostOp::CrfGreedy: return fcd_crf_greedy_search_dev(h, din, dinit, c.n_init, c.init_stride, dout);
    }
    // (fcd_set_overlap: the search may sit on an internal stream; the download that follows is in the handle's stream)
    if (rc == FCD_OK && h->overlap_n >= 2) rc = overlap_join(h, h->stream);
    return rc;
}

// Copies the fixed-stride device result of a staged call into the caller's arrays and waits.
int host_download(fcd_handle *h, const HostStage &st, const fcd_result &dout, const fcd_result *out) {
    const size_t B = (size_t)st.B;
    if (st.mirror) {
        char *pin = reinterpret_cast<char *>(h->pin);
        char *base = reinterpret_cast<char *>(h->stage);
        FCD_HIP(h, hipMemcpyAsync(pin + st.o_lab, base + st.o_lab, st.used - st.o_lab, hipMemcpyDeviceToHost, h->stream));
        FCD_HIP(h, hipStreamSynchronize(h->stream));
        memcpy(out->labels, pin + st.o_lab, st.n_out);
        if (out->path) memcpy(out->path, pin + st.o_path, st.n_out * 4);
        if (out->qual) memcpy(out->qual, pin + st.o_qual, st.n_out * 4);
        memcpy(out->out_len, pin + st.o_olen, B * 4);
        if (out->status) memcpy(out->status, pin + st.o_stat, B * 4);
        if (st.want_amb) memcpy(out->ambiguous, pin + st.o_amb, B * 8);
        return FCD_OK;
    }
    FCD_HIP(h, hipMemcpyAsync(out->labels, dout.labels, st.n_out, hipMemcpyDeviceToHost, h->stream));
    if (out->path) FCD_HIP(h, hipMemcpyAsync(out->path, dout.path, st.n_out * 4, hipMemcpyDeviceToHost, h->stream));
    if (out->qual) FCD_HIP(h, hipMemcpyAsync(out->qual, dout.qual, st.n_out * 4, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipMemcpyAsync(out->out_len, dout.out_len, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (out->status) FCD_HIP(h, hipMemcpyAsync(out->status, dout.status, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (st.want_amb)
        FCD_HIP(h, hipMemcpyAsync(out->ambiguous, dout.ambiguous, B * 8, hipMemcpyDeviceToHost, h->stream));
    FCD_HIP(h, hipStreamSynchronize(h->stream));
    return FCD_OK;
}

}  // namespace fcd

namespace {

int run_host(fcd_handle *h, const fcd_batch *in, const fcd_result *out, const HostCall &c) {
    if (!h) return FCD_E_INVALID;
    // held from staging to copy-back: two threads sharing a handle are serialised (fcd.h), they cannot
    // interleave on the staging buffer
    std::lock_guard<std::recursive_mutex> whole_call(h->mu);
    int rc = host_check(h, in, out, c);
    if (rc) return rc;
    if (in->n_reads == 0) return FCD_OK;
    FCD_DEVICE(h);  // every HIP call of this entry point runs on the handle's device
    // Large batches: chunks on several internal lanes, upload || search || packed download (hostjob.hip)
    if (host_job_wanted(h, in, c)) return host_job_run_fixed(h, in, out, c);
    HostStage st;
    fcd_batch din{};
    fcd_result dout{};
    rc = host_upload(h, in, out, c, true, &st, &din, &dout);
    if (rc == FCD_OK) rc = host_search(h, st, &din, c, &dout);
    if (rc) return rc;
    return host_download(h, st, dout, out);
}

}  // namespace

extern "C" {

int fcd_viterbi_search_host(fcd_handle *h, const fcd_batch *in, int collapse_repeats,
                            const fcd_result *out) {
    HostCall c{HostOp::Viterbi};
    c.collapse = collapse_repeats;
    return run_host(h, in, out, c);
}

int fcd_beam_search_host(fcd_handle *h, const fcd_batch *in, int64_t beam_size,
                         float beam_cut_threshold, int collapse_repeats, int kernel,
                         const fcd_result *out) {
    HostCall c{HostOp::Beam};
    c.collapse = collapse_repeats;
    c.beam_size = beam_size;
    c.thr = beam_cut_threshold;
    c.kernel = kernel;
    return run_host(h, in, out, c);
}

int fcd_crf_beam_search_host(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                             int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                             const fcd_result *out) {
    return fcd_crf_beam_search_host_k(h, in, init, n_init, init_stride, beam_size, beam_cut_threshold,
                                      FCD_KERNEL_AUTO, out);
}

int fcd_crf_beam_search_host_k(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                               int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                               int kernel, const fcd_result *out) {
    HostCall c{HostOp::CrfBeam};
    c.beam_size = beam_size;
    c.thr = beam_cut_threshold;
    c.kernel = kernel;
    c.init = init;
    c.n_init = n_init;
    c.init_stride = init_stride;
    return run_host(h, in, out, c);
}

int fcd_crf_greedy_search_host(fcd_handle *h, const fcd_batch *in, const float *init,
                               int64_t n_init, int64_t init_stride, const fcd_result *out) {
    HostCall c{HostOp::CrfGreedy};
    c.init = init;
    c.n_init = n_init;
    c.init_stride = init_stride;
    return run_host(h, in, out, c);
}

// phred (src/search.rs:31-36), host side; log10f is libm's, as in the reference
uint32_t fcd_phred(float prob, float qscale, float qbias) {
    const float mx = 1e-4f;
    const float om = 1.0f - prob;
    const float p = (om < mx) ? mx : om;
    const float q = -10.0f * log10f(p) * qscale + qbias;
    const float rq = roundf(q);  // f32::round: half away from zero
    uint32_t u;                  // Rust `as u32`: saturating, NaN -> 0
    if (!(rq == rq) || rq <= 0.0f) u = 0;
    else if (rq >= 4294967296.0f) u = 4294967295u;
    else u = (uint32_t)rq;
    return u + 33u;
}

}  // extern "C"
