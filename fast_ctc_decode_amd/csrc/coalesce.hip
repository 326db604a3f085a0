// coalesce.hip -- a front door for PER-READ callers (host code only; layered on the public C ABI).
//
// The reference decodes one read per call (src/lib.rs:170-212 viterbi_search, :318-365 beam_search) and
// releases the GIL around the search (:199, :353) precisely so that callers -- basecallers -- can run it from
// many threads.  On a GPU a lone read is one wavefront: every such call pays a launch, two copies and a
// stream synchronisation for 1/2048th of the machine, and HIP multiplexes the callers' streams onto a handful
// of hardware queues, so concurrent single-read launches mostly run one after another.
//
// A coalescer turns concurrent calls into batches without changing the callers: the first thread to arrive
// becomes the LEADER, takes every compatible request that is pending (same search, alphabet size, beam size,
// threshold, collapse flag; the pair searches of src/duplex.rs too: one PAIR per call, same log-add flavour), decodes them with ONE batched launch (ragged lengths) on the coalescer's own
// handle and hands each caller its own result; the callers that arrive while a launch is in flight form the
// next batch.  A launch takes about as long for one read as for a thousand (a read is one wavefront), so the
// policy is "batch first": a leader takes everything that is pending, and before launching it waits briefly
// for the callers of the PREVIOUS batches to come back with their next read -- until as many requests are
// pending as recent batches held, for at most an eighth of the last launch's duration (<= 1 ms); a lone caller
// never waits (recent batches held one read).  An explicit max_wait_us > 0 instead makes every leader wait that
// long for company (or until max_batch requests are pending).  Up to kLanes leaders
// work at the same time, each with its own handle (stream, workspace) and pinned staging buffers, but only while
// fewer than kLanes reads are in flight: a handful of callers overlap their launches the way independent
// per-read calls would, many callers share big batches instead of fragmenting them.
// Results are those of the batched entry points, i.e. bit-identical to the per-read calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fcd.h"

namespace {

enum Kind { kBeam = 0, kViterbi = 1, kCrfBeam = 2, kCrfGreedy = 3, kDuplex = 4, kCrfDuplex = 5 };

struct Req {
    int kind;
    const fcd_batch *in;
    int64_t beam;
    float thr;
    int collapse;
    const fcd_result *out;
    const float *init = nullptr;  // CRF searches: the read's init_state (n_init entries, contiguous)
    int64_t n_init = 0;
    // the pair searches (duplex::beam_search, duplex::crf_beam_search): the second read, its init_state, the pair's
    // envelope (in->T rows of {lo, hi}), the LogSpace::add flavour
    const fcd_batch *in2 = nullptr;
    const float *init2 = nullptr;
    int64_t n_init2 = 0;
    const uint64_t *env = nullptr;
    int mode = 0;
    int rc = FCD_OK;
    bool done = false;
    std::string err;
    std::chrono::steady_clock::time_point arrived;
};

bool compatible(const Req &a, const Req &b) {
    return a.kind == b.kind && a.in->N == b.in->N && a.in->S == b.in->S && a.n_init == b.n_init && a.beam == b.beam &&
           a.collapse == b.collapse && std::memcmp(&a.thr, &b.thr, sizeof(float)) == 0 && a.n_init2 == b.n_init2 &&
           a.mode == b.mode;
}

thread_local std::string t_error;

constexpr int kLanes = 4;  // concurrent leaders (HIP multiplexes streams onto about this many hardware queues)

// a grow-only pinned host buffer (page-locked memory is copied by DMA straight from / to the caller's pages)
struct Pinned {
    void *p = nullptr;
    size_t cap = 0;
    void *need(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = bytes + bytes / 2 + 4096;
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
                p = nullptr;
                return nullptr;
            }
            cap = want;
        }
        return p;
    }
    ~Pinned() {
        if (p) (void)hipHostFree(p);
    }
};

// one leader's working set
struct Lane {
    fcd_handle *h = nullptr;
    bool busy = false;
    Pinned x, qual, labels, path, out_len, status, lengths, init;
    Pinned x2, lengths2, init2, env;  // the pair searches' second read and envelopes
};

}  // namespace

struct fcd_coalescer {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Req *> pending;
    Lane lanes[kLanes];
    int device = 0;
    int max_batch = 256;
    int max_wait_us = 0;
    int64_t n_calls = 0, n_launches = 0, largest = 0;
    int64_t in_flight = 0;     // reads in running batches
    int64_t recent = 1;        // decaying maximum of recent batch sizes: how much company is worth waiting for
    int64_t last_launch_us = 0;
};

namespace {

void run_batch_once(Lane &L, std::vector<Req *> &batch);

// one batched launch for `batch` (all compatible) on lane `L`; fills every request's outputs and rc.  A failure of
// the BATCH (staging memory, a shape limit decided by the longest read of the batch) is not every caller's
// failure: the requests are then decoded one by one, so that each caller gets exactly what a per-read call gives.
void run_batch(Lane &L, std::vector<Req *> &batch) {
    run_batch_once(L, batch);
    if (batch.size() > 1 && batch[0]->rc != FCD_OK) {
        for (Req *r : batch) {
            std::vector<Req *> one{r};
            run_batch_once(L, one);
        }
    }
}

// one read into its padded slot: dense rows are one copy, anything else goes element by element
void stage_read(const fcd_batch *b, bool crf, int64_t S, int64_t N, float *dst) {
    const int64_t row = S * N;
    const float *src = static_cast<const float *>(b->post);  // (the coalescer takes float32 reads only)
    const bool dense = b->stride_n == 1 && (crf ? (b->stride_s == N && b->stride_t == row) : b->stride_t == N);
    if (dense) {
        std::memcpy(dst, src, (size_t)(b->T * row) * sizeof(float));
        return;
    }
    for (int64_t t = 0; t < b->T; ++t)
        for (int64_t sidx = 0; sidx < S; ++sidx)
            for (int64_t j = 0; j < N; ++j)
                dst[(t * S + sidx) * N + j] = src[t * b->stride_t + (crf ? sidx * b->stride_s : 0) + j * b->stride_n];
}

// the pair searches: every request is one pair (two reads, an envelope of in->T rows); the batch is the ragged batch
// fcd_beam_search_duplex_host / fcd_crf_beam_search_duplex_host take
void run_pair_batch_once(Lane &L, std::vector<Req *> &batch) {
    const int64_t n = (int64_t)batch.size();
    const Req &first = *batch[0];
    const int64_t N = first.in->N;
    const bool crf = first.kind == kCrfDuplex;
    const int64_t S = crf ? first.in->S : 1, row = S * N;
    int64_t T1 = 1, T2 = 1;
    for (Req *r : batch) {
        T1 = std::max<int64_t>(T1, r->in->T);
        T2 = std::max<int64_t>(T2, r->in2->T);
    }
    float *x1 = static_cast<float *>(L.x.need((size_t)(n * T1 * row) * sizeof(float)));
    float *x2 = static_cast<float *>(L.x2.need((size_t)(n * T2 * row) * sizeof(float)));
    uint64_t *env = static_cast<uint64_t *>(L.env.need((size_t)(n * T1 * 2) * sizeof(uint64_t)));
    uint8_t *labels = static_cast<uint8_t *>(L.labels.need((size_t)(n * T1)));
    uint32_t *out_len = static_cast<uint32_t *>(L.out_len.need((size_t)n * sizeof(uint32_t)));
    int32_t *status = static_cast<int32_t *>(L.status.need((size_t)n * sizeof(int32_t)));
    int64_t *len1 = static_cast<int64_t *>(L.lengths.need((size_t)n * sizeof(int64_t)));
    int64_t *len2 = static_cast<int64_t *>(L.lengths2.need((size_t)n * sizeof(int64_t)));
    float *init1 = crf ? static_cast<float *>(L.init.need((size_t)(n * first.n_init) * sizeof(float))) : nullptr;
    float *init2 = crf ? static_cast<float *>(L.init2.need((size_t)(n * first.n_init2) * sizeof(float))) : nullptr;
    int rc = FCD_OK;
    std::string err;
    if (!x1 || !x2 || !env || !labels || !out_len || !status || !len1 || !len2 || (crf && (!init1 || !init2))) {
        rc = FCD_E_NOMEM;
        err = "coalescer: cannot allocate pinned staging memory";
    }
    if (rc == FCD_OK) {
        std::memset(env, 0, (size_t)(n * T1 * 2) * sizeof(uint64_t));
        for (int64_t i = 0; i < n; ++i) {
            const Req *r = batch[i];
            stage_read(r->in, crf, S, N, x1 + i * T1 * row);
            stage_read(r->in2, crf, S, N, x2 + i * T2 * row);
            len1[i] = r->in->T;
            len2[i] = r->in2->T;
            std::memcpy(env + i * T1 * 2, r->env, (size_t)(r->in->T * 2) * sizeof(uint64_t));
            if (crf) {
                std::memcpy(init1 + i * first.n_init, r->init, (size_t)first.n_init * sizeof(float));
                std::memcpy(init2 + i * first.n_init2, r->init2, (size_t)first.n_init2 * sizeof(float));
            }
        }
        fcd_batch a{}, b{};
        a.post = x1;
        a.n_reads = n;
        a.T = T1;
        a.S = S;
        a.N = N;
        a.stride_read = T1 * row;
        a.stride_t = row;
        a.stride_s = crf ? N : 0;
        a.stride_n = 1;
        a.lengths = len1;
        b = a;
        b.post = x2;
        b.T = T2;
        b.stride_read = T2 * row;
        b.lengths = len2;
        fcd_result out{};
        out.labels = labels;
        out.out_len = out_len;
        out.status = status;
        out.out_stride = T1;
        if (crf)
            rc = fcd_crf_beam_search_duplex_host(L.h, &a, init1, first.n_init, first.n_init, &b, init2, first.n_init2,
                                                 first.n_init2, env, T1, first.beam, first.thr, first.mode, &out);
        else
            rc = fcd_beam_search_duplex_host(L.h, &a, &b, env, T1, first.beam, first.thr, first.collapse, first.mode, &out);
        if (rc != FCD_OK) err = fcd_last_error(L.h);
    }
    for (int64_t i = 0; i < n; ++i) {
        Req *r = batch[i];
        r->rc = rc;
        r->err = err;
        if (rc != FCD_OK) continue;
        const fcd_result *o = r->out;
        const size_t len = std::min<size_t>(out_len[i], (size_t)std::max<int64_t>(o->out_stride, 0));
        if (o->out_len) *o->out_len = out_len[i];
        *o->status = status[i];
        std::memcpy(o->labels, labels + i * T1, len);
    }
}

void run_batch_once(Lane &L, std::vector<Req *> &batch) {
    if (batch[0]->kind == kDuplex || batch[0]->kind == kCrfDuplex) {
        run_pair_batch_once(L, batch);
        return;
    }
    const int64_t n = (int64_t)batch.size();
    const Req &first = *batch[0];
    const int64_t N = first.in->N;
    const bool crf = first.kind == kCrfBeam || first.kind == kCrfGreedy;
    const int64_t S = crf ? first.in->S : 1, row = S * N, n_init = first.n_init;
    int64_t Tmax = 1;
    bool want_path = false, want_qual = false;
    for (Req *r : batch) {
        Tmax = std::max<int64_t>(Tmax, r->in->T);
        want_path = want_path || r->out->path;
        want_qual = want_qual || r->out->qual;
    }
    float *x = static_cast<float *>(L.x.need((size_t)(n * Tmax * row) * sizeof(float)));
    uint8_t *labels = static_cast<uint8_t *>(L.labels.need((size_t)(n * Tmax)));
    uint32_t *out_len = static_cast<uint32_t *>(L.out_len.need((size_t)n * sizeof(uint32_t)));
    int32_t *status = static_cast<int32_t *>(L.status.need((size_t)n * sizeof(int32_t)));
    int64_t *lengths = static_cast<int64_t *>(L.lengths.need((size_t)n * sizeof(int64_t)));
    uint32_t *path = want_path ? static_cast<uint32_t *>(L.path.need((size_t)(n * Tmax) * sizeof(uint32_t))) : nullptr;
    float *qual = want_qual ? static_cast<float *>(L.qual.need((size_t)(n * Tmax) * sizeof(float))) : nullptr;
    float *init = crf ? static_cast<float *>(L.init.need((size_t)(n * n_init) * sizeof(float))) : nullptr;
    int rc = FCD_OK;
    std::string err;
    if (!x || !labels || !out_len || !status || !lengths || (want_path && !path) || (want_qual && !qual) || (crf && !init)) {
        rc = FCD_E_NOMEM;
        err = "coalescer: cannot allocate pinned staging memory";
    }
    if (rc == FCD_OK) {
        for (int64_t i = 0; i < n; ++i) {
            const fcd_batch *b = batch[i]->in;
            lengths[i] = b->T;
            stage_read(b, crf, S, N, x + i * Tmax * row);
            if (crf) std::memcpy(init + i * n_init, batch[i]->init, (size_t)n_init * sizeof(float));
        }
        fcd_batch in{};
        in.post = x;
        in.n_reads = n;
        in.T = Tmax;
        in.S = S;
        in.N = N;
        in.stride_read = Tmax * row;
        in.stride_t = row;
        in.stride_s = crf ? N : 0;
        in.stride_n = 1;
        in.lengths = lengths;
        fcd_result out{};
        out.labels = labels;
        out.path = path;
        out.qual = qual;
        out.out_len = out_len;
        out.status = status;
        out.out_stride = Tmax;
        if (first.kind == kBeam)
            rc = fcd_beam_search_host(L.h, &in, first.beam, first.thr, first.collapse, FCD_KERNEL_AUTO, &out);
        else if (first.kind == kViterbi)
            rc = fcd_viterbi_search_host(L.h, &in, first.collapse, &out);
        else if (first.kind == kCrfBeam)
            rc = fcd_crf_beam_search_host(L.h, &in, init, n_init, n_init, first.beam, first.thr, &out);
        else
            rc = fcd_crf_greedy_search_host(L.h, &in, init, n_init, n_init, &out);
        if (rc != FCD_OK) err = fcd_last_error(L.h);
    }
    for (int64_t i = 0; i < n; ++i) {
        Req *r = batch[i];
        r->rc = rc;
        r->err = err;
        if (rc != FCD_OK) continue;
        const fcd_result *o = r->out;
        const size_t len = std::min<size_t>(out_len[i], (size_t)std::max<int64_t>(o->out_stride, 0));
        if (o->out_len) *o->out_len = out_len[i];
        if (o->status) *o->status = status[i];
        std::memcpy(o->labels, labels + i * Tmax, len);
        if (o->path) std::memcpy(o->path, path + i * Tmax, len * sizeof(uint32_t));
        if (o->qual) std::memcpy(o->qual, qual + i * Tmax, len * sizeof(float));
    }
}

int submit(fcd_coalescer *c, Req &req) {
    if (!c || !req.in || !req.out || !req.in->post || !req.out->labels) {
        t_error = "coalescer: null argument";
        return FCD_E_INVALID;
    }
    const bool pair = req.kind == kDuplex || req.kind == kCrfDuplex;
    const bool crf = req.kind == kCrfBeam || req.kind == kCrfGreedy || req.kind == kCrfDuplex;
    if (pair) {
        const fcd_batch *b = req.in2;
        if (!b || !b->post || !req.env || (crf && (!req.init2 || req.n_init2 < 1))) {
            t_error = "coalescer: a pair search needs both reads and the pair's envelope (CRF: both init_states)";
            return FCD_E_INVALID;
        }
        if (b->dtype != FCD_DTYPE_F32) {
            t_error = "coalescer: float32 reads only (the per-read surface is the reference's, which takes float32)";
            return FCD_E_UNSUPPORTED;
        }
        if (b->n_reads != 1 || b->T < 0 || b->N != req.in->N || b->stride_t < 0 || b->stride_n < 0 || b->lengths ||
            (crf ? (b->S != req.in->S || b->stride_s < 0) : b->S > 1)) {
            t_error = "coalescer: the second read must be one matrix of the first read's inner shape, with non-negative strides";
            return FCD_E_INVALID;
        }
    }
    if (req.kind != kViterbi && !req.out->status) {  // a per-read FCD_ST_* outcome must have somewhere to go
        t_error = "coalescer: the beam and CRF searches need out->status";
        return FCD_E_INVALID;
    }
    if (req.in->dtype != FCD_DTYPE_F32) {
        t_error = "coalescer: float32 reads only (the per-read surface is the reference's, which takes float32)";
        return FCD_E_UNSUPPORTED;
    }
    if (req.in->n_reads != 1 || req.in->T < 0 || req.in->N < 1 || req.in->stride_t < 0 || req.in->stride_n < 0 ||
        req.out->out_stride < req.in->T || req.in->lengths ||
        (crf ? (req.in->S < 1 || req.in->stride_s < 0 || !req.init || req.n_init < 1) : req.in->S > 1)) {
        t_error = crf ? "coalescer: expects exactly one (T, S, N) read with non-negative strides, an init_state and out_stride >= T"
                      : "coalescer: expects exactly one (T, N) read with non-negative strides and out_stride >= T";
        return FCD_E_INVALID;
    }
    req.arrived = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(c->mu);
    c->pending.push_back(&req);
    ++c->n_calls;
    c->cv.notify_all();  // a leader waiting for company looks again
    while (!req.done) {
        int li = -1;
        for (int j = 0; j < kLanes && li < 0; ++j)
            if (!c->lanes[j].busy) li = j;
        // my request may already be in another leader's batch; lead only while something is pending, and side
        // by side with other leaders only while little is in flight (batch first, see the header)
        if (li < 0 || c->pending.empty() || (c->in_flight > 0 && c->in_flight >= kLanes)) {
            c->cv.wait(lk);
            continue;
        }
        // become a leader: serve the oldest pending request's group (possibly not my own: then go round again)
        Lane &L = c->lanes[li];
        L.busy = true;
        {
            const int64_t budget_us = c->max_wait_us > 0 ? c->max_wait_us
                                                         : std::min<int64_t>(c->last_launch_us / 8, 1000);
            // an explicit max_wait_us asks for company outright; the adaptive default only waits for as many
            // requests as recent batches held
            const int64_t target = c->max_wait_us > 0 ? c->max_batch : std::min<int64_t>(c->recent, c->max_batch);
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(budget_us);
            while (!c->pending.empty() && (int64_t)c->pending.size() < target &&
                   std::chrono::steady_clock::now() < deadline)
                c->cv.wait_until(lk, deadline);
        }
        std::vector<Req *> batch;
        if (!c->pending.empty()) {
            batch.push_back(c->pending.front());
            c->pending.pop_front();
            for (auto it = c->pending.begin(); it != c->pending.end() && (int)batch.size() < c->max_batch;) {
                if (compatible(*batch[0], **it)) {
                    batch.push_back(*it);
                    it = c->pending.erase(it);
                } else {
                    ++it;
                }
            }
        }
        if (!batch.empty()) {
            const int64_t nb = (int64_t)batch.size();
            ++c->n_launches;
            c->largest = std::max(c->largest, nb);
            c->recent = std::max(nb, c->recent - (c->recent + 3) / 4);
            c->in_flight += nb;
            int rc = FCD_OK;
            if (!L.h) rc = fcd_create(c->device, &L.h);  // lanes beyond the first get their handle on first use
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            if (rc == FCD_OK) {
                run_batch(L, batch);
            } else {
                for (Req *r : batch) {
                    r->rc = rc;
                    r->err = "coalescer: fcd_create failed";
                }
            }
            const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            lk.lock();
            c->last_launch_us = us;
            c->in_flight -= nb;
            for (Req *r : batch) r->done = true;
        }
        L.busy = false;
        c->cv.notify_all();
    }
    if (req.rc != FCD_OK) t_error = req.err;
    return req.rc;
}

}  // namespace

extern "C" {

int fcd_coalescer_create(int device, int max_batch, int max_wait_us, fcd_coalescer **out) {
    if (!out || max_batch < 1 || max_wait_us < 0) return FCD_E_INVALID;
    *out = nullptr;
    fcd_handle *h = nullptr;
    const int rc = fcd_create(device, &h);
    if (rc != FCD_OK) return rc;
    fcd_coalescer *c = new fcd_coalescer();
    c->lanes[0].h = h;
    c->device = device;
    c->max_batch = max_batch;
    c->max_wait_us = max_wait_us;
    *out = c;
    return FCD_OK;
}

int fcd_coalescer_destroy(fcd_coalescer *c) {
    if (!c) return FCD_E_INVALID;
    {
        std::lock_guard<std::mutex> g(c->mu);
        bool busy = !c->pending.empty();
        for (const Lane &L : c->lanes) busy = busy || L.busy;
        if (busy) return FCD_E_INVALID;  // calls still in flight
    }
    for (Lane &L : c->lanes)
        if (L.h) fcd_destroy(L.h);
    delete c;
    return FCD_OK;
}

int fcd_coalescer_beam_search(fcd_coalescer *c, const fcd_batch *read, int64_t beam_size, float beam_cut_threshold,
                              int collapse_repeats, const fcd_result *out) {
    Req r{kBeam, read, beam_size, beam_cut_threshold, collapse_repeats ? 1 : 0, out};
    return submit(c, r);
}

int fcd_coalescer_viterbi_search(fcd_coalescer *c, const fcd_batch *read, int collapse_repeats, const fcd_result *out) {
    Req r{kViterbi, read, 0, 0.0f, collapse_repeats ? 1 : 0, out};
    return submit(c, r);
}

int fcd_coalescer_crf_beam_search(fcd_coalescer *c, const fcd_batch *read, const float *init, int64_t n_init,
                                  int64_t beam_size, float beam_cut_threshold, const fcd_result *out) {
    Req r{kCrfBeam, read, beam_size, beam_cut_threshold, 0, out};
    r.init = init;
    r.n_init = n_init;
    return submit(c, r);
}

int fcd_coalescer_crf_greedy_search(fcd_coalescer *c, const fcd_batch *read, const float *init, int64_t n_init,
                                    const fcd_result *out) {
    Req r{kCrfGreedy, read, 0, 0.0f, 0, out};
    r.init = init;
    r.n_init = n_init;
    return submit(c, r);
}

int fcd_coalescer_beam_search_duplex(fcd_coalescer *c, const fcd_batch *read1, const fcd_batch *read2,
                                     const uint64_t *envelope, int64_t beam_size, float beam_cut_threshold,
                                     int collapse_repeats, int logadd_mode, const fcd_result *out) {
    Req r{kDuplex, read1, beam_size, beam_cut_threshold, collapse_repeats ? 1 : 0, out};
    r.in2 = read2;
    r.env = envelope;
    r.mode = logadd_mode;
    return submit(c, r);
}

int fcd_coalescer_crf_beam_search_duplex(fcd_coalescer *c, const fcd_batch *read1, const float *init1, int64_t n_init1,
                                         const fcd_batch *read2, const float *init2, int64_t n_init2,
                                         const uint64_t *envelope, int64_t beam_size, float beam_cut_threshold,
                                         int logadd_mode, const fcd_result *out) {
    Req r{kCrfDuplex, read1, beam_size, beam_cut_threshold, 0, out};
    r.init = init1;
    r.n_init = n_init1;
    r.in2 = read2;
    r.init2 = init2;
    r.n_init2 = n_init2;
    r.env = envelope;
    r.mode = logadd_mode;
    return submit(c, r);
}

int fcd_coalescer_stats(fcd_coalescer *c, int64_t *n_calls, int64_t *n_launches, int64_t *largest_batch) {
    if (!c) return FCD_E_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_calls) *n_calls = c->n_calls;
    if (n_launches) *n_launches = c->n_launches;
    if (largest_batch) *largest_batch = c->largest;
    return FCD_OK;
}

const char *fcd_coalescer_last_error(void) { return t_error.c_str(); }

}  // extern "C"
