// hostjob.hip -- large HOST batches of the four 1D searches as a pipeline of chunks (host code only).
//
// The reference's callers hold posteriors in host numpy arrays (src/lib.rs:182,325 take &PyArray2<f32>), so
// for them the drop-in path is: PCIe up, search, PCIe down, Python objects.  A search launch takes about as
// long for 512 reads as for 4096 (a read is one wavefront walking T dependent steps), the upload of BASELINE
// config 2 (4096 x 80 KB) takes longer than its search, and only ~48 % of every fixed-stride result row is
// used.  So a batch is cut into chunks that travel through a few LANES -- internal sub-handles with their own
// HIP stream, staging area, tree arena and page-locked result buffer:
//
//   lane thread     chunk c = lane, lane + L, ...: upload (blocking from pageable memory; the lanes take turns in
//                   chunk order, so chunk c is on the GPU before chunk c+1 starts to travel) | search | offsets +
//                   pack | download of the chunk's header (out_len, status) into page-locked memory | event
//   caller thread   fcd_job_next: wait for the event, download exactly the used label / path / quality bytes,
//                   hand out a view (fcd_chunk) of the lane's page-locked buffer
//
// (one thread per lane, not one for all: a wide-beam search that sizes its arena in two passes waits for its
// own stream between them, and must not hold up the other lanes)
//
// so the upload of chunk c+1 overlaps the searches of chunks <= c (which run side by side on the GPU: each is
// a fraction of a wavefront per SIMD), downloads carry only used prefixes (u16 times when T < 65536), and the
// caller builds its objects for chunk c while later chunks are still in flight.  A lane's buffers are reused
// by chunk c + L once the caller has moved on from chunk c (fcd_job_next / fcd_job_end release it).
// fcd_*_host on large batches = the same job with the views expanded into the caller's fixed-stride arrays.
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <thread>

#include "fcd_internal.h"

using namespace fcd;

struct fcd_host_lane {
    fcd_handle *h = nullptr;  // sub-handle: stream, staging area, tree arena
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_header = nullptr;
    void *pack_dev = nullptr;  // wire buffer (pack.hip layout) + read offsets
    size_t pack_dev_bytes = 0;
    void *pack_pin = nullptr;  // page-locked mirror the views point into
    size_t pack_pin_bytes = 0;
    void *in_pin = nullptr;    // page-locked gather buffer of a chunk whose reads arrive as separate arrays (fcd_*_host_ptrs_begin)
    size_t in_pin_bytes = 0;
};

struct fcd_job {
    fcd_handle *h = nullptr;
    fcd_batch in{};
    HostCall call{HostOp::Viterbi};
    const void *const *read_ptrs = nullptr;  // one base pointer per read instead of in.post (contiguous (T_r, N) matrices)
    const int64_t *read_rows = nullptr;      // their row counts (host array, as long as the job lives)
    int want = 0;
    int64_t chunk = 0;
    int n_chunks = 0, n_lanes = 0;
    int path_bytes = 2;
    std::vector<std::thread> workers;  // one per lane
    int upload_turn = 0;               // the chunk whose upload may start
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> state;  // per chunk: 0 not issued yet, 1 issued, 2 failed
    int handed = 0;          // chunks handed to the caller so far
    int released = 0;        // chunks the caller has moved on from
    bool cancel = false;
    int rc = FCD_OK;
    std::string err;
    std::vector<uint64_t> offsets;  // of the chunk being viewed
};

namespace {

int env_int(const char *name, int fallback) {
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}

size_t header_bytes(int64_t n) { return 16 + 8 * (size_t)n; }

// bytes after the header: labels (padded to 4) | path | pad | qual
size_t payload_bytes(uint64_t total, int path_bytes, bool has_path, bool has_qual) {
    size_t b = (size_t)((total + 3) & ~(uint64_t)3);
    if (has_path) b += (size_t)total * (size_t)path_bytes;
    b = (b + 3) & ~(size_t)3;
    if (has_qual) b += (size_t)total * 4;
    return b;
}

int lane_fail(fcd_job *j, int rc, const std::string &msg) {
    std::lock_guard<std::mutex> lk(j->mu);
    if (j->rc == FCD_OK) {  // the first failure is the one reported
        j->rc = rc;
        j->err = msg;
    }
    return rc;
}

// one chunk on its lane: everything is enqueued on the lane's stream; returns once the upload has been issued
int issue_chunk(fcd_job *j, int c) {
    fcd_host_lane *L = j->h->lanes[c % j->n_lanes];
    fcd_handle *lh = L->h;
    std::lock_guard<std::recursive_mutex> g(lh->mu);
    const int64_t b0 = (int64_t)c * j->chunk;
    const int64_t n = std::min<int64_t>(j->chunk, j->in.n_reads - b0);
    const bool crf = j->call.op == HostOp::CrfBeam || j->call.op == HostOp::CrfGreedy;
    const bool has_path = (j->want & FCD_JOB_PATH) != 0, has_qual = (j->want & FCD_JOB_QUAL) != 0;
    const bool has_amb = (j->want & FCD_JOB_AMBIGUOUS) != 0;
    fcd_batch sub = j->in;
    const size_t esz = j->in.dtype == FCD_DTYPE_F32 ? 4 : 2;
    sub.n_reads = n;
    sub.lengths = j->in.lengths ? j->in.lengths + b0 : nullptr;
    if (j->read_ptrs) {
        // Reads that live in separate host arrays (a Python list of ragged matrices: what the reference's callers hold,
        // src/lib.rs:325,352 take them one by one) are gathered into the lane's page-locked buffer, every lane its own
        // chunk at the same time, and leave as ONE DMA -- no padded copy of the batch on the caller's side.
        const size_t row_bytes = (size_t)j->in.N * esz, slot = (size_t)j->in.T * row_bytes;
        const size_t need = std::max<size_t>((size_t)n * slot, 16);
        if (L->in_pin_bytes < need) {
            if (L->in_pin) (void)hipHostFree(L->in_pin);
            L->in_pin = nullptr;
            L->in_pin_bytes = 0;
            if (hipHostMalloc(&L->in_pin, need, hipHostMallocDefault) != hipSuccess)
                return lane_fail(j, FCD_E_NOMEM, "hipHostMalloc failed (gather buffer)");
            L->in_pin_bytes = need;
        }
        char *dst = reinterpret_cast<char *>(L->in_pin);
        auto gather = [&](int64_t i0, int64_t i1) {
            for (int64_t i = i0; i < i1; ++i) {
                const int64_t rows = std::min<int64_t>(std::max<int64_t>(j->read_rows[b0 + i], 0), j->in.T);
                if (rows > 0) memcpy(dst + (size_t)i * slot, j->read_ptrs[b0 + i], (size_t)rows * row_bytes);
            }
        };
        // one core copies ~8 GB/s, PCIe takes 55: a large chunk is gathered by a few threads (FCD_HOST_GATHER_THREADS;
        // measured on 4096 ragged reads, three lanes: 2 -> 14.6 ms, 4 -> 15.2, 6..12 -> 17-18: the lanes gather at the
        // same time, and the boxes this runs on sustain about a dozen busy cores)
        const int kmax = std::max(1, std::min(env_int("FCD_HOST_GATHER_THREADS", 2), 16));
        const int k = (int)std::min<int64_t>(kmax, std::max<int64_t>(1, (int64_t)((size_t)n * slot >> 23)));  // >= 8 MB each
        if (k <= 1) {
            gather(0, n);
        } else {
            std::vector<std::thread> helpers;
            for (int t = 1; t < k; ++t) helpers.emplace_back(gather, n * t / k, n * (t + 1) / k);
            gather(0, n / k);
            for (std::thread &t : helpers) t.join();
        }
        sub.post = L->in_pin;
        sub.lengths = j->read_rows + b0;
    } else {
        sub.post = reinterpret_cast<const char *>(j->in.post) + b0 * j->in.stride_read * (int64_t)esz;
    }
    HostCall call = j->call;
    if (crf) call.init = j->call.init + b0 * j->call.init_stride;
    const int64_t W = std::max<int64_t>(j->in.T, 1);
    // which result arrays the search should produce: any non-null pointer asks for the array
    fcd_result shape{};
    uint8_t dummy = 0;
    shape.labels = &dummy;
    shape.path = has_path ? reinterpret_cast<uint32_t *>(&dummy) : nullptr;
    shape.qual = has_qual ? reinterpret_cast<float *>(&dummy) : nullptr;
    shape.out_len = reinterpret_cast<uint32_t *>(&dummy);
    shape.status = reinterpret_cast<int32_t *>(&dummy);
    shape.ambiguous = has_amb ? reinterpret_cast<uint32_t *>(&dummy) : nullptr;
    shape.out_stride = W;
    HostStage st;
    fcd_batch din{};
    fcd_result dout{};
    {   // uploads go one at a time, in chunk order
        std::unique_lock<std::mutex> lk(j->mu);
        j->cv.wait(lk, [&] { return j->cancel || j->upload_turn == c; });
        if (j->cancel) return FCD_E_INVALID;  // the job was abandoned (or another lane failed)
    }
    int rc = host_upload(lh, &sub, &shape, call, false, &st, &din, &dout);
    {
        std::lock_guard<std::mutex> lk(j->mu);
        j->upload_turn = c + 1;
        j->cv.notify_all();
    }
    if (rc == FCD_OK) rc = host_search(lh, st, &din, call, &dout);
    if (rc) return lane_fail(j, rc, lh->err);

    const size_t worst = header_bytes(n) + payload_bytes((uint64_t)n * (uint64_t)W, j->path_bytes, has_path, has_qual);
    const size_t amb_off = (worst + 15) & ~(size_t)15;
    const size_t offs_off = (amb_off + (has_amb ? (size_t)n * 8 : 0) + 15) & ~(size_t)15;
    const size_t dev_need = offs_off + (size_t)(n + 1) * 8;
    if (L->pack_dev_bytes < dev_need) {
        if (L->pack_dev) {
            (void)hipStreamSynchronize(lh->stream);
            (void)hipStreamSynchronize(L->copy_stream);
            (void)hipFree(L->pack_dev);
            L->pack_dev = nullptr;
            L->pack_dev_bytes = 0;
        }
        if (hipMalloc(&L->pack_dev, dev_need) != hipSuccess) return lane_fail(j, FCD_E_NOMEM, "hipMalloc failed (result chunk)");
        L->pack_dev_bytes = dev_need;
    }
    if (L->pack_pin_bytes < offs_off) {
        if (L->pack_pin) (void)hipHostFree(L->pack_pin);
        L->pack_pin = nullptr;
        L->pack_pin_bytes = 0;
        if (hipHostMalloc(&L->pack_pin, offs_off, hipHostMallocDefault) != hipSuccess)
            return lane_fail(j, FCD_E_NOMEM, "hipHostMalloc failed (result chunk)");
        L->pack_pin_bytes = offs_off;
    }
    char *dev = reinterpret_cast<char *>(L->pack_dev);
    char *pin = reinterpret_cast<char *>(L->pack_pin);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(dev + offs_off);
    ResultDesc rd{dout.labels, dout.path, dout.qual, dout.out_len, dout.status, dout.out_stride, nullptr};
    hipError_t e = launch_result_offsets(dout.out_len, n, W, d_offs, lh->stream);
    if (e == hipSuccess) e = launch_pack(rd, n, j->path_bytes, d_offs, reinterpret_cast<uint8_t *>(dev), lh->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pin, dev, header_bytes(n), hipMemcpyDeviceToHost, lh->stream);
    if (e == hipSuccess && has_amb)
        e = hipMemcpyAsync(pin + amb_off, dout.ambiguous, (size_t)n * 8, hipMemcpyDeviceToHost, lh->stream);
    if (e == hipSuccess) e = hipEventRecord(L->ev_header, lh->stream);
    if (e != hipSuccess) return lane_fail(j, FCD_E_HIP, std::string("result chunk: ") + hipGetErrorString(e));
    return FCD_OK;
}

void fail_from(fcd_job *j, int c) {  // (mutex held) chunk c and everything after it will never arrive
    for (int k = c; k < j->n_chunks; ++k)
        if (j->state[k] == 0) j->state[k] = 2;
    j->cancel = true;  // the other lanes stop at their next chunk
    j->cv.notify_all();
}

void lane_main(fcd_job *j, int lane) {
    (void)hipSetDevice(j->h->device);
    for (int c = lane; c < j->n_chunks; c += j->n_lanes) {
        {
            std::unique_lock<std::mutex> lk(j->mu);
            // the lane's buffers are free once the caller has moved on from chunk c - L
            j->cv.wait(lk, [&] { return j->cancel || c - j->n_lanes < j->released; });
            if (j->cancel) {
                fail_from(j, c);
                return;
            }
        }
        const int rc = issue_chunk(j, c);
        std::lock_guard<std::mutex> lk(j->mu);
        if (rc != FCD_OK) {
            if (j->upload_turn <= c) j->upload_turn = c + 1;
            fail_from(j, c);
            return;
        }
        j->state[c] = 1;
        j->cv.notify_all();
    }
}

int ensure_lanes(fcd_handle *h, int n_lanes) {
    while ((int)h->lanes.size() < n_lanes) {
        fcd_host_lane *L = new fcd_host_lane();
        int rc = fcd_create(h->device, &L->h);
        if (rc != FCD_OK) {
            delete L;
            h->err = "host pipeline: cannot create a lane";
            return rc;
        }
        L->h->is_lane = true;
        if (hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&L->ev_header, hipEventDisableTiming) != hipSuccess) {
            fcd_destroy(L->h);
            delete L;
            h->err = "host pipeline: cannot create a lane's stream / event";
            return FCD_E_HIP;
        }
        h->lanes.push_back(L);
    }
    // the lanes share the handle's workspace limit
    for (fcd_host_lane *L : h->lanes) L->h->ws_limit = h->ws_limit > 0 ? std::max<int64_t>(h->ws_limit / n_lanes, 1) : 0;
    for (fcd_host_lane *L : h->lanes) L->h->tie_order = effective_tie_order(h);  // the lanes search as their owner would
    return FCD_OK;
}

int lanes_default(const fcd_handle *h) {
    // three: measured best or equal at BASELINE configs 2 / 3 / 4 (361 k vs 345 k, 175 k vs 163 k, 142 k vs 144 k reads/s
    // against four; a search takes ~3.5 ms however few reads it holds, so fewer, larger chunks cost less than the extra
    // overlap buys)
    const int n = h->pipe_lanes > 0 ? h->pipe_lanes : env_int("FCD_HOST_LANES", 3);
    return std::max(1, std::min(n, 16));
}

// reads per chunk: the batch split evenly over the lanes, at most 2048 reads (a chunk's upload should not take
// much longer than its search), whole wavefront pairs
int64_t chunk_default(const fcd_handle *h, int64_t B, int n_lanes) {
    const int64_t forced = h->pipe_chunk > 0 ? h->pipe_chunk : env_int("FCD_HOST_CHUNK", 0);
    if (forced > 0) return forced;
    int64_t c = (B + n_lanes - 1) / n_lanes;
    c = std::min<int64_t>(std::max<int64_t>(c, 64), 2048);
    return (c + 63) & ~63ll;
}

// A chunk is uploaded as ONE copy of the span its reads cover.  In read-major storage that span is the chunk; in
// time-major storage -- the (T, B, N) tensor seen as a batch: stride_read = S * N, stride_t = B * S * N -- it is nearly
// the whole tensor, for every chunk: the pipeline would move the input over PCIe n_chunks times.  Such a batch goes
// up once (one chunk; fcd_*_host: the single-shot path).
bool chunk_span_wasteful(const fcd_batch *in, int64_t chunk, bool crf) {
    const int64_t n = std::min<int64_t>(chunk, in->n_reads);
    if (n <= 0 || in->T <= 0 || n >= in->n_reads) return false;
    const double S = crf ? (double)in->S : 1.0;
    const double dense = (double)n * (double)in->T * S * (double)in->N;
    double span = 1.0 + (double)(n - 1) * (double)in->stride_read + (double)(in->T - 1) * (double)in->stride_t +
                  (double)(in->N - 1) * (double)in->stride_n;
    if (crf) span += (double)(in->S - 1) * (double)in->stride_s;
    return span > 2.0 * dense;
}

int job_begin(fcd_handle *h, const fcd_batch *in, const HostCall &call, int want, fcd_job **out,
              const void *const *read_ptrs = nullptr, const int64_t *read_rows = nullptr) {
    if (!h || !out) return FCD_E_INVALID;
    *out = nullptr;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (h->is_lane) return FCD_E_INVALID;
    if (h->job_active) {
        h->err = "a host job is already running on this handle (fcd_job_end it first)";
        return FCD_E_INVALID;
    }
    // the same argument checks as fcd_*_host, on a result of the job's own shape
    fcd_result shape{};
    uint8_t dummy = 0;
    shape.labels = &dummy;
    shape.out_len = reinterpret_cast<uint32_t *>(&dummy);
    shape.status = reinterpret_cast<int32_t *>(&dummy);
    shape.out_stride = in ? std::max<int64_t>(in->T, 1) : 1;
    int rc = host_check(h, in, &shape, call);
    if (rc) return rc;
    if ((want & FCD_JOB_QUAL) && call.op != HostOp::Viterbi && call.op != HostOp::CrfGreedy) {
        h->err = "qualities exist for viterbi_search / crf_greedy_search only";
        return FCD_E_INVALID;
    }
    if ((want & FCD_JOB_AMBIGUOUS) && call.op != HostOp::Beam && call.op != HostOp::CrfBeam) want &= ~FCD_JOB_AMBIGUOUS;
    fcd_job *j = new fcd_job();
    j->h = h;
    j->in = *in;
    j->call = call;
    j->read_ptrs = read_ptrs;
    j->read_rows = read_rows;
    j->want = want;
    j->path_bytes = in->T <= 65535 ? 2 : 4;
    const int64_t B = in->n_reads;
    j->n_lanes = lanes_default(h);
    j->chunk = chunk_default(h, B, j->n_lanes);
    if (chunk_span_wasteful(in, j->chunk, call.op == HostOp::CrfBeam || call.op == HostOp::CrfGreedy)) j->chunk = std::max<int64_t>(B, 1);
    j->n_chunks = (int)((B + j->chunk - 1) / j->chunk);
    j->n_lanes = std::max(1, std::min(j->n_lanes, j->n_chunks));
    j->state.assign((size_t)j->n_chunks, 0);
    if (j->n_chunks > 0) {
        hipError_t de;
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        de = prev == h->device ? hipSuccess : hipSetDevice(h->device);
        rc = de == hipSuccess ? ensure_lanes(h, j->n_lanes) : FCD_E_HIP;
        if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
        if (rc) {
            delete j;
            return rc;
        }
        for (int k = 0; k < j->n_lanes; ++k) j->workers.emplace_back(lane_main, j, k);
    }
    h->job_active = true;
    *out = j;
    return FCD_OK;
}

}  // namespace

namespace fcd {

bool host_job_wanted(fcd_handle *h, const fcd_batch *in, const HostCall &c) {
    if (h->is_lane || h->job_active) return false;
    if (lanes_default(h) < 2) return false;
    const bool crf = c.op == HostOp::CrfBeam || c.op == HostOp::CrfGreedy;
    const double bytes = (double)in->n_reads * (double)in->T * (double)(crf ? in->S : 1) * (double)in->N *
                         (in->dtype == FCD_DTYPE_F32 ? 4.0 : 2.0);
    if (chunk_span_wasteful(in, chunk_default(h, in->n_reads, lanes_default(h)), crf)) return false;  // e.g. time-major storage
    if (h->pipe_min_bytes >= 0) return in->n_reads >= 2 && bytes >= (double)h->pipe_min_bytes;
    return in->n_reads >= 128 && bytes >= (double)(16 << 20);
}

int host_job_run_fixed(fcd_handle *h, const fcd_batch *in, const fcd_result *out, const HostCall &c) {
    int want = 0;
    if (out->path) want |= FCD_JOB_PATH;
    if (out->qual && (c.op == HostOp::Viterbi || c.op == HostOp::CrfGreedy)) want |= FCD_JOB_QUAL;
    if (out->ambiguous) want |= FCD_JOB_AMBIGUOUS;
    fcd_job *j = nullptr;
    int rc = job_begin(h, in, c, want, &j);
    if (rc) return rc;
    fcd_chunk ch;
    while ((rc = fcd_job_next(j, &ch)) == FCD_OK) {
        for (int64_t i = 0; i < ch.n_reads; ++i) {
            const int64_t r = ch.read_begin + i;
            const uint64_t off = ch.offsets[i];
            const size_t len = (size_t)(ch.offsets[i + 1] - off);
            out->out_len[r] = ch.out_len[i];
            if (out->status) out->status[r] = ch.status[i];
            memcpy(out->labels + r * out->out_stride, ch.labels + off, len);
            if (out->path) {
                uint32_t *dst = out->path + r * out->out_stride;
                if (ch.path_bytes == 2) {
                    const uint16_t *src = static_cast<const uint16_t *>(ch.path) + off;
                    for (size_t k = 0; k < len; ++k) dst[k] = src[k];
                } else {
                    memcpy(dst, static_cast<const uint32_t *>(ch.path) + off, len * 4);
                }
            }
            if (ch.qual) memcpy(out->qual + r * out->out_stride, ch.qual + off, len * 4);
            if (ch.ambiguous) memcpy(out->ambiguous + 2 * r, ch.ambiguous + 2 * i, 8);
        }
    }
    const int rc_end = fcd_job_end(j);
    if (rc == FCD_JOB_DONE) rc = FCD_OK;
    return rc != FCD_OK ? rc : rc_end;
}

void host_job_release_lanes(fcd_handle *h, bool destroy) {
    for (fcd_host_lane *L : h->lanes) {
        (void)hipStreamSynchronize(L->h->stream);
        (void)hipStreamSynchronize(L->copy_stream);
        if (L->pack_dev) (void)hipFree(L->pack_dev);
        if (L->pack_pin) (void)hipHostFree(L->pack_pin);
        if (L->in_pin) (void)hipHostFree(L->in_pin);
        L->pack_dev = L->pack_pin = L->in_pin = nullptr;
        L->pack_dev_bytes = L->pack_pin_bytes = L->in_pin_bytes = 0;
        if (destroy) {
            (void)hipStreamDestroy(L->copy_stream);
            (void)hipEventDestroy(L->ev_header);
            fcd_destroy(L->h);
            delete L;
        } else {
            fcd_release_workspace(L->h);
        }
    }
    if (destroy) h->lanes.clear();
}

}  // namespace fcd

extern "C" {

int fcd_viterbi_search_host_begin(fcd_handle *h, const fcd_batch *in, int collapse_repeats, int want, fcd_job **job) {
    HostCall c{HostOp::Viterbi};
    c.collapse = collapse_repeats;
    return job_begin(h, in, c, want, job);
}

int fcd_beam_search_host_begin(fcd_handle *h, const fcd_batch *in, int64_t beam_size, float beam_cut_threshold,
                               int collapse_repeats, int kernel, int want, fcd_job **job) {
    HostCall c{HostOp::Beam};
    c.collapse = collapse_repeats;
    c.beam_size = beam_size;
    c.thr = beam_cut_threshold;
    c.kernel = kernel;
    return job_begin(h, in, c, want, job);
}

namespace {
// a batch whose reads are separate contiguous (rows[r], N) host matrices: described as the dense batch they are
// gathered into (T = the longest read)
int ptrs_batch(fcd_handle *h, const void *const *reads, const int64_t *rows, int64_t n_reads, int64_t N, int dtype,
               fcd_batch *b) {
    if (!h) return FCD_E_INVALID;
    if (n_reads < 0 || N < 1 || (n_reads > 0 && (!reads || !rows))) {
        std::lock_guard<std::recursive_mutex> g(h->mu);
        h->err = "reads / rows missing";
        return FCD_E_INVALID;
    }
    int64_t T = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        if (rows[r] < 0 || (rows[r] > 0 && !reads[r])) {
            std::lock_guard<std::recursive_mutex> g(h->mu);
            h->err = "a read is missing (null pointer or negative row count)";
            return FCD_E_INVALID;
        }
        T = std::max(T, rows[r]);
    }
    memset(b, 0, sizeof *b);
    static const float dummy = 0.0f;
    b->post = &dummy;  // (never read: every chunk is gathered from `reads`)
    b->n_reads = n_reads;
    b->T = T;
    b->S = 1;
    b->N = N;
    b->stride_read = T * N;
    b->stride_t = N;
    b->stride_n = 1;
    b->lengths = rows;
    b->dtype = dtype;
    return FCD_OK;
}
}  // namespace

int fcd_viterbi_search_host_ptrs_begin(fcd_handle *h, const void *const *reads, const int64_t *rows, int64_t n_reads,
                                       int64_t N, int dtype, int collapse_repeats, int want, fcd_job **job) {
    fcd_batch b;
    const int rc = ptrs_batch(h, reads, rows, n_reads, N, dtype, &b);
    if (rc) return rc;
    HostCall c{HostOp::Viterbi};
    c.collapse = collapse_repeats;
    return job_begin(h, &b, c, want, job, reads, rows);
}

int fcd_beam_search_host_ptrs_begin(fcd_handle *h, const void *const *reads, const int64_t *rows, int64_t n_reads,
                                    int64_t N, int dtype, int64_t beam_size, float beam_cut_threshold,
                                    int collapse_repeats, int kernel, int want, fcd_job **job) {
    fcd_batch b;
    const int rc = ptrs_batch(h, reads, rows, n_reads, N, dtype, &b);
    if (rc) return rc;
    HostCall c{HostOp::Beam};
    c.collapse = collapse_repeats;
    c.beam_size = beam_size;
    c.thr = beam_cut_threshold;
    c.kernel = kernel;
    return job_begin(h, &b, c, want, job, reads, rows);
}

int fcd_crf_beam_search_host_begin(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                                   int64_t init_stride, int64_t beam_size, float beam_cut_threshold, int kernel,
                                   int want, fcd_job **job) {
    HostCall c{HostOp::CrfBeam};
    c.beam_size = beam_size;
    c.thr = beam_cut_threshold;
    c.kernel = kernel;
    c.init = init;
    c.n_init = n_init;
    c.init_stride = init_stride;
    return job_begin(h, in, c, want, job);
}

int fcd_crf_greedy_search_host_begin(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                                     int64_t init_stride, int want, fcd_job **job) {
    HostCall c{HostOp::CrfGreedy};
    c.init = init;
    c.n_init = n_init;
    c.init_stride = init_stride;
    return job_begin(h, in, c, want, job);
}

int fcd_set_host_pipeline(fcd_handle *h, int lanes, int64_t chunk_reads, int64_t min_bytes) {
    if (!h || lanes < 0 || lanes > 16 || chunk_reads < 0) return FCD_E_INVALID;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (h->job_active) return FCD_E_INVALID;
    h->pipe_lanes = lanes;
    h->pipe_chunk = chunk_reads;
    h->pipe_min_bytes = min_bytes < 0 ? -1 : min_bytes;
    return FCD_OK;
}

int fcd_job_chunks(const fcd_job *j, int64_t *chunk_reads, int *n_lanes) {
    if (!j) return -1;
    if (chunk_reads) *chunk_reads = j->chunk;
    if (n_lanes) *n_lanes = j->n_lanes;
    return j->n_chunks;
}

int fcd_job_next(fcd_job *j, fcd_chunk *out) {
    if (!j || !out) return FCD_E_INVALID;
    int c;
    {
        std::unique_lock<std::mutex> lk(j->mu);
        j->released = j->handed;  // the caller is done with the previous view: its lane may be reused
        j->cv.notify_all();
        if (j->handed >= j->n_chunks) return FCD_JOB_DONE;
        c = j->handed;
        j->cv.wait(lk, [&] { return j->state[c] != 0; });
        if (j->state[c] == 2) {
            j->h->err = j->err.empty() ? "host job cancelled" : j->err;
            return j->rc != FCD_OK ? j->rc : FCD_E_INVALID;
        }
    }
    fcd_handle *h = j->h;
    fcd_host_lane *L = h->lanes[c % j->n_lanes];
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device) (void)hipSetDevice(h->device);
    struct Restore {
        int prev, dev;
        ~Restore() {
            if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
        }
    } restore{prev, h->device};
    const int64_t b0 = (int64_t)c * j->chunk;
    const int64_t n = std::min<int64_t>(j->chunk, j->in.n_reads - b0);
    const int64_t W = std::max<int64_t>(j->in.T, 1);
    const bool has_path = (j->want & FCD_JOB_PATH) != 0, has_qual = (j->want & FCD_JOB_QUAL) != 0;
    const bool has_amb = (j->want & FCD_JOB_AMBIGUOUS) != 0;
    hipError_t e = hipEventSynchronize(L->ev_header);
    char *pin = reinterpret_cast<char *>(L->pack_pin);
    char *dev = reinterpret_cast<char *>(L->pack_dev);
    uint64_t total = 0;
    if (e == hipSuccess) {
        const uint32_t *hdr = reinterpret_cast<const uint32_t *>(pin);
        total = (uint64_t)hdr[0] | ((uint64_t)hdr[1] << 32);
        if (total > (uint64_t)n * (uint64_t)W) {
            h->err = "host job: corrupt chunk header";
            return FCD_E_HIP;
        }
        const size_t pay = payload_bytes(total, j->path_bytes, has_path, has_qual);
        if (pay) {
            e = hipMemcpyAsync(pin + header_bytes(n), dev + header_bytes(n), pay, hipMemcpyDeviceToHost, L->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(L->copy_stream);
        }
    }
    if (e != hipSuccess) {
        h->err = std::string("host job: ") + hipGetErrorString(e);
        return FCD_E_HIP;
    }
    const uint32_t *lens = reinterpret_cast<const uint32_t *>(pin + 16);
    j->offsets.resize((size_t)n + 1);
    uint64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        j->offsets[(size_t)i] = acc;
        acc += lens[i];
    }
    j->offsets[(size_t)n] = acc;
    if (acc != total) {
        h->err = "host job: chunk lengths do not add up";
        return FCD_E_HIP;
    }
    const size_t worst = header_bytes(n) + payload_bytes((uint64_t)n * (uint64_t)W, j->path_bytes, has_path, has_qual);
    const size_t amb_off = (worst + 15) & ~(size_t)15;
    out->read_begin = b0;
    out->n_reads = n;
    out->out_len = lens;
    out->status = reinterpret_cast<const int32_t *>(pin + 16 + 4 * (size_t)n);
    out->offsets = j->offsets.data();
    const char *p = pin + header_bytes(n);
    out->labels = reinterpret_cast<const uint8_t *>(p);
    size_t o = (size_t)((total + 3) & ~(uint64_t)3);
    out->path = has_path ? static_cast<const void *>(p + o) : nullptr;
    out->path_bytes = has_path ? j->path_bytes : 0;
    if (has_path) o += (size_t)total * (size_t)j->path_bytes;
    o = (o + 3) & ~(size_t)3;
    out->qual = has_qual ? reinterpret_cast<const float *>(p + o) : nullptr;
    out->ambiguous = has_amb ? reinterpret_cast<const uint32_t *>(pin + amb_off) : nullptr;
    std::lock_guard<std::mutex> lk(j->mu);
    j->handed = c + 1;
    return FCD_OK;
}

int fcd_job_end(fcd_job *j) {
    if (!j) return FCD_E_INVALID;
    {
        std::lock_guard<std::mutex> lk(j->mu);
        j->released = j->handed;
        if (j->handed < j->n_chunks) j->cancel = true;
        j->cv.notify_all();
    }
    for (std::thread &t : j->workers)
        if (t.joinable()) t.join();
    fcd_handle *h = j->h;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device) (void)hipSetDevice(h->device);
    int rc = FCD_OK;
    for (int k = 0; k < j->n_lanes && k < (int)h->lanes.size(); ++k) {
        if (hipStreamSynchronize(h->lanes[k]->h->stream) != hipSuccess) rc = FCD_E_HIP;
        if (hipStreamSynchronize(h->lanes[k]->copy_stream) != hipSuccess) rc = FCD_E_HIP;
    }
    if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
    {
        std::lock_guard<std::recursive_mutex> g(h->mu);
        h->job_active = false;
    }
    delete j;
    return rc;
}

}  // extern "C"
