// comm.hip -- the multi-GPU step of the path at the C-ABI level (host code).
//
// Reads shard across the ranks with no exchange inside the search; what is left is ONE gather of every rank's
// decoded results on the destination rank.  fast_ctc_decode_amd/dist.py does that through torch.distributed;
// a host that is not Python (north star: "host code stays Rust calling HIP through a thin C-ABI") gets the same
// step here, over RCCL directly:
//
//   fcd_gather_results_dev   offsets + pack of the used prefixes (pack.hip)            this rank's stream
//                            ncclAllReduce(MAX) of the label totals -> 8 bytes to the host (the one host wait:
//                            a gather needs one count for all ranks)
//                            ncclGather of the packed buffers over xGMI
//                            unpack of ALL shards with one pair of launches            destination rank only
//
// RCCL is looked up at run time (dlopen: librccl.so.1, the one PyTorch bundles if it is already in the process),
// so libfcd_hip.so has no link-time dependency on it and single-GPU users never load it.  An fcd_comm can also
// wrap a communicator the host already owns (fcd_comm_wrap), or none at all for world = 1.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fcd_internal.h"

using namespace fcd;

#define FCD_HIP(h, expr)                                                   \
    do {                                                                   \
        hipError_t e__ = (expr);                                           \
        if (e__ != hipSuccess) {                                           \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__); \
            return FCD_E_HIP;                                              \
        }                                                                  \
    } while (0)

namespace {

// the few RCCL declarations used (values as in rccl/rccl.h of ROCm 7: ncclUint8 = 1, ncclUint64 = 5, ncclMax = 2)
struct NcclId {
    char internal[FCD_COMM_ID_BYTES];
};
typedef void *NcclComm;
enum { kNcclUint8 = 1, kNcclUint64 = 5, kNcclMax = 2 };

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Gather)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // FCD_RCCL_LIBRARY names the library outright (a site's own build; tests/stubs' shared-memory stand-in)
        if (const char *named = getenv("FCD_RCCL_LIBRARY"))
            if (*named) r.lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        // a copy already in the process (PyTorch's) first: two RCCLs in one process would not share state
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (r.lib) break;
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.Gather = reinterpret_cast<decltype(r.Gather)>(dlsym(r.lib, "ncclGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Gather;
    });
    return r;
}

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct fcd_comm {
    fcd_handle *h = nullptr;
    NcclComm comm = nullptr;
    bool owns_comm = false;
    int world = 1, rank = 0;
    Buf send, recv, meta;  // grow-only device buffers: packed shard | world packed shards | offsets
    Buf fixed;             // [0, 32) the agreed {label total, out_stride, ~capacity, error}, [32, 36) bad-header flag,
                           // [64, 96) this rank's four words, [128, ...) first[world + 1]; allocated with the communicator
    std::vector<int64_t> counts;  // the counts `first` on the device was computed from
    void *pin = nullptr;   // page-locked bytes for the size / flag read-back
};

namespace {

int comm_fail(fcd_comm *c, int code, const std::string &msg) {
    c->h->err = msg;
    return code;
}

// test hooks (tests/capi/comm_world.c): FCD_DEBUG_FAIL_GATHER_ALLOC=<rank> makes that rank's gather buffers
// "unallocatable" once they have to grow -- the failure every rank must learn of before any of them enqueues its
// ncclGather; FCD_DEBUG_FAIL_GATHER_PREP=<rank> fails that rank's preparation (offsets workspace) BEFORE the size
// agreement -- the rank must still join the all-reduce and every rank must come back with the error.  A value that is
// not a plain rank number switches nothing on.
bool debug_rank_env(const char *name, int rank) {
    const char *e = getenv(name);
    if (!e || !*e) return false;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return end && *end == 0 && v >= 0 && v == (long)rank;
}
bool injected_alloc_failure(const fcd_comm *c) { return debug_rank_env("FCD_DEBUG_FAIL_GATHER_ALLOC", c->rank); }
bool injected_prep_failure(const fcd_comm *c) { return debug_rank_env("FCD_DEBUG_FAIL_GATHER_PREP", c->rank); }

int need(fcd_comm *c, Buf &b, size_t bytes) {
    if (b.cap >= bytes) return FCD_OK;
    if (b.p) {
        (void)hipStreamSynchronize(c->h->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&b.p, want) != hipSuccess) return comm_fail(c, FCD_E_NOMEM, "hipMalloc failed (gather buffers)");
    b.cap = want;
    return FCD_OK;
}

int nccl_check(fcd_comm *c, int rc, const char *what) {
    if (rc == 0) return FCD_OK;
    const Rccl &r = rccl();
    return comm_fail(c, FCD_E_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

constexpr size_t kFixedBytes = 128 + 65 * 8;

// The communicator's fixed-size pieces -- the words of the size agreement and the page-locked read-back -- exist from
// creation on (the handle's device current): nothing a gather needs BEFORE its first collective can then fail on one
// rank only, except the per-batch offsets workspace, whose failure travels through that collective (below).
fcd_comm *new_comm(fcd_handle *h, int world, int rank) {
    fcd_comm *c = new fcd_comm();
    c->h = h;
    c->world = world;
    c->rank = rank;
    if (hipMalloc(&c->fixed.p, kFixedBytes) != hipSuccess || hipMemset(c->fixed.p, 0, kFixedBytes) != hipSuccess ||
        hipHostMalloc(&c->pin, 64, hipHostMallocDefault) != hipSuccess) {
        if (c->fixed.p) (void)hipFree(c->fixed.p);
        h->err = "allocation failed (communicator)";
        delete c;
        return nullptr;
    }
    c->fixed.cap = kFixedBytes;
    return c;
}

}  // namespace

extern "C" {

int fcd_comm_unique_id(uint8_t id[FCD_COMM_ID_BYTES]) {
    if (!id) return FCD_E_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return FCD_E_UNSUPPORTED;
    NcclId nid;
    memset(&nid, 0, sizeof(nid));
    if (r.GetUniqueId(&nid) != 0) return FCD_E_HIP;
    memcpy(id, nid.internal, FCD_COMM_ID_BYTES);
    return FCD_OK;
}

int fcd_comm_create(fcd_handle *h, int world, int rank, const uint8_t id[FCD_COMM_ID_BYTES], fcd_comm **out) {
    if (!h || !out || !id || world < 1 || rank < 0 || rank >= world) return FCD_E_INVALID;
    *out = nullptr;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    Rccl &r = rccl();
    if (!r.ok) {
        h->err = "RCCL (librccl.so) could not be loaded";
        return FCD_E_UNSUPPORTED;
    }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (world > 64) {
        h->err = "more than 64 ranks";
        return FCD_E_UNSUPPORTED;
    }
    if (prev != h->device && hipSetDevice(h->device) != hipSuccess) return FCD_E_HIP;
    fcd_comm *c = new_comm(h, world, rank);  // (before ncclCommInitRank: a rank that cannot allocate never joins)
    if (!c) {
        if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
        return FCD_E_NOMEM;
    }
    NcclId nid;
    memcpy(nid.internal, id, FCD_COMM_ID_BYTES);
    const int rc = r.CommInitRank(&c->comm, world, nid, rank);  // one process per GPU: the handle's device
    if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
    if (rc != 0) {
        h->err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
        c->owns_comm = false;
        (void)fcd_comm_destroy(c);
        return FCD_E_HIP;
    }
    c->owns_comm = true;
    *out = c;
    return FCD_OK;
}

int fcd_comm_wrap(fcd_handle *h, void *nccl_comm, int world, int rank, fcd_comm **out) {
    if (!h || !out || world < 1 || rank < 0 || rank >= world) return FCD_E_INVALID;
    *out = nullptr;
    if (!nccl_comm && world != 1) return FCD_E_INVALID;  // no communicator: a single rank only
    if (nccl_comm && !rccl().ok) return FCD_E_UNSUPPORTED;
    if (world > 64) return FCD_E_UNSUPPORTED;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device && hipSetDevice(h->device) != hipSuccess) return FCD_E_HIP;
    fcd_comm *c = new_comm(h, world, rank);
    if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
    if (!c) return FCD_E_NOMEM;
    c->comm = nccl_comm;
    *out = c;
    return FCD_OK;
}

int fcd_comm_destroy(fcd_comm *c) {
    if (!c) return FCD_OK;
    fcd_handle *h = c->h;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device) (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (c->owns_comm && c->comm) (void)rccl().CommDestroy(c->comm);
    for (Buf *b : {&c->send, &c->recv, &c->meta, &c->fixed})
        if (b->p) (void)hipFree(b->p);
    if (c->pin) (void)hipHostFree(c->pin);
    if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
    delete c;
    return FCD_OK;
}

int fcd_gather_results_dev(fcd_comm *c, const fcd_result *res, int64_t n_reads, const int64_t *counts, int dst,
                           const fcd_result *out) {
    if (!c || !res || !counts || n_reads < 0 || dst < 0 || dst >= c->world) return FCD_E_INVALID;
    fcd_handle *h = c->h;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    if (counts[c->rank] != n_reads) return comm_fail(c, FCD_E_INVALID, "counts[rank] differs from n_reads");
    int64_t n_total = 0, n_max = 0;
    for (int k = 0; k < c->world; ++k) {
        if (counts[k] < 0) return comm_fail(c, FCD_E_INVALID, "negative read count");
        n_total += counts[k];
        n_max = std::max(n_max, counts[k]);
    }
    const bool is_dst = c->rank == dst;
    if (n_reads > 0 && (!res->labels || !res->path || !res->out_len))
        return comm_fail(c, FCD_E_INVALID, "null labels/path/out_len");
    if (is_dst && n_total > 0 && (!out || !out->labels || !out->out_len || out->out_stride < res->out_stride))
        return comm_fail(c, FCD_E_INVALID, "destination result missing or narrower than the shards");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device && hipSetDevice(h->device) != hipSuccess) return FCD_E_HIP;
    struct Restore {
        int prev, dev;
        ~Restore() {
            if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
        }
    } restore{prev, h->device};
    hipStream_t st = h->stream;
    const int W = (int)std::min<int64_t>(res->out_stride, 1 << 30);
    int rc = FCD_OK;
    // Nothing between here and the first collective may return on ONE rank only -- the others are about to enter the
    // all-reduce and would wait for this one for ever.  What fails here (the per-batch offsets workspace, a launch) is
    // remembered, the rank joins the reduction with its error word set, and EVERY rank returns an error after it.
    int pre_rc = FCD_OK;
    std::string pre_msg;
    auto pre_fail = [&](int code, const char *what) {
        if (pre_rc == FCD_OK) {
            pre_rc = code;
            pre_msg = what;
        }
    };
#define FCD_PRE(call)                                                            \
    do {                                                                         \
        const hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) pre_fail(FCD_E_HIP, hipGetErrorString(e_));        \
    } while (0)
    // meta: offsets of this shard [n_reads + 1] | unpack offsets [n_total + world] (destination)
    const size_t o_uoffs = ((size_t)(n_reads + 1) * 8 + 15) & ~(size_t)15;
    if (injected_prep_failure(c)) pre_fail(FCD_E_NOMEM, "hipMalloc failed (gather offsets; injected)");
    else if (need(c, c->meta, o_uoffs + (is_dst ? (size_t)(n_total + c->world) * 8 : 0) + 16) != FCD_OK)
        pre_fail(FCD_E_NOMEM, "hipMalloc failed (gather offsets)");
    char *meta = reinterpret_cast<char *>(c->meta.p);
    char *fixed = reinterpret_cast<char *>(c->fixed.p);
    uint64_t *d_offs = reinterpret_cast<uint64_t *>(meta);
    // fixed: [0, 32) the agreed {largest label total, largest out_stride, ~smallest capacity, any error} | [32, 36)
    // header flag | [64, 96) this rank's four words | [128, ...) prefix sums of the read counts
    uint64_t *d_max = reinterpret_cast<uint64_t *>(fixed);
    int32_t *d_bad = reinterpret_cast<int32_t *>(fixed + 32);
    uint64_t *d_mine = reinterpret_cast<uint64_t *>(fixed + 64);
    int64_t *d_first = reinterpret_cast<int64_t *>(fixed + 128);
    if (pre_rc == FCD_OK) FCD_PRE(launch_result_offsets(res->out_len, n_reads, W, d_offs, st));
    // Every rank must hand ncclGather the SAME byte count, and that depends on two things: the largest label total
    // and the width of the time indices (2 bytes below 65536 rows) -- i.e. on the largest out_stride.  Both are agreed
    // on with one 32-byte all-reduce (a rank whose shard is padded differently would otherwise send another size
    // and hang or corrupt the collective).
    // The third word is the complement of what this rank's buffers hold already (send; the destination: receive / world),
    // so the same reduction also says whether ANY rank will have to allocate: only then a second, 8-byte all-reduce
    // follows, in which the ranks tell each other whether they could (below).  The fourth word is 1 on a rank whose
    // preparation failed.
    uint64_t *pin64 = reinterpret_cast<uint64_t *>(c->pin);
    pin64[4] = 0;  // (label total: copied device to device below when there is one)
    pin64[5] = (uint64_t)res->out_stride;
    pin64[6] = ~(uint64_t)std::min<size_t>(c->send.cap, is_dst ? c->recv.cap / (size_t)c->world : ~(size_t)0);
    pin64[7] = pre_rc != FCD_OK ? 1u : 0u;
    FCD_PRE(hipMemcpyAsync(d_mine, pin64 + 4, 32, hipMemcpyHostToDevice, st));
    if (pre_rc == FCD_OK) {
        FCD_PRE(hipMemcpyAsync(d_mine, d_offs + n_reads, 8, hipMemcpyDeviceToDevice, st));
        if (pre_rc != FCD_OK) {  // (the error word has to say so: once more, synchronously)
            pin64[7] = 1u;
            (void)hipMemcpyAsync(d_mine, pin64 + 4, 32, hipMemcpyHostToDevice, st);
        }
    }
#undef FCD_PRE
    if (c->comm) {
        rc = nccl_check(c, rccl().AllReduce(d_mine, d_max, 4, kNcclUint64, kNcclMax, c->comm, st), "ncclAllReduce");
        if (rc) return rc;
    } else {
        FCD_HIP(h, hipMemcpyAsync(d_max, d_mine, 32, hipMemcpyDeviceToDevice, st));
    }
    // (the flag behind them is the PREVIOUS gather's header check: reported one call late, or by fcd_comm_synchronize)
    // (a failing copy or wait here is a lost device: this rank cannot learn the agreed sizes and returns; the other
    // ranks' collectives end with RCCL's own error)
    FCD_HIP(h, hipMemcpyAsync(c->pin, d_max, 40, hipMemcpyDeviceToHost, st));
    FCD_HIP(h, hipStreamSynchronize(st));  // the one host wait of the gather
    const uint64_t max_total = pin64[0], max_stride = pin64[1], min_cap = ~pin64[2];
    if (pin64[3] != 0)  // (the same verdict on every rank: nobody goes on to the gather)
        return pre_rc != FCD_OK ? comm_fail(c, pre_rc, "gather: " + pre_msg)
                                : comm_fail(c, FCD_E_HIP, "gather: another rank failed before the size agreement");
    // From here to the ncclGather nothing may return early on one rank only: the other ranks are about to enqueue
    // theirs.  What this rank has to complain about is remembered and reported once its own gather is in the stream.
    const char *late_error = nullptr;
    if (reinterpret_cast<const int32_t *>(c->pin)[8] != 0) {
        (void)hipMemsetAsync(d_bad, 0, 4, st);
        late_error = "gather: a shard's header contradicted the read counts (earlier call)";
    }
    const int Wmax = (int)std::min<uint64_t>(max_stride, 1u << 30);
    if (max_total > (uint64_t)n_max * (uint64_t)Wmax) return comm_fail(c, FCD_E_HIP, "gather: impossible label total");  // (the same on every rank)
    if (is_dst && n_total > 0 && (uint64_t)out->out_stride < max_stride)
        late_error = "gather: the destination's out_stride is narrower than a shard's (out_stride must match across ranks)";
    const int pb = max_stride <= 65535 ? 2 : 4;  // time indices are < out_stride (csrc/pack.hip)
    const int64_t nbytes = fcd_packed_result_bytes(n_max, (int64_t)max_total, pb);
    if (min_cap < (uint64_t)nbytes) {
        // Some rank's buffers have to grow (every rank knows: the capacities rode along in the reduction).  An
        // allocation can fail on one rank only -- and that rank must not walk away while the others enqueue a gather
        // it will never join.  So the ranks tell each other how it went before anybody enqueues anything: one more
        // 8-byte all-reduce, on growth rounds only (the buffers keep a quarter of slack: a steady workload has none).
        int grow_rc = injected_alloc_failure(c) ? comm_fail(c, FCD_E_NOMEM, "hipMalloc failed (gather buffers; injected)") : need(c, c->send, (size_t)nbytes);
        if (!grow_rc && is_dst) grow_rc = need(c, c->recv, (size_t)nbytes * (size_t)c->world);
        pin64[4] = grow_rc ? 1u : 0u;
        FCD_HIP(h, hipMemcpyAsync(d_mine, pin64 + 4, 8, hipMemcpyHostToDevice, st));
        if (c->comm) {
            rc = nccl_check(c, rccl().AllReduce(d_mine, d_max, 1, kNcclUint64, kNcclMax, c->comm, st), "ncclAllReduce");
            if (rc) return rc;
        } else {
            FCD_HIP(h, hipMemcpyAsync(d_max, d_mine, 8, hipMemcpyDeviceToDevice, st));
        }
        FCD_HIP(h, hipMemcpyAsync(c->pin, d_max, 8, hipMemcpyDeviceToHost, st));
        FCD_HIP(h, hipStreamSynchronize(st));
        if (pin64[0] != 0)  // (the same verdict on every rank: nobody is left waiting in a collective)
            return grow_rc ? grow_rc : comm_fail(c, FCD_E_NOMEM, "gather: another rank could not allocate its buffers");
    }
    ResultDesc wire{res->labels, res->path, nullptr, res->out_len, res->status, res->out_stride, nullptr};
    {   // (a launch that fails here is reported AFTER this rank's gather is in the stream, like the header check)
        const hipError_t pe = n_reads > 0 ? launch_pack(wire, n_reads, pb, d_offs, reinterpret_cast<uint8_t *>(c->send.p), st)
                                          : hipMemsetAsync(c->send.p, 0, 16, st);
        if (pe != hipSuccess && !late_error) late_error = "gather: packing this rank's shard failed";
    }
    const uint8_t *gathered = reinterpret_cast<const uint8_t *>(c->send.p);
    if (c->comm) {
        rc = nccl_check(c, rccl().Gather(c->send.p, is_dst ? c->recv.p : nullptr, (size_t)nbytes, kNcclUint8, dst,
                                         c->comm, st), "ncclGather");
        if (rc) return rc;
        gathered = reinterpret_cast<const uint8_t *>(c->recv.p);
    }
    if (late_error) return comm_fail(c, FCD_E_INVALID, late_error);
    if (!is_dst || n_total == 0) return FCD_OK;
    // shard s owns rows [first[s], first[s + 1]) of the destination; the prefix sums live on the device and
    // are refreshed only when the counts change
    if (c->counts.size() != (size_t)c->world || !std::equal(c->counts.begin(), c->counts.end(), counts)) {
        int64_t first[65];
        first[0] = 0;
        for (int k = 0; k < c->world; ++k) first[k + 1] = first[k] + counts[k];
        FCD_HIP(h, hipStreamSynchronize(st));
        FCD_HIP(h, hipMemcpy(d_first, first, (size_t)(c->world + 1) * 8, hipMemcpyHostToDevice));
        c->counts.assign(counts, counts + c->world);
    }
    ResultDesc o{out->labels, out->path, nullptr, out->out_len, out->status, out->out_stride, nullptr};
    FCD_HIP(h, launch_unpack_gathered(gathered, nbytes, c->world, d_first, n_total,
                                      reinterpret_cast<uint64_t *>(meta + o_uoffs), o, d_bad, st));
    return FCD_OK;
}

int fcd_comm_synchronize(fcd_comm *c) {
    if (!c) return FCD_E_INVALID;
    fcd_handle *h = c->h;
    std::lock_guard<std::recursive_mutex> g(h->mu);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != h->device && hipSetDevice(h->device) != hipSuccess) return FCD_E_HIP;
    int rc = FCD_OK;
    int32_t bad = 0;
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = FCD_E_HIP;
    if (rc == FCD_OK && c->fixed.p) {
        if (hipMemcpy(&bad, reinterpret_cast<char *>(c->fixed.p) + 32, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = FCD_E_HIP;
        if (bad) {
            (void)hipMemset(reinterpret_cast<char *>(c->fixed.p) + 32, 0, 4);
            rc = comm_fail(c, FCD_E_INVALID, "gather: the header of shard " + std::to_string(bad - 1) +
                                                 " contradicts the read counts / buffer size");
        }
    }
    if (prev >= 0 && prev != h->device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
