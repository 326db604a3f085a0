// envelope.hip -- alignment-band estimator for the duplex searches.
//
// The reference has no such function: beam_search_duplex defaults to the full envelope and its
// docstring only anticipates "a non-trivial default" (/root/reference/src/lib.rs:376-378).  The
// algorithm is therefore specified by tests/envelope_model.py (numpy), which this kernel reproduces
// exactly (integer work):
//   1. global alignment of the two label sequences of a pair under unit-cost edit distance -- one
//      wavefront per pair, one DP row at a time: the in-row dependency D[i][j-1] + 1 is a min-plus
//      prefix scan (D[i][j] = j + min_{j' <= j} (E[j'] - j')), done with DPP row shifts across the 64
//      lanes and a carry between 64-column chunks; the previous row lives in LDS as u16, the
//      "came from diagonal / from above" decisions leave the kernel as two 64-bit ballots per chunk;
//   2. traceback (diagonal, then up, then left) turns matches of equal labels into anchors
//      (time in read 1 -> time in read 2) through the two paths;
//   3. row i of the envelope is the anchors' piecewise-linear interpolation +- band, clipped, with
//      lo(0) = 0, hi(T1 - 1) = T2 and lo(i) <= hi(i - 1) (src/duplex.rs:485-488).
#include <limits.h>

#include <algorithm>

#include "device_utils.h"
#include "fcd_internal.h"

namespace fcd {

namespace {

struct EnvParams {
    EnvelopeArgs a;
    int64_t pair_begin;
};

constexpr int kBig = INT_MAX / 2;

// DPP controls (gfx9 family): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_or_big(int x) {
    return __builtin_amdgcn_update_dpp(kBig, x, CTRL, ROW_MASK, BANK_MASK, false);
}

// inclusive prefix minimum over the 64 lanes (the classic GCN scan: three shifts of the input,
// then shifts by 4 and 8 inside each row of 16, then the two row broadcasts)
__device__ __forceinline__ int wave_prefix_min(int x) {
    int t = min(x, dpp_or_big<0x111, 0xf, 0xf>(x));
    t = min(t, dpp_or_big<0x112, 0xf, 0xf>(x));
    t = min(t, dpp_or_big<0x113, 0xf, 0xf>(x));
    t = min(t, dpp_or_big<0x114, 0xf, 0xe>(t));
    t = min(t, dpp_or_big<0x118, 0xf, 0xc>(t));
    t = min(t, dpp_or_big<0x142, 0xa, 0xf>(t));
    t = min(t, dpp_or_big<0x143, 0xc, 0xf>(t));
    return t;
}

__device__ __forceinline__ uint64_t load_u64_l2(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void envelope_kernel(EnvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const EnvelopeArgs &a = p.a;
    const int lane = threadIdx.x;
    const int64_t local = blockIdx.x;
    const int64_t r = p.pair_begin + local;

    int64_t T1 = a.T1cap, T2 = a.T2cap;
    if (a.T1) { const int64_t v = a.T1[r]; T1 = v < 0 ? 0 : (v < T1 ? v : T1); }
    if (a.T2) { const int64_t v = a.T2[r]; T2 = v < 0 ? 0 : (v < T2 ? v : T2); }
    int L1 = (int)a.len1[r], L2 = (int)a.len2[r];
    L1 = L1 < 0 ? 0 : (L1 < T1 ? L1 : (int)T1);
    L2 = L2 < 0 ? 0 : (L2 < T2 ? L2 : (int)T2);
    L2 = L2 < a.L2cap ? L2 : (int)a.L2cap;
    const uint8_t *lab1 = a.labels1 + r * a.stride1;
    const uint8_t *lab2 = a.labels2 + r * a.stride2;
    const uint32_t *pth1 = a.path1 + r * a.stride1;
    const uint32_t *pth2 = a.path2 + r * a.stride2;
    uint64_t *dirs = a.dirs + local * a.dirs_stride;
    int32_t *anchor = a.anchor + local * (a.T1cap + 1);
    uint64_t *env = a.env + r * a.env_stride * 2;
    const int nchunk = a.nchunk;

    // LDS: two u16 DP rows of L2cap + 1 entries (L2cap = the batch's longest read-2 labelling), then
    // the labels of read 2
    const int rowlen = (int)a.L2cap + 1;
    uint16_t *prev = reinterpret_cast<uint16_t *>(smem);
    uint16_t *cur = prev + rowlen;
    uint8_t *s2 = reinterpret_cast<uint8_t *>(cur + rowlen);

    for (int64_t t = lane; t <= T1; t += kWave) anchor[t] = -1;
    for (int j = lane; j <= L2; j += kWave) prev[j] = (uint16_t)j;
    for (int j = lane; j < L2; j += kWave) s2[j] = lab2[j];
    wave_sync();

    // ---- 1. edit-distance rows ----
    int lab_reg = 0;
    for (int i = 1; i <= L1; ++i) {
        if (((i - 1) & 63) == 0) lab_reg = (i - 1 + lane < L1) ? lab1[i - 1 + lane] : 0;
        const int la = __shfl(lab_reg, (i - 1) & 63);
        int carry = i;  // D[i][0] - 0
        if (lane == 0) cur[0] = (uint16_t)i;
        const int nact = (L2 + 63) >> 6;
        for (int c = 0; c < nact; ++c) {
            const int j = c * 64 + lane + 1;
            const bool act = j <= L2;
            const int jj = act ? j : 1;
            const int up = (int)prev[jj] + 1;
            const int dg = (int)prev[jj - 1] + ((int)s2[jj - 1] != la ? 1 : 0);
            const int e = up < dg ? up : dg;
            int f = act ? e - j : kBig;
            f = wave_prefix_min(f);
            f = f < carry ? f : carry;
            const int d = f + j;
            carry = __shfl(f, 63);
            const uint64_t m_diag = ballot(act && d == dg);
            const uint64_t m_up = ballot(act && d == up);
            if (lane == 0) {
                uint64_t *q = dirs + ((int64_t)(i - 1) * nchunk + c) * 2;
                q[0] = m_diag;
                q[1] = m_up;
            }
            if (act) cur[j] = (uint16_t)d;
        }
        wave_sync();
        uint16_t *tmp = prev;
        prev = cur;
        cur = tmp;
    }

    // ---- 2. traceback: one lane chases the decisions; matches of equal labels become anchors ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) {
        int i = L1, j = L2;
        while (i > 0 && j > 0) {
            const uint64_t *q = dirs + ((int64_t)(i - 1) * nchunk + ((j - 1) >> 6)) * 2;
            const uint64_t md = load_u64_l2(q), mu = load_u64_l2(q + 1);
            const int bit = (j - 1) & 63;
            if ((md >> bit) & 1ull) {
                if (lab1[i - 1] == s2[j - 1]) {
                    const uint32_t t1 = pth1[i - 1];
                    if (t1 > 0 && (int64_t)t1 <= T1) anchor[t1] = (int32_t)pth2[j - 1];
                }
                --i;
                --j;
            } else if ((mu >> bit) & 1ull) {
                --i;
            } else {
                --j;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (T1 <= 0) return;

    // ---- 3a. next anchor time for every row (suffix minimum), parked in env[i].hi ----
    const int band = (int)a.band;
    {
        int carry = (int)T1;  // the closing anchor (T1, T2)
        const int nrow = (int)((T1 + 63) >> 6);
        for (int c = nrow - 1; c >= 0; --c) {
            // lane k handles row c*64 + 63 - k so that a prefix scan over lanes is a suffix scan over rows
            const int i = c * 64 + 63 - lane;
            const bool act = i < T1;
            // "next" = smallest anchored time strictly greater than i
            int v = kBig;
            if (act && i + 1 < T1 && load_i32_l2(&anchor[i + 1]) >= 0) v = i + 1;
            v = wave_prefix_min(v);
            // exclusive over rows > i already, because row i looked at time i + 1
            v = v < carry ? v : carry;
            carry = __shfl(v, 63);
            if (act) env[(int64_t)i * 2 + 1] = (uint64_t)v;
        }
    }
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // ---- 3b. previous anchor (prefix maximum), interpolation, band, row contiguity ----
    {
        int carry_t = 0;        // the opening anchor (0, 0)
        int carry_hi = 0;       // hi of the last row of the previous chunk
        const int nrow = (int)((T1 + 63) >> 6);
        for (int c = 0; c < nrow; ++c) {
            const int i = c * 64 + lane;
            const bool act = i < T1;
            int v = -kBig;  // "previous" = largest anchored time <= i (time 0 is never anchored)
            if (act && i > 0 && load_i32_l2(&anchor[i]) >= 0) v = i;
            v = -wave_prefix_min(-v);
            v = v > carry_t ? v : carry_t;
            carry_t = __shfl(v, 63);
            const int ta = v;
            const int ua = ta > 0 ? load_i32_l2(&anchor[act ? ta : 0]) : 0;
            const int tb = act ? (int)load_u64_l2(&env[(int64_t)i * 2 + 1]) : (int)T1;
            const int ub = tb < T1 ? load_i32_l2(&anchor[tb]) : (int)T2;
            const int64_t num = (int64_t)(i - ta) * (int64_t)(ub - ua);
            const int den = tb - ta > 0 ? tb - ta : 1;
            const int cc = ua + (int)(num / den);  // num >= 0: floor
            int lo = cc - band;
            lo = lo < 0 ? 0 : lo;
            int hi = cc + band + 1;
            hi = hi > T2 ? (int)T2 : hi;
            if (i == 0) lo = 0;
            if (i == T1 - 1) hi = (int)T2;
            int hprev = __shfl_up(hi, 1);
            if (lane == 0) hprev = carry_hi;
            if (i > 0 && lo > hprev) lo = hprev;
            carry_hi = __shfl(hi, 63);
            if (act) {
                env[(int64_t)i * 2] = (uint64_t)lo;
                env[(int64_t)i * 2 + 1] = (uint64_t)hi;
            }
        }
    }
}

}  // namespace

size_t envelope_lds_bytes(int64_t L2cap) { return (size_t)(2 * (L2cap + 1) * 2 + L2cap + 16); }

namespace {
__global__ void max_u32_kernel(const uint32_t *a, const uint32_t *b, int64_t n, uint32_t *out2) {
    uint32_t ma = 0, mb = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        ma = max(ma, a[i]);
        mb = max(mb, b[i]);
    }
    atomicMax(&out2[0], ma);
    atomicMax(&out2[1], mb);
}
}  // namespace

// out2[0] = max(a), out2[1] = max(b); out2 must be zeroed by the caller
hipError_t launch_max_u32(const uint32_t *a, const uint32_t *b, int64_t n, uint32_t *out2, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 256);
    hipLaunchKernelGGL(max_u32_kernel, dim3(blocks), dim3(256), 0, stream, a, b, n, out2);
    return hipGetLastError();
}

hipError_t launch_envelope(const EnvelopeArgs &a, int64_t pair_begin, int64_t n_pairs, hipStream_t stream) {
    if (n_pairs <= 0) return hipSuccess;
    EnvParams p{a, pair_begin};
    hipLaunchKernelGGL(envelope_kernel, dim3((unsigned)n_pairs), dim3(64), envelope_lds_bytes(a.L2cap), stream, p);
    return hipGetLastError();
}

}  // namespace fcd
