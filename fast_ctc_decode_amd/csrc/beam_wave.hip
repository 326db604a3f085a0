// beam_wave.hip -- register-resident CTC prefix beam search for search::beam_search
// (/root/reference/src/search.rs:159-301): beam_size <= 8, 3 <= N <= 7.  One read per wavefront.
//
// Lane map (wave64): lane = 8*i + k.  i = beam slot (beam kept in rank order, as the reference's
// sorted Vec), k = 0 is the slot's own node, k = 1..NL the child reached by label k-1, lane 8*i+7
// is a scratch target for cross-lane pushes.  Every lane of group i carries a copy of slot i's
// (node, label_prob, gap_prob, tip label, depth); lane (i,k>=1) also carries node i's child entry
// for label k-1.  The whole search state therefore lives in VGPRs; LDS is used only as the
// cross-lane network (ds_permute / ds_bpermute) and for the 64 sort keys of the prune step.
//
// Per timestep:
//   * the posterior row comes from a 64-row register tile (lane r holds row tile*64+r, loaded one
//     tile ahead; v_readlane broadcasts row t to SGPRs) -- no LDS, no per-step memory latency;
//   * child lanes evaluate the extension (:200-239); an extension whose target is already in the
//     beam is pushed to that slot's lane 0 (ds_permute), which adds it to its own blank/stay
//     terms -- the reference's sort-by-node + fold (:245-260), order-independent because at most
//     two non-zero f32 addends ever meet;
//   * new tree nodes get ids in the reference's creation order via ballot + prefix popcount;
//   * pruning ranks the <= 40 candidates exactly on a 64-bit key (probability desc, node asc);
//   * the survivors are gathered into rank order with ds_bpermute and divided by the top
//     probability (:278-282, IEEE f32 division).
//
// Tree arena (HBM, per read): rec[node] = {parent, time<<3 | label}; rows[node] = child entries.
// A child entry is -1 (none) or id | EVER, EVER marking children that have themselves been in the
// beam: only those can own children, so only their row is ever re-read when they re-enter the
// beam (3.9 % of steps on BASELINE's generator) -- everything else stays in registers.
#include "device_utils.h"
#include "fcd_internal.h"

namespace fcd {

namespace {

constexpr int kEver = 1 << 30;
constexpr int kIdMask = kEver - 1;

struct WaveParams {
    BatchDesc in;
    BeamArgs a;
    WaveArena arena;
    ResultDesc out;
    int64_t read_begin;
};

__device__ __forceinline__ int bperm(int src_lane, int v) {
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ __forceinline__ float bpermf(int src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int perm(int dst_lane, int v) {
    return __builtin_amdgcn_ds_permute(dst_lane << 2, v);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int N>
__device__ __forceinline__ void load_tile(float (&dst)[N], const float *post, int64_t row, int64_t T,
                                          int64_t st_t, int64_t st_n) {
    const bool ok = row < T;
    const float *p = post + (ok ? row : 0) * st_t;
#pragma unroll
    for (int c = 0; c < N; ++c) dst[c] = ok ? p[c * st_n] : 0.0f;
}

constexpr int kWavesPerBlock = 4;

template <int N>
__global__ __launch_bounds__(64 * kWavesPerBlock) void beam_wave_kernel(WaveParams p) {
    constexpr int NL = N - 1;
    constexpr int RW = NL <= 4 ? 4 : 8;  // child-row width in the arena
    __shared__ uint64_t s_keys[kWavesPerBlock][64];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t local = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (local >= p.in.n_reads) return;  // n_reads here = reads in this launch
    const int64_t r = p.read_begin + local;
    uint64_t *keys = s_keys[wave];

    const int i = lane >> 3, k = lane & 7;
    const bool is_self = k == 0;
    const bool is_child = k >= 1 && k <= NL;
    const int l = k - 1;
    const int beam_size = p.a.beam_size;
    const bool collapse = p.a.collapse != 0;
    const float thr = p.a.thr;

    int64_t T = p.in.T;
    if (p.in.lengths) {
        int64_t t = p.in.lengths[r];
        T = t < 0 ? 0 : (t < T ? t : T);
    }
    const float *post = p.in.post + r * p.in.stride_read;
    const int64_t st_t = p.in.stride_t, st_n = p.in.stride_n;
    int2 *rec = p.arena.rec + local * p.arena.cap_nodes;
    int32_t *rows = p.arena.rows + local * p.arena.cap_nodes * RW;
    const int cap = (int)p.arena.cap_nodes;

    // ---- beam state (search.rs:170-175: root, label_prob 0, gap_prob 1) ----
    int node = -1;
    float lp = 0.0f, gp = 1.0f;
    int tip = -1;
    int depth = 0;
    int child = -1;
    int B = 1;
    int nn = 0;

    float cur[N], nxt[N];
    load_tile<N>(cur, post, lane, T, st_t, st_n);
    load_tile<N>(nxt, post, 64 + lane, T, st_t, st_n);

    for (int64_t t = 0; t < T; ++t) {
        const int rr = (int)(t & 63);
        if (rr == 0 && t > 0) {
#pragma unroll
            for (int c = 0; c < N; ++c) cur[c] = nxt[c];
            load_tile<N>(nxt, post, t + 64 + lane, T, st_t, st_n);
        }
        float pr[N];
#pragma unroll
        for (int c = 0; c < N; ++c) pr[c] = readlane_f(cur[c], rr);
        const float pr0 = pr[0];
        float pk = pr0, ptip = 0.0f;
#pragma unroll
        for (int c = 1; c <= NL; ++c) {
            pk = (k == c) ? pr[c] : pk;
            ptip = (tip + 1 == c) ? pr[c] : ptip;
        }
        const bool grp = i < B;

        // ---- child lanes: extension by label l (:200-239) ----
        const bool pass = !(pk < thr);  // :201 skips only when pr_b < thr
        const bool rep = collapse && l == tip;
        const float contrib = rep ? gp * pk : (lp + gp) * pk;
        const bool exists = child >= 0;
        const int cid = child & kIdMask;
        const bool cvalid = grp && is_child && pass && (exists || !rep || gp > 0.0f);  // :212-218

        // is the extension's target already in the beam?  (beam node ids -> SGPRs)
        int mslot = -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int bn = __builtin_amdgcn_readlane(node, j * 8);
            if (j < B) mslot = (cid == bn) ? j : mslot;
        }
        const bool merged = cvalid && exists && mslot >= 0;
        const int dst = merged ? mslot * 8 : (lane | 7);
        const float inc = __int_as_float(perm(dst, __float_as_int(merged ? contrib : 0.0f)));
        const int incv = perm(dst, merged ? 1 : 0);

        // ---- self lanes: blank (:191-198) + repeat-stay (:206-211) + incoming extension ----
        const bool blank = pr0 > thr;
        const float gpn = (lp + gp) * pr0;
        const bool stay = collapse && tip >= 0 && !(ptip < thr);
        const float lpn = lp * ptip;
        const bool has_inc = is_self && incv != 0;
        const float slp = (stay ? lpn : 0.0f) + (has_inc ? inc : 0.0f);
        const float sgp = blank ? gpn : 0.0f;
        const bool svalid = grp && is_self && (blank || stay || has_inc);

        const bool valid = is_self ? svalid : (cvalid && !merged);
        const float clp = is_self ? slp : contrib;
        const float cgp = is_self ? sgp : 0.0f;
        const float prob = clp + cgp;

        // ---- tree.rs:125-145 add_node: ids in (beam order, label order) == lane order ----
        const bool is_new = cvalid && !exists;
        const uint64_t m_new = __ballot(is_new);
        const int newid = nn + popc64(m_new & lanemask_lt());
        nn += popc64(m_new);
        if (nn > cap) {
            if (lane == 0) {
                p.out.status[r] = FCD_ST_INTERNAL;
                p.out.out_len[r] = 0;
            }
            return;
        }
        if (is_new) {
            rec[newid] = make_int2(node, (int)(t << 3) | l);
            if (node >= 0) rows[(int64_t)node * RW + l] = newid;
            child = newid;
        }
        const int id = is_self ? node : (is_new ? newid : cid);

        // ---- search.rs:261-277 ----
        const int n_valid = popc64(__ballot(valid));
        const bool any_nan = __ballot(valid && prob != prob) != 0ull;
        if ((n_valid >= 2 && any_nan) || n_valid == 0) {
            if (lane == 0) {
                p.out.status[r] = n_valid == 0 ? FCD_ST_RAN_OUT_OF_BEAM : FCD_ST_INCOMPARABLE;
                p.out.out_len[r] = 0;
            }
            return;
        }

        // ---- prune: exact rank on (probability desc, node asc) ----
        const uint64_t key = valid ? (prob == prob ? make_key(prob, id) : 1ull) : 0ull;
        keys[lane] = key;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < B) {
#pragma unroll
                for (int c = 0; c <= NL; ++c) rank += (keys[j * 8 + c] > key) ? 1 : 0;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- gather the survivors into rank order ----
        const int Bn = n_valid < beam_size ? n_valid : beam_size;
        const bool sel = valid && rank < beam_size;
        // a child entering the beam for the first time: mark it EVER in its parent's row (the
        // register copy here, the HBM copy below) and give it an empty row of its own
        const bool first_entry = sel && is_child && !(exists && (child & kEver));
        if (first_entry) {
            child = id | kEver;
            if (node >= 0) rows[(int64_t)node * RW + l] = child;
            int32_t *row = rows + (int64_t)id * RW;
            if (RW == 4) {
                *reinterpret_cast<int4 *>(row) = make_int4(-1, -1, -1, -1);
            } else {
                *reinterpret_cast<int4 *>(row) = make_int4(-1, -1, -1, -1);
                *reinterpret_cast<int4 *>(row + 4) = make_int4(-1, -1, -1, -1);
            }
        }
        const int kind = is_self ? 0 : (first_entry ? 1 : 2);  // 2: re-entering, row is in HBM
        const int src0 = perm(sel ? rank * 8 : (lane | 7), lane);
        const int src = bperm(lane & ~7, src0);  // every lane of new group s knows its source lane
        const int tipc = is_self ? tip : l;
        const int depc = is_self ? depth : depth + 1;
        const int meta = kind | ((tipc + 1) << 2) | (depc << 5);
        const int n_node = bperm(src, id);
        const float n_lp = bpermf(src, clp);
        const float n_gp = bpermf(src, cgp);
        const int n_meta = bperm(src, meta);
        int n_child = bperm(src + k, child);  // meaningful when the source is a self lane
        const int n_kind = n_meta & 3;
        const bool ngrp = i < Bn;
        if (n_kind == 1 || !is_child) n_child = -1;
        if (__ballot(ngrp && n_kind == 2 && is_child) != 0ull) {
            if (ngrp && n_kind == 2 && is_child) n_child = load_i32_l2(&rows[(int64_t)n_node * RW + l]);
        }
        const float top = readlane_f(n_lp + n_gp, 0);  // beam[0].probability() :278
        node = n_node;
        lp = n_lp / top;
        gp = n_gp / top;
        tip = ((n_meta >> 2) & 7) - 1;
        depth = n_meta >> 5;
        child = n_child;
        B = Bn;
    }

    // ---- walk the best labelling leaf -> root (:285-300) ----
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane == 0) {
        uint8_t *lab = p.out.labels + r * p.out.out_stride;
        uint32_t *pth = p.out.path ? p.out.path + r * p.out.out_stride : nullptr;
        int cur_node = node;
        for (int j = depth - 1; j >= 0 && cur_node >= 0; --j) {
            const int2 q = rec[cur_node];
            lab[j] = (uint8_t)((q.y & 7) + 1);
            if (pth) pth[j] = (uint32_t)(q.y >> 3);
            cur_node = q.x;
        }
        p.out.out_len[r] = (uint32_t)depth;
        p.out.status[r] = FCD_ST_OK;
    }
}

}  // namespace

bool beam_wave_supported(int beam_size, int N, int crf) {
    return !crf && beam_size >= 1 && beam_size <= 8 && N >= 3 && N <= 7;
}

hipError_t launch_beam_wave(const BatchDesc &in, int64_t read_begin, int64_t n_reads,
                            const BeamArgs &a, const WaveArena &arena, const ResultDesc &out,
                            hipStream_t stream) {
    if (n_reads <= 0) return hipSuccess;
    WaveParams p{in, a, arena, out, read_begin};
    p.in.n_reads = n_reads;  // reads in this launch
    const unsigned blocks = (unsigned)((n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
    const dim3 grid(blocks), block(64 * kWavesPerBlock);
    switch (in.N) {
        case 3: hipLaunchKernelGGL(beam_wave_kernel<3>, grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL(beam_wave_kernel<4>, grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL(beam_wave_kernel<5>, grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL(beam_wave_kernel<6>, grid, block, 0, stream, p); break;
        case 7: hipLaunchKernelGGL(beam_wave_kernel<7>, grid, block, 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fcd
