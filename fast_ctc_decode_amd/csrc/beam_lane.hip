// beam_lane.hip -- CTC prefix beam search (search::beam_search, /root/reference/src/search.rs:159-301)
// for WIDE beams: one beam ENTRY per lane, N <= 8; one read per wavefront with beam_size <= 64, or TWO
// reads per wavefront (one per 32-lane half) with beam_size <= 32.
//
// beam_wave.hip gives every candidate slot its own lane, which stops at 12 beam entries x 5 lanes;
// beam_generic.hip keeps the beam in LDS and spends ~2500 instructions per step at beam 32.  Here lane i
// holds beam entry i (rank order) with its whole state in registers -- node, label/gap probability,
// tip, depth, jump pointer and the node's child entries -- and evaluates the entry's own candidate
// and its N-1 extensions itself, so the expansion needs no cross-lane traffic at all:
//   * the posterior row is wave-uniform: v_readlane out of a register FIFO filled by coalesced loads;
//   * an extension whose target is already a beam entry (IN-BEAM/slot bits in the child entry, as in
//     beam_wave.hip) is dropped into that entry's slot of a 64-entry LDS table -- a node has one
//     parent, so at most one write per slot -- and picked up by the target lane (:245-260: at most
//     two non-zero f32 addends meet, so the fold order is immaterial);
//   * new nodes are numbered in the reference's creation order (beam order x label order) with a
//     wave prefix sum of the per-lane counts;
//   * prune: a 256-bucket histogram over "distance below the step's maximum probability" finds the
//     bucket of the beam_size-th largest candidate; the candidates in that bucket or above are
//     compacted into a list and ranked exactly on the 64-bit key (probability desc, node asc); ranks
//     travel back to the owning lanes through a byte table; heavily tied steps fall back to all-pairs;
//   * survivors publish their record in rank order (LDS), lane r picks up record r, the child entries
//     follow through a second table, a re-entering node re-reads its row from HBM.
// Tree arena and the segment-parallel leaf -> root walk are beam_wave.hip's, with dense node ids: a record is
// (parent, label) and the creation time of a node -- what `path` reports -- is looked up in first[t], the read's
// node count when step t began (one 4-byte store per step).  A leaving node's child row is written only if the
// node can ever re-enter the beam (some beam entry is shallower; see beam_wave.hip).
#ifdef FCD_HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif
#include "device_utils.h"
#include "fcd_internal.h"
#include "pdq178.h"
#include "pdq178_coop.h"

namespace fcd {

namespace {

// child entry: node id in bits 0..22, beam slot in bits 23..28, IN-BEAM bit 29, EVER bit 30
constexpr int kEver = 1 << 30;
constexpr int kInBeam = 1 << 29;
constexpr int kSlotShift = 23;
constexpr int kSlotMask = 63;
constexpr int kIdMask = (1 << 23) - 1;
constexpr int kStored = kEver | kIdMask;
constexpr int kParentStays = 1 << 20;  // s_fate marker: "your node leaves the beam, its parent stays" (no slot field looks like it)

constexpr int kSeg = 64;        // nodes per traceback segment
constexpr int kFifo = 4;        // registers in the row FIFO
constexpr int kBuckets = 256;   // prune pre-selection histogram
constexpr int kBucketShift = 18;
constexpr int kListCap = 128;

struct LaneParams {
    BatchDesc in;
    BeamArgs a;
    WaveArena arena;
    ResultDesc out;
    int64_t read_begin;
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_or_zero(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, BANK_MASK, false);
}

// inclusive prefix sum over the 64 lanes (the classic GCN DPP scan; see envelope.hip) -- or, without the
// last step, over each 32-lane half separately
template <int RPW>
__device__ __forceinline__ int half_prefix_add(int x) {
    int t = x + dpp_or_zero<0x111, 0xf, 0xf>(x);
    t += dpp_or_zero<0x112, 0xf, 0xf>(x);
    t += dpp_or_zero<0x113, 0xf, 0xf>(x);
    t += dpp_or_zero<0x114, 0xf, 0xe>(t);
    t += dpp_or_zero<0x118, 0xf, 0xc>(t);
    t += dpp_or_zero<0x142, 0xa, 0xf>(t);
    if (RPW == 1) t += dpp_or_zero<0x143, 0xc, 0xf>(t);
    return t;
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_or_self(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, BANK_MASK, false);
}

// minimum over the 64 lanes -- or over each 32-lane half -- delivered in the LAST lane of the wavefront / half
template <int RPW>
__device__ __forceinline__ int half_min_in_last_lane(int x) {
    int t = min(x, dpp_or_self<0x111, 0xf, 0xf>(x));  // row_shr:1
    t = min(t, dpp_or_self<0x112, 0xf, 0xf>(t));      // row_shr:2
    t = min(t, dpp_or_self<0x114, 0xf, 0xf>(t));      // row_shr:4
    t = min(t, dpp_or_self<0x118, 0xf, 0xf>(t));      // row_shr:8: lane 15 of every row holds the row's minimum
    t = min(t, dpp_or_self<0x142, 0xa, 0xf>(t));      // row_bcast:15 into rows 1 and 3
    if (RPW == 1) t = min(t, dpp_or_self<0x143, 0xc, 0xf>(t));  // row_bcast:31 into rows 2 and 3
    return t;
}

// the same for unsigned values
template <int RPW>
__device__ __forceinline__ int half_umin_in_last_lane(int x) {
    auto umin = [](int a, int b) { return (int)min((uint32_t)a, (uint32_t)b); };
    int t = umin(x, dpp_or_self<0x111, 0xf, 0xf>(x));
    t = umin(t, dpp_or_self<0x112, 0xf, 0xf>(t));
    t = umin(t, dpp_or_self<0x114, 0xf, 0xf>(t));
    t = umin(t, dpp_or_self<0x118, 0xf, 0xf>(t));
    t = umin(t, dpp_or_self<0x142, 0xa, 0xf>(t));
    if (RPW == 1) t = umin(t, dpp_or_self<0x143, 0xc, 0xf>(t));
    return t;
}

__device__ __forceinline__ int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float bpermf(int src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rdlanef(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// AMB: the same search plus the tie instrument of SURVEY.md 8a A4 (fcd_result.ambiguous, see beam_wave.hip).
// CRF: search::crf_beam_search (:38-157) with a run-time state count S (a power of two >= 4, N = 5, as in
// beam_wave.hip's gather mode): every entry reads the row of ITS state, probs[t, state, :] -- N values per
// lane, requested for step t+1 as soon as the entry's next state is known; no repeat-stay; an extension
// moves to state (state * 4) & (S - 1) + label (:97), which cannot leave the table.
// PDQ: FCD_TIE_PDQ178 (include/fcd.h; see beam_wave.hip) -- the candidates of rank <= beam_size leave their probability
// in a table by rank; when a kept candidate ties with its successor among more than 20 candidates, the half builds
// the node-ordered list of ALL its candidates in LDS, one lane replays Rust 1.78's quicksort on it (pdq178.h) and the
// ranks it produces replace the exact ones.
template <int N, int RPW, bool AMB, bool CRF, bool PDQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void beam_lane_kernel(LaneParams p) {
    constexpr int NL = N - 1;
    constexpr int HALF = 64 / RPW;          // lanes (= beam slots) per read
    constexpr int RPR = HALF / N;           // rows per FIFO register
    constexpr int RW = NL <= 4 ? 4 : 8;     // child-row width in the arena
    constexpr int KB = kBuckets / RPW;      // histogram buckets per read (four per lane)
    constexpr int LCAP = kListCap / RPW;    // list capacity per read
    static_assert(N >= 2 && N <= 8, "candidate ids are lane * 8 + k");

    __shared__ uint64_t s_inc[64];                        // {flag, contribution} pushed to a beam slot
    __shared__ __attribute__((aligned(16))) int4 s_u[64 * N / 2 > 160 ? 64 * N / 2 : 160];  // hist+list | all keys
    __shared__ __attribute__((aligned(8))) int8_t s_rank[64 * 8];  // rank of candidate (lane, k), -1 = out
    __shared__ __attribute__((aligned(16))) int4 s_rec[64 * 2];     // survivor records by (half, rank)
    __shared__ int s_fate[64];   // where old slot i's own candidate went: new slot | 64 (the IN-BEAM bit of the field), or 0
    __shared__ __attribute__((aligned(16))) int s_child[64 * RW];   // child entries of old slot i
    __shared__ int s_heads[64];
    __shared__ uint32_t s_tie[PDQ ? 2 * 66 : 2];  // probability (orderable bits) of the candidate of rank r <= beam_size
    // the wave-cooperative quicksort's tables (pdq178_coop.h); with them the N = 5 instantiation holds 10 KB of LDS:
    // sixteen wavefronts per CU still fit
    __shared__ pdq178::CoopScratch<PDQ ? N : 1> s_coop;

#ifdef FCD_HIPEMU  // (lockstep emulation: LDS arrives zeroed there, as garbage on the GPU -- make it garbage here too)
    if (threadIdx.x == 0) {
        memset(s_inc, 0xA5, sizeof(s_inc));
        memset(s_u, 0xA5, sizeof(s_u));
        memset(s_rank, 0xA5, sizeof(s_rank));
        memset(s_rec, 0xA5, sizeof(s_rec));
        memset(s_fate, 0xA5, sizeof(s_fate));
        memset(s_child, 0xA5, sizeof(s_child));
        memset(s_tie, 0xA5, sizeof(s_tie));
        memset(&s_coop, 0xA5, sizeof(s_coop));
    }
    __syncthreads();
#endif
    int *hist = reinterpret_cast<int *>(s_u);                        // 256 ints      (1 KB)
    uint64_t *l_key_all = reinterpret_cast<uint64_t *>(s_u + 64);    // 128 u64      (1 KB)
    int *l_src_all = reinterpret_cast<int *>(s_u + 128);             // 128 ints     (0.5 KB)
    uint64_t *c_key = reinterpret_cast<uint64_t *>(s_u);             // 64 * N u64, fallback only

    const int lane = threadIdx.x;
    const int q = lane & (HALF - 1);
    const int hbase = lane - q;
    const int hh = lane / HALF;
    uint64_t *l_key = l_key_all + hh * LCAP;
    int *l_src = l_src_all + hh * LCAP;
    uint32_t *tie_tab = s_tie + (PDQ ? hh * 66 : 0);
    const int64_t local = (int64_t)blockIdx.x * RPW + hh;
    const bool has_read = local < p.in.n_reads;  // n_reads here = reads in this launch
    const int64_t r = p.read_begin + (has_read ? local : 0);
    const int beam_size = p.a.beam_size;
    const bool collapse = !CRF && p.a.collapse != 0;
    const float thr = p.a.thr;

    int T = 0;
    if (has_read) {
        int64_t t64 = p.in.T;
        if (p.in.lengths) {
            const int64_t tl = p.in.lengths[r];
            t64 = tl < 0 ? 0 : (tl < t64 ? tl : t64);
        }
        T = (int)t64;
    }
    int Tmax = T;
    if (RPW == 2) Tmax = max(Tmax, __shfl_xor(Tmax, 32));
    Tmax = __builtin_amdgcn_readfirstlane(Tmax);
    const int dt = p.in.dtype;
    const float *post = post_at(p.in.post, r * p.in.stride_read, dt);
    const int64_t st_t = p.in.stride_t, st_n = p.in.stride_n, st_s = p.in.stride_s;
    int64_t slab = has_read ? local : 0;
    if (RPW == 1 && p.arena.retry_counter) {
        // retry pass (capi.hip): the first pass ran in slabs sized for the usual tree; a read that outgrew its
        // slab was stopped with FCD_ST_INTERNAL and is decoded again here, in a slab that holds the worst case
        const bool mine = has_read && p.out.status[r] == FCD_ST_INTERNAL;  // wave-uniform: one read per wavefront
        int slot = -1;
        if (mine && lane == 0) slot = atomicAdd(p.arena.retry_counter, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (!mine || slot >= p.arena.retry_slots) return;  // (left over: the host runs another round)
        slab = slot;
    }
    // Arena addressing: a wave-uniform base (the slab of the wavefront's first read: scalar registers) plus a
    // 32-bit BYTE offset per lane (the second read's slab starts cap_nodes elements further on; a pair of slabs
    // stays below 4 GiB: cap_nodes < 2^23), so that no tree access needs 64-bit vector arithmetic.
    const int cap = (int)p.arena.cap_nodes;
    const int64_t wslab = (RPW == 1 && p.arena.retry_counter) ? slab : (int64_t)blockIdx.x * RPW;
    char *const rec_w = reinterpret_cast<char *>(p.arena.rec + wslab * p.arena.cap_nodes);
    char *const jmp_w = reinterpret_cast<char *>(p.arena.jmp + wslab * p.arena.cap_nodes);
    char *const rows_w = reinterpret_cast<char *>(p.arena.rows + wslab * p.arena.cap_nodes * RW);
    const uint32_t hoff = (RPW == 2 && has_read && hh) ? (uint32_t)cap : 0u;  // this half's slab, in nodes
    auto rec_at = [&](int id) -> int32_t * { return reinterpret_cast<int32_t *>(rec_w + ((hoff + (uint32_t)id) << 2)); };
    // first[t] = the read's node count when step t began: ids are dense and in creation order, so node h was
    // created in the last step t with first[t] <= h -- which is what `path` reports.  One 4-byte store per step
    // instead of a time word in every record (43 per step at beam 32).
    char *const first_w = reinterpret_cast<char *>(p.arena.first + wslab * p.arena.first_stride);
    const uint32_t foff = (RPW == 2 && has_read && hh) ? (uint32_t)p.arena.first_stride : 0u;
    auto first_at = [&](int t) -> int32_t * { return reinterpret_cast<int32_t *>(first_w + ((foff + (uint32_t)t) << 2)); };
    auto jmp_at = [&](int id) -> int32_t * { return reinterpret_cast<int32_t *>(jmp_w + ((hoff + (uint32_t)id) << 2)); };
    auto row_at = [&](int id) -> char * { return rows_w + (hoff + (uint32_t)id) * (uint32_t)(RW * 4); };

    // votes and counts over this lane's half
    auto hmask = [&](uint64_t m) -> uint64_t { return RPW == 1 ? m : (hh ? (m >> 32) : (m & 0xFFFFFFFFull)); };
    auto hcount = [&](bool pred) -> int { return popc64(hmask(ballot(pred))); };

    // ---- beam state: lane q of a half = beam entry q (search.rs:170-175: root, label_prob 0, gap_prob 1) ----
    int node = -1;
    float lp = 0.0f, gp = 1.0f;
    int tip = -1;
    int depth = 0;
    int jump = -1;
    int child[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) child[l] = -1;
    int B = 1;
    int nn = 0;
    bool alive = has_read;
    int n_amb = 0, n_crit = 0;
    int state = 0;
    const int s_mask = CRF ? (int)p.in.S - 1 : 0;
    if (CRF && has_read) {
        // search.rs:54-59: state = argmax(init), label_prob = max(init), gap_prob = init[0];
        // ndarray-stats: first maximum wins, NaN -> Err -> unwrap() panics
        const float *init = p.a.init + r * p.a.init_stride;
        float m = init[0];
        bool bad = m != m;
        for (int64_t j = 1; j < p.a.n_init; ++j) {
            const float e = init[j];
            bad = bad || (e != e);
            if (e > m) {
                m = e;
                state = (int)j;
            }
        }
        lp = m;
        gp = init[0];
        if ((bad || state > s_mask) && T > 0) {
            if (q == 0) {
                p.out.status[r] = FCD_ST_BAD_STATE;
                p.out.out_len[r] = 0;
            }
            alive = false;
            state = 0;
        }
    }

    // ---- row FIFO: register j holds rows [blk*RPR, blk*RPR + RPR) of block (front + j) ----
    const int fg = q / N, fc = q - fg * N;
    const bool f_lane = q < RPR * N;
    auto load_block = [&](int blk) -> float {
        const int row = blk * RPR + fg;
        return (f_lane && row < T) ? load_post(post, (int64_t)row * st_t + fc * st_n, dt) : 0.0f;
    };
    float win[kFifo];
#pragma unroll
    for (int j = 0; j < kFifo; ++j) win[j] = CRF ? 0.0f : load_block(j);
    float incoming = CRF ? 0.0f : load_block(kFifo);
    int g = 0, blk = 0;
    // CRF: the row of this entry's state, requested one step ahead
    float rowv[N];
#pragma unroll
    for (int c = 0; c < N; ++c)
        rowv[c] = (CRF && T > 0) ? load_post(post, (int64_t)state * st_s + c * st_n, dt) : 0.0f;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see beam_wave.hip

    {
        // Two copies of the step at two loop depths (beam_lane_step.inc): the time loop proper holds no cold path.
        int t = 0;
        bool act = false;
        float pr[N];
        int g_row = 0;
        float win_row = 0.0f;
        while (t < Tmax) {
            bool again = false;
            for (; t < Tmax; ++t) {
                act = alive && t < T;
                // ---- the posterior row: uniform over the read's lanes ----
#pragma unroll
                for (int c = 0; c < N; ++c)
                    pr[c] = CRF ? rowv[c] : (RPW == 1 ? rdlanef(win[0], g * N + c) : bpermf(hbase + g * N + c, win[0]));
                g_row = g;            // (the FIFO may rotate below: the tip's column is fetched from this row later)
                win_row = win[0];
                if (!CRF && ++g == RPR) {
                    g = 0;
#pragma unroll
                    for (int j = 0; j + 1 < kFifo; ++j) win[j] = win[j + 1];
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    win[kFifo - 1] = incoming;
                    ++blk;
                    incoming = load_block(blk + kFifo);
                }
#define FCD_STEP_SLOW 0
#include "beam_lane_step.inc"
#undef FCD_STEP_SLOW
            }
            if (PDQ && again) {  // (act, pr[], g_row, win_row: still this step's)
                do {
#define FCD_STEP_SLOW 1
#include "beam_lane_step.inc"
#undef FCD_STEP_SLOW
                } while (false);
                ++t;
            }
        }
    }

    // ---- walk the best labelling leaf -> root (:285-300), segment-parallel (see beam_wave.hip) ----
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    uint8_t *lab = p.out.labels + r * p.out.out_stride;
    uint32_t *pth = p.out.path ? p.out.path + r * p.out.out_stride : nullptr;
    if (q == 0 && alive) {
        p.out.out_len[r] = (uint32_t)depth;
        p.out.status[r] = FCD_ST_OK;
    }
    if (AMB && q == 0 && has_read) {
        p.out.ambiguous[2 * r] = (uint32_t)n_amb;
        p.out.ambiguous[2 * r + 1] = (uint32_t)n_crit;
    }
    int h0 = bperm(hbase, node);
    int d0 = bperm(hbase, alive ? depth : 0);
    const int j0 = bperm(hbase, jump);
    while (ballot(d0 > 0) != 0ull) {
        int cnt = 0, nh = h0, nd = d0;
        if (q == 0) {
            while (cnt < HALF && nd > 0) {
                s_heads[hbase + cnt] = nh;
                nh = (nd % kSeg != 0) ? j0 : *jmp_at(nh);
                nd = ((nd - 1) / kSeg) * kSeg;
                ++cnt;
            }
        }
        cnt = bperm(hbase, cnt);
        nh = bperm(hbase, nh);
        nd = bperm(hbase, nd);
        wave_sync();
        if (q < cnt) {
            const int d1 = ((d0 - 1) / kSeg) * kSeg;
            const int ds = q == 0 ? d0 : d1 - (q - 1) * kSeg;
            const int de = q == 0 ? d1 : ds - kSeg;
            int h = s_heads[hbase + q];
            int dd = ds;
            // creation step of the segment's first node: the last t with first[t] <= h (binary search); every
            // further node on the way up was created strictly earlier, usually a step or two: scan backwards
            int tc = 0;
            if (h >= 0) {
                int lo = 0, hi = T - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (*first_at(mid) <= h) lo = mid;
                    else hi = mid - 1;
                }
                tc = lo;
            }
            int fc = h >= 0 ? *first_at(tc) : 0;        // first[tc], kept in a register
            auto time_of = [&](int id) -> uint32_t {  // tc: creation step of the node visited before (or of `id`)
                while (fc > id) fc = *first_at(--tc);   // first[0] = 0 <= id: terminates
                return (uint32_t)tc;
            };
            auto one = [&]() {  // emit position dd - 1, step to the parent
                const int e = *rec_at(h);
                lab[dd - 1] = (uint8_t)((e & 7) + 1);
                if (pth) pth[dd - 1] = time_of(h);
                h = (e >> 3) - 1;
                --dd;
            };
            // aligned rows: four positions leave as one 4-byte label store and one 16-byte path store (beam_wave.hip)
            const bool wide = (reinterpret_cast<uintptr_t>(lab) & 3) == 0 && (!pth || (reinterpret_cast<uintptr_t>(pth) & 15) == 0);
            while (dd > de && h >= 0 && (!wide || (dd & 3) != 0)) one();
            for (; dd - 4 >= de && h >= 0; dd -= 4) {
                uint32_t lw = 0;
                uint32_t tw[4];
#pragma unroll
                for (int j = 3; j >= 0; --j) {  // positions dd-1 (j = 3) ... dd-4 (j = 0)
                    const int e = *rec_at(h);
                    lw |= (uint32_t)((e & 7) + 1) << (8 * j);
                    tw[j] = pth ? time_of(h) : 0u;
                    h = (e >> 3) - 1;
                }
                *reinterpret_cast<uint32_t *>(lab + dd - 4) = lw;
                if (pth) *reinterpret_cast<uint4 *>(pth + dd - 4) = make_uint4(tw[0], tw[1], tw[2], tw[3]);
            }
            while (dd > de && h >= 0) one();
        }
        __builtin_amdgcn_wave_barrier();
        h0 = nh;
        d0 = nd;
    }
}

template <int N, bool AMB, bool CRF, bool PDQ>
hipError_t launch_nap(const LaneParams &p, int64_t n_reads, hipStream_t stream) {
    if (p.a.beam_size <= 32 && !p.arena.retry_counter) {  // two reads per wavefront
        hipLaunchKernelGGL((beam_lane_kernel<N, 2, AMB, CRF, PDQ>), dim3((unsigned)((n_reads + 1) / 2)), dim3(64), 0, stream, p);
    } else {
        hipLaunchKernelGGL((beam_lane_kernel<N, 1, AMB, CRF, PDQ>), dim3((unsigned)n_reads), dim3(64), 0, stream, p);
    }
    return hipGetLastError();
}

template <int N, bool AMB, bool CRF>
hipError_t launch_na(const LaneParams &p, int64_t n_reads, hipStream_t stream) {
    return p.a.tie_order == FCD_TIE_PDQ178 ? launch_nap<N, AMB, CRF, true>(p, n_reads, stream)
                                           : launch_nap<N, AMB, CRF, false>(p, n_reads, stream);
}

template <int N>
hipError_t launch_n(const LaneParams &p, int64_t n_reads, hipStream_t stream) {
    return p.out.ambiguous ? launch_na<N, true, false>(p, n_reads, stream) : launch_na<N, false, false>(p, n_reads, stream);
}

}  // namespace

bool beam_lane_supported(int beam_size, int N, int crf, int S) {
    if (beam_size < 1 || beam_size > 64) return false;
    if (crf) return N == 5 && S >= 4 && (S & (S - 1)) == 0;  // 5 symbols x 2^k states
    return N >= 2 && N <= 8;
}

hipError_t launch_beam_lane(const BatchDesc &in, int64_t read_begin, int64_t n_reads, const BeamArgs &a,
                            const WaveArena &arena, const ResultDesc &out, hipStream_t stream) {
    if (n_reads <= 0) return hipSuccess;
    LaneParams p{in, a, arena, out, read_begin};
    p.in.n_reads = n_reads;
    if (a.crf) {
        if (!beam_lane_supported(a.beam_size, in.N, 1, in.S)) return hipErrorInvalidValue;
        return out.ambiguous ? launch_na<5, true, true>(p, n_reads, stream) : launch_na<5, false, true>(p, n_reads, stream);
    }
    switch (in.N) {
        case 2: return launch_n<2>(p, n_reads, stream);
        case 3: return launch_n<3>(p, n_reads, stream);
        case 4: return launch_n<4>(p, n_reads, stream);
        case 5: return launch_n<5>(p, n_reads, stream);
        case 6: return launch_n<6>(p, n_reads, stream);
        case 7: return launch_n<7>(p, n_reads, stream);
        case 8: return launch_n<8>(p, n_reads, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace fcd
