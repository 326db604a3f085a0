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
// node count when step t began (one 4-byte store per step).  A node's record is written when the node FIRST ENTERS
// THE BEAM, not when it is created (the traceback only ever walks through nodes that were beam entries).  A leaving
// node's child row is written only if the node can ever re-enter the beam (some beam entry is shallower; see
// beam_wave.hip).
#ifdef FCD_HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif

#include <map>
#include <mutex>
#include <utility>

#include "device_utils.h"
#include "fcd_internal.h"
#include "pdq178.h"
#ifdef FCD_LANE_TIE_PROF  // developer build (tools/dev/lane_tie_prof.sh): shader-clock stamps inside the tie-flagged step
#define FCD_WAVE_PROF 1
#endif
#include "pdq178_wave.h"
#include "slab_pool.h"

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
// the node-ordered list of ALL its candidates in LDS, the wavefront replays Rust 1.78's quicksort on it (pdq178_wave.h)
// and the ranks it produces replace the exact ones.
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
    // the wave-cooperative quicksort's position tables (pdq178_wave.h: a list of HALF * N candidates spans PL planes of
    // 64 positions), which first hold the ids of the candidates on older nodes while the list is put in node order
    constexpr int PL = PDQ ? (HALF * N + 63) / 64 : 1;
    union TieScratch {
        pdq178::WaveScratch<PL> ws;
        uint32_t eid[PDQ ? RPW * (HALF * N + 16) : 1];
    };
    __shared__ TieScratch s_tie_scr;

#ifdef FCD_HIPEMU  // (lockstep emulation: LDS arrives zeroed there, as garbage on the GPU -- make it garbage here too)
    if (threadIdx.x == 0) {
        memset(s_inc, 0xA5, sizeof(s_inc));
        memset(s_u, 0xA5, sizeof(s_u));
        memset(s_rank, 0xA5, sizeof(s_rank));
        memset(s_rec, 0xA5, sizeof(s_rec));
        memset(s_fate, 0xA5, sizeof(s_fate));
        memset(s_child, 0xA5, sizeof(s_child));
        memset(s_tie, 0xA5, sizeof(s_tie));
        memset(&s_tie_scr, 0xA5, sizeof(s_tie_scr));
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
    const bool retry = RPW == 1 && p.arena.retry_counter;
    if (retry) {
        // retry pass (capi.hip): the first pass ran in slabs sized for the usual tree; a read that outgrew its
        // slab was stopped with FCD_ST_INTERNAL and is decoded again here, in a slab that holds the worst case
        const bool mine = has_read && p.out.status[r] == FCD_ST_INTERNAL;  // wave-uniform: one read per wavefront
        if (!mine) return;
        if (lane == 0) atomicAdd(p.arena.retry_counter, 1);  // (how many did: it steers the sizing of later jobs)
    }
    // Arena addressing: a wave-uniform base (the slab of the wavefront's first read: scalar registers) plus a
    // 32-bit BYTE offset per lane (the second read's slab starts cap_nodes elements further on; a pair of slabs
    // stays below 4 GiB: cap_nodes < 2^23), so that no tree access needs 64-bit vector arithmetic.  Large jobs
    // take their (pair of) slab(s) from the device-side pool and hand it back after the traceback (slab_pool.h).
    const int cap = (int)p.arena.cap_nodes;
    const int pool_id = p.arena.pool ? slab_pool::pop(p.arena.pool, lane) : 0;
    const int64_t wslab = p.arena.pool ? (int64_t)pool_id * RPW : (int64_t)blockIdx.x * RPW;
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

    for (int t = 0; t < Tmax; ++t) {
        const bool act = alive && t < T;
        // ---- the posterior row: uniform over the read's lanes ----
        float pr[N];
#pragma unroll
        for (int c = 0; c < N; ++c)
            pr[c] = CRF ? rowv[c] : (RPW == 1 ? rdlanef(win[0], g * N + c) : bpermf(hbase + g * N + c, win[0]));
        const int g_row = g;            // (the FIFO may rotate below: the tip's column is fetched from this row later)
        const float win_row = win[0];
        if (!CRF && ++g == RPR) {
            g = 0;
#pragma unroll
            for (int j = 0; j + 1 < kFifo; ++j) win[j] = win[j + 1];
            __builtin_amdgcn_s_waitcnt(0x0F70);
            win[kFifo - 1] = incoming;
            blk = (t + 1) / RPR;  // (from the step counter: no count carried through the loop)
            incoming = load_block(blk + kFifo);
        }
        const bool ent = act && q < B;

        // ---- extensions by label l (:200-239); targets that are beam entries get the push ----
        s_inc[lane] = 0ull;
        wave_sync();
        float contrib[NL];
        bool cvalid[NL], merged[NL];
        int ccand[NL];  // candidate id of extension l (existing child or the new node)
        int n_new = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float pk = pr[l + 1];
            const bool pass = !(pk < thr);  // :201 skips only when pr_b < thr
            const bool rep = collapse && l == tip;
            contrib[l] = rep ? gp * pk : (lp + gp) * pk;
            const int ch = child[l];
            const bool exists = ch >= 0;
            cvalid[l] = ent && pass && (exists || !rep || gp > 0.0f);  // :212-218
            merged[l] = cvalid[l] && exists && (ch & kInBeam);
            if (merged[l])
                s_inc[hbase + ((ch >> kSlotShift) & kSlotMask)] = (1ull << 32) | (uint32_t)__float_as_int(contrib[l]);
            n_new += (cvalid[l] && !exists) ? 1 : 0;
        }
        wave_sync();

        // ---- the entry's own node: blank (:191-198) + repeat-stay (:206-211) + incoming extension ----
        const uint64_t iv = s_inc[lane];
        const bool has_inc = ent && (iv >> 32) != 0ull;
        const float inc = __int_as_float((int)(uint32_t)iv);
        // the tip label's column of the row: one more cross-lane fetch instead of a compare-and-select per label
        // (CRF: no repeat-stay, unused)
        float ptip = 0.0f;
        if (!CRF) ptip = bpermf(hbase + g_row * N + tip + 1, win_row);
        const float pr0 = pr[0];
        const bool blank = pr0 > thr;
        const float gpn = (lp + gp) * pr0;
        const bool stay = collapse && tip >= 0 && !(ptip < thr);
        const float lpn = lp * ptip;
        const float slp = (stay ? lpn : 0.0f) + (has_inc ? inc : 0.0f);
        const float sgp = blank ? gpn : 0.0f;
        const bool svalid = ent && (blank || stay || has_inc);

        // ---- tree.rs:125-145 add_node: ids in (beam order, label order) ----
        const int incl = half_prefix_add<RPW>(n_new);
        const int nn0 = nn;  // ids from here on are this step's new nodes
        if (q == 0 && act) *first_at(t) = nn;
        int next_id = nn + incl - n_new;
        nn += RPW == 1 ? rdlane(incl, 63) : bperm(hbase + HALF - 1, incl);
        const bool f_cap = act && nn > cap;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const bool is_new = cvalid[l] && child[l] < 0;
            // (the node's record -- parent, label -- is not written here: only a node that becomes a beam entry is ever
            // walked through by the traceback, and most of a step's new nodes never do; see the publish loop below)
            if (is_new && !f_cap) child[l] = next_id++;
            ccand[l] = child[l] & kIdMask;
        }

        // ---- search.rs:261-277 ----
        uint64_t key[N];
        bool cand_valid[N];
        cand_valid[0] = svalid;
        int n_valid = hcount(svalid);
        bool lane_nan = svalid && (slp + sgp) != (slp + sgp);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            cand_valid[l + 1] = cvalid[l] && !merged[l];
            n_valid += hcount(cand_valid[l + 1]);
            lane_nan = lane_nan || (cand_valid[l + 1] && contrib[l] != contrib[l]);
        }
        const bool any_nan = hcount(lane_nan) != 0;
        const bool f_nan = act && n_valid >= 2 && any_nan;
        const bool f_empty = act && n_valid == 0;
        if (f_nan || f_empty || f_cap) {
            if (q == 0) {
                p.out.status[r] = f_cap ? FCD_ST_INTERNAL : (f_empty ? FCD_ST_RAN_OUT_OF_BEAM : FCD_ST_INCOMPARABLE);
                p.out.out_len[r] = 0;
            }
            alive = false;
        }
        const bool go = act && alive;  // this read completes the step
        // (a NaN that gets this far is the lone candidate of the read: any non-zero key ranks it first)
        key[0] = (svalid && go) ? make_key(slp + sgp, node) : 0ull;
#pragma unroll
        for (int l = 0; l < NL; ++l) key[l + 1] = (cand_valid[l + 1] && go) ? make_key(contrib[l], ccand[l]) : 0ull;

        // ---- prune: the top beam_size candidates in exact key order ----
        const int Bn = go ? (n_valid < beam_size ? n_valid : beam_size) : 0;
        int rank[N];
#pragma unroll
        for (int k = 0; k < N; ++k) rank[k] = -1;
        const bool need_sel = go && n_valid > beam_size;
        int bstar = KB - 1;
        int Lc = go ? n_valid : 0;
        uint32_t mx = 0;
        bool tie = false, crit = false;  // AMB: the two tie conditions of include/fcd.h (fcd_result.ambiguous)
        if (ballot(need_sel) != 0ull) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const uint32_t hi = (uint32_t)(key[k] >> 32);
                mx = hi > mx ? hi : mx;
            }
            // maximum over the half: the DPP reduction of half_min_in_last_lane on the complement (keys are unsigned)
            mx = ~(uint32_t)bperm(hbase + HALF - 1, half_umin_in_last_lane<RPW>((int)~mx));
            {   // (zeros made HERE: hoisted out of the time loop, the four registers of a zero vector are the first thing
                // the PDQ instantiation spills at its 128-register budget -- and a scratch load per step is a memory wait)
                int z0 = 0;
                if (PDQ) FCD_OPAQUE_V(z0);
                *reinterpret_cast<int4 *>(hist + 4 * lane) = make_int4(z0, z0, z0, z0);
            }
            wave_sync();
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (need_sel && key[k] != 0ull) {
                    const uint32_t d = (mx - (uint32_t)(key[k] >> 32)) >> kBucketShift;
                    atomicAdd(&hist[4 * hbase + (d < (uint32_t)(KB - 1) ? d : (uint32_t)(KB - 1))], 1);
                }
            wave_sync();
            const int4 h = *reinterpret_cast<const int4 *>(hist + 4 * lane);
            const int s0 = h.x, s1 = s0 + h.y, s2 = s1 + h.z, s3 = s2 + h.w;
            const int hin = half_prefix_add<RPW>(s3);
            const int hex = hin - s3;
            // exactly one lane of a selecting half owns the bucket where the running count reaches beam_size
            const bool cross = need_sel && hex < beam_size && hin >= beam_size;
            const int kk = (hex + s0 >= beam_size) ? 0 : (hex + s1 >= beam_size) ? 1 : (hex + s2 >= beam_size) ? 2 : 3;
            const int cum = hex + (kk == 0 ? s0 : kk == 1 ? s1 : kk == 2 ? s2 : s3);
            const uint64_t m_cross = hmask(ballot(cross));
            const int owner = hbase + (m_cross ? __builtin_ctzll(m_cross) : 0);
            const int b_sel = bperm(owner, 4 * q + kk);
            const int l_sel = bperm(owner, cum);
            if (need_sel) {
                bstar = b_sel;
                Lc = l_sel;
            }
            wave_sync();
        }
        const bool listed = ballot(Lc > LCAP) == 0ull;
        if (listed) {
            // "bucket <= bstar" as one compare per candidate: bucket = min((mx - hi) >> shift, KB - 1) <= bstar  <=>
            // mx - hi < (bstar + 1) << shift (everything qualifies when bstar is the catch-all bucket)  <=>  hi >= lim
            const uint32_t span = (uint32_t)(bstar + 1) << kBucketShift;
            const uint32_t lim = (!need_sel || bstar >= KB - 1 || mx < span) ? 0u : mx - span + 1u;
            // the list in lane order (its order does not matter to the ranking): one prefix sum over the lanes'
            // counts instead of a vote and two population counts per candidate
            bool in[N];
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                in[k] = key[k] != 0ull && (uint32_t)(key[k] >> 32) >= lim;
                cnt += in[k] ? 1 : 0;
            }
            int pos = half_prefix_add<RPW>(cnt) - cnt;
            int src0 = lane * 8;
            if (PDQ) FCD_OPAQUE_V(src0);  // (made here, not kept in five registers across the loop: see the zeros above)
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (in[k]) {
                    l_key[pos] = key[k];
                    l_src[pos] = src0 + k;
                }
                pos += in[k] ? 1 : 0;
            }
            // the ranking loop runs to a wave-uniform bound, eight keys at a time: pad this read's list with zero keys
            int lmax = Lc;
            if (RPW == 2) lmax = max(lmax, __shfl_xor(lmax, 32));
            lmax = (__builtin_amdgcn_readfirstlane(lmax) + 7) & ~7;
            for (int z = Lc + q; z < lmax; z += HALF) l_key[z] = 0ull;
            {
                uint32_t m1 = ~0u;  // (same remark)
                if (PDQ) FCD_OPAQUE_V(m1);
                *reinterpret_cast<uint2 *>(s_rank + 8 * lane) = make_uint2(m1, m1);
            }
            wave_sync();
            for (int e0 = 0; e0 < lmax; e0 += HALF) {
                if (!AMB && e0 > 0 && lmax - e0 <= HALF / 4) {
                    // The list is usually a handful of entries longer than the half has lanes (beam_size survivors
                    // plus whatever shares the last bucket): a second pass of every lane over the whole list for
                    // their sake would double the ranking work.  Four lanes share each of them instead, each
                    // counting a quarter of the list, and add their counts up across the quad.
                    const int e = e0 + (q >> 2), sub = q & 3;
                    const uint64_t ke = e < Lc ? l_key[e] : ~0ull;
                    const int per = lmax >> 2;  // keys per lane: a quarter of the list (lmax is a multiple of 8)
                    int rk = 0, rk2 = 0;
                    for (int jj = 0; jj < per; jj += 2) {
                        const int j = sub * per + jj;
                        ulonglong2 kk2;
                        kk2.x = 0ull;
                        kk2.y = 0ull;
                        if (j < lmax) kk2 = *reinterpret_cast<const ulonglong2 *>(l_key + j);
                        rk += (kk2.x > ke) ? 1 : 0;
                        rk2 += (kk2.y > ke) ? 1 : 0;
                    }
                    rk += rk2;
                    rk += __builtin_amdgcn_update_dpp(0, rk, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
                    rk += __builtin_amdgcn_update_dpp(0, rk, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
                    if (sub == 0 && e < Lc && rk < beam_size) s_rank[l_src[e]] = (int8_t)rk;
                    if (PDQ && sub == 0 && e < Lc && rk <= beam_size) tie_tab[rk] = (uint32_t)(ke >> 32);
                    break;
                }
                const int e = e0 + q;
                const uint64_t ke = e < Lc ? l_key[e] : ~0ull;
                int rk = 0, rk2 = 0, n_eq = 0, n_gt = 0;
                if (!AMB) {
                    // eight keys per trip: four 16-byte LDS reads in flight, four independent compare-and-count chains
                    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
                    for (int j = 0; j < lmax; j += 8) {
                        const ulonglong2 ka = *reinterpret_cast<const ulonglong2 *>(l_key + j);
                        const ulonglong2 kb = *reinterpret_cast<const ulonglong2 *>(l_key + j + 2);
                        const ulonglong2 kc = *reinterpret_cast<const ulonglong2 *>(l_key + j + 4);
                        const ulonglong2 kd = *reinterpret_cast<const ulonglong2 *>(l_key + j + 6);
                        FCD_RANK4(ke, ka.x, ka.y, kb.x, kb.y, r0, r1, r2, r3);
                        FCD_RANK4(ke, kc.x, kc.y, kd.x, kd.y, r0, r1, r2, r3);
                    }
                    rk = (r0 + r1) + (r2 + r3);
                }
                for (int j = 0; AMB && j < lmax; j += 2) {  // two keys per 16-byte LDS read
                    const ulonglong2 kk2 = *reinterpret_cast<const ulonglong2 *>(l_key + j);
                    rk += (kk2.x > ke) ? 1 : 0;
                    rk2 += (kk2.y > ke) ? 1 : 0;
                    if (AMB) {  // equal probabilities share a bucket: every candidate tied with a kept one is listed
                        n_eq += (kk2.x != 0ull && (uint32_t)(kk2.x >> 32) == (uint32_t)(ke >> 32)) ? 1 : 0;
                        n_eq += (kk2.y != 0ull && (uint32_t)(kk2.y >> 32) == (uint32_t)(ke >> 32)) ? 1 : 0;
                        n_gt += ((uint32_t)(kk2.x >> 32) > (uint32_t)(ke >> 32)) ? 1 : 0;
                        n_gt += ((uint32_t)(kk2.y >> 32) > (uint32_t)(ke >> 32)) ? 1 : 0;
                    }
                }
                rk += rk2;
                if (e < Lc && rk < beam_size) s_rank[l_src[e]] = (int8_t)rk;
                if (PDQ && e < Lc && rk <= beam_size) tie_tab[rk] = (uint32_t)(ke >> 32);
                if (AMB && e < Lc) {
                    tie = tie || (rk < beam_size && n_valid > 20 && n_eq >= 2);
                    crit = crit || (n_eq >= 2 && (n_gt == 0 || (n_gt < beam_size && n_gt + n_eq > beam_size)));
                }
            }
            wave_sync();
            const uint64_t mine = *reinterpret_cast<const uint64_t *>(s_rank + 8 * lane);
#pragma unroll
            for (int k = 0; k < N; ++k) rank[k] = (int)(int8_t)(uint8_t)(mine >> (8 * k));
        } else {
            // heavily tied (or extremely spread) probabilities: rank every candidate against all of its read's
#pragma unroll
            for (int k = 0; k < N; ++k) c_key[lane * N + k] = key[k];
            wave_sync();
            int rk[N], n_eq[N], n_gt[N];
#pragma unroll
            for (int k = 0; k < N; ++k) rk[k] = n_eq[k] = n_gt[k] = 0;
            for (int j = 0; j < HALF * N; ++j) {
                const uint64_t kj = c_key[hbase * N + j];
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    rk[k] += (kj > key[k]) ? 1 : 0;
                    if (AMB) {
                        n_eq[k] += (kj != 0ull && (uint32_t)(kj >> 32) == (uint32_t)(key[k] >> 32)) ? 1 : 0;
                        n_gt[k] += ((uint32_t)(kj >> 32) > (uint32_t)(key[k] >> 32)) ? 1 : 0;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                rank[k] = (key[k] != 0ull && rk[k] < beam_size) ? rk[k] : -1;
                if (PDQ && key[k] != 0ull && rk[k] <= beam_size) tie_tab[rk[k]] = (uint32_t)(key[k] >> 32);
                if (AMB && key[k] != 0ull) {
                    tie = tie || (rank[k] >= 0 && n_valid > 20 && n_eq[k] >= 2);
                    crit = crit || (n_eq[k] >= 2 && (n_gt[k] == 0 || (n_gt[k] < beam_size && n_gt[k] + n_eq[k] > beam_size)));
                }
            }
            wave_sync();
        }

        if (AMB) {
            n_amb += hcount(tie) != 0 ? 1 : 0;
            n_crit += hcount(crit) != 0 ? 1 : 0;
        }
        if (PDQ) {
            // ranks q and q + 1 hold one probability, rank q is kept and rank q + 1 was ranked in this step (a
            // candidate outside the list shares no probability with a listed one): sort_unstable_by's order of the
            // two is pdqsort's business once the list is longer than 20 (:262)
            const int n_ranked = listed ? Lc : n_valid;
            const bool tied = go && n_valid > 20 && q < beam_size && q + 1 < n_ranked && tie_tab[q] == tie_tab[q + 1];
            const uint64_t m_tied = ballot(tied);
            if (__builtin_expect(m_tied != 0ull, 0)) {
                const bool mine_h = hmask(m_tied) != 0ull;
                // A read that ties once usually ties at most of its later steps, and each such step costs several plain
                // ones: its wavefront is one of the launch's stragglers.  Alone on its SIMD it runs an instruction per
                // ~4 cycles; sharing the SIMD with three wavefronts of the NEXT launch (another stream) it would get a
                // quarter of that and stretch its launch fourfold -- so from its first tied step on it keeps the
                // issue priority (a few dozen wavefronts per launch: nobody else notices).
                __builtin_amdgcn_s_setprio(3);
#ifdef FCD_LANE_TIE_PROF
                const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
                // Fourteen registers of loop-carried state sit the rare block out in LDS that is dead until the survivors
                // publish their records (s_rec, s_child, s_inc): the quicksort is inlined, and at the 128-register budget it
                // would otherwise push them to scratch memory.
                int *const park = reinterpret_cast<int *>(s_rec) + lane;
                int *const park2 = s_child + lane;
#pragma unroll
                for (int j = 0; j < kFifo; ++j) park[64 * j] = __float_as_int(win[j]);
                park[64 * 4] = __float_as_int(incoming);
                park[64 * 5] = jump;
                park[64 * 6] = __float_as_int(lp);
                park[64 * 7] = __float_as_int(gp);
#pragma unroll
                for (int l = 0; l < NL && l < RW; ++l) park2[64 * l] = child[l];
                int *const park3 = reinterpret_cast<int *>(s_inc) + lane;  // (dead until the next step's pushes)
                park3[0] = depth;
                park3[64] = tip;
                // The list sort_unstable_by is handed: the merged candidates in ascending node order (:245-260).  This
                // step's new nodes are numbered in candidate order and follow every older node, so only the candidates
                // on OLDER nodes need ranking -- against each other, four to a 16-byte LDS read.
                // (the keys, made again from what the step still holds: kept alive from the ranking to this rare block
                // they would cost the step's hot path ten registers -- and at the 128-register budget, spills)
                uint64_t key[N];
                key[0] = (svalid && go) ? make_key(slp + sgp, node) : 0ull;
#pragma unroll
                for (int l = 0; l < NL; ++l) key[l + 1] = (cand_valid[l + 1] && go) ? make_key(contrib[l], ccand[l]) : 0ull;
                // (each half's ids are followed by sixteen words of padding OF ITS OWN -- the lanes run in lockstep, a
                // padding store that reached into the other half's region would land after that half's ids)
                uint32_t *eid = s_tie_scr.eid + hh * (HALF * N + 16);
                bool older[N];
                int n_older = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    older[k] = mine_h && key[k] != 0ull && (k == 0 || ccand[k > 0 ? k - 1 : 0] < nn0);
                    n_older += older[k] ? 1 : 0;
                }
                const int o_incl = half_prefix_add<RPW>(n_older);
                const int n_old = RPW == 1 ? rdlane(o_incl, 63) : bperm(hbase + HALF - 1, o_incl);
                int pos[N];
#pragma unroll
                for (int k = 0; k < N; ++k) pos[k] = 0;
                // Rank of an older candidate's node among the older candidates' nodes.  Node ids are dense creation-order
                // integers and the candidates of a step sit on RECENT nodes (a beam entry is rarely more than a few dozen
                // steps old), so the ids span a small range: every older candidate sets ITS bit of a bitmap over that range
                // (ids are unique: one candidate per node), and a node's rank is the number of bits below its own -- a
                // per-word prefix count plus one masked population count.  ~100 instructions where comparing every
                // candidate with every older id took ~1000 (round 4; a straggler's wavefront pays per instruction).  A
                // range the table cannot hold (an entry that has sat in the beam for ~60 steps) takes the all-pairs path.
                constexpr int CAPW = (HALF * N + 16) / 2;  // words per table: the bitmap, and the per-word prefix counts behind it
                uint32_t idv[N];                            // node + 1: the root (-1) counts as 0
#pragma unroll
                for (int k = 0; k < N; ++k) idv[k] = (uint32_t)((k == 0 ? node : ccand[k > 0 ? k - 1 : 0]) + 1);
                uint32_t id_lo = ~0u, id_hi_c = ~0u;  // smallest id; complement of the largest (both by an unsigned minimum)
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if (older[k]) {
                        id_lo = idv[k] < id_lo ? idv[k] : id_lo;
                        id_hi_c = ~idv[k] < id_hi_c ? ~idv[k] : id_hi_c;
                    }
                id_lo = (uint32_t)bperm(hbase + HALF - 1, half_umin_in_last_lane<RPW>((int)id_lo));
                const uint32_t id_hi = ~(uint32_t)bperm(hbase + HALF - 1, half_umin_in_last_lane<RPW>((int)id_hi_c));
                const int W32 = n_old > 0 ? (int)((id_hi - id_lo) >> 5) + 1 : 0;  // bitmap words this read needs
                bool by_bitmap = ballot(mine_h && W32 > CAPW) == 0ull;   // (wave-uniform: both reads take the same path)
#ifdef FCD_HIPEMU  // (tests/test_emu_parity.py runs the tie-order tests once more on the all-pairs path)
                if (getenv("FCD_EMU_LANE_ALLPAIRS")) by_bitmap = false;
#endif
                if (by_bitmap) {
                    uint32_t *bm = eid, *pf = eid + CAPW;
                    for (int w = q; mine_h && w < W32; w += HALF) bm[w] = 0u;
                    wave_sync();
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (older[k]) {
                            const uint32_t d = idv[k] - id_lo;
                            atomicOr(&bm[d >> 5], 1u << (d & 31u));
                        }
                    wave_sync();
                    {   // exclusive count of the bits in the words below each word: lane q takes words [q * WPL, (q + 1) * WPL)
                        constexpr int WPL = (CAPW + HALF - 1) / HALF;
                        uint32_t wv[WPL];
                        int c = 0;
#pragma unroll
                        for (int u = 0; u < WPL; ++u) {
                            const int w = q * WPL + u;
                            wv[u] = (mine_h && w < W32) ? bm[w] : 0u;
                            c += __builtin_popcount(wv[u]);
                        }
                        int run = half_prefix_add<RPW>(c) - c;
#pragma unroll
                        for (int u = 0; u < WPL; ++u) {
                            const int w = q * WPL + u;
                            if (mine_h && w < W32) pf[w] = (uint32_t)run;
                            run += __builtin_popcount(wv[u]);
                        }
                    }
                    wave_sync();
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (older[k]) {
                            const uint32_t d = idv[k] - id_lo;
                            pos[k] = (int)pf[d >> 5] + __builtin_popcount(bm[d >> 5] & ((1u << (d & 31u)) - 1u));
                        }
                } else {
                    int o_pos = o_incl - n_older;
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (older[k]) eid[o_pos++] = (uint32_t)key[k];  // low word: larger = smaller node
                    if (q < 16) eid[n_old + q] = 0u;                    // (padding of the last block of reads: never "smaller")
                    wave_sync();
                    int n_old_max = n_old;
                    if (RPW == 2) n_old_max = max(n_old_max, __shfl_xor(n_old_max, 32));
                    n_old_max = __builtin_amdgcn_readfirstlane(n_old_max);
                    // sixteen ids per trip: four 16-byte LDS reads in flight together
#pragma unroll 1
                    for (int j = 0; j < n_old_max; j += 16) {
                        uint4 e[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            e[u] = make_uint4(0u, 0u, 0u, 0u);
                            if (j + 4 * u < n_old) e[u] = *reinterpret_cast<const uint4 *>(eid + j + 4 * u);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
#pragma unroll
                            for (int k = 0; k < N; ++k) {
                                const uint32_t me = (uint32_t)key[k];
                                pos[k] += ((e[u].x > me) ? 1 : 0) + ((e[u].y > me) ? 1 : 0) + ((e[u].z > me) ? 1 : 0) + ((e[u].w > me) ? 1 : 0);
                            }
                        }
                    }
                }
                wave_sync();  // (the table the ids sat in is the sort's scratch from here on)
                uint64_t *list = c_key + hbase * N;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    if (!mine_h || key[k] == 0ull) continue;
                    const int at = older[k] ? pos[k] : n_old + (ccand[k > 0 ? k - 1 : 0] - nn0);
                    list[at] = (key[k] & 0xFFFFFFFF00000000ull) | (uint32_t)(lane * 8 + k);
                }
                if (mine_h) *reinterpret_cast<uint64_t *>(s_rank + 8 * lane) = ~0ull;
                wave_sync();
#ifdef FCD_LANE_TIE_PROF
                const unsigned long long tp1 = __builtin_amdgcn_s_memtime();
#endif
                // all 64 lanes replay the quicksort on a flagged read's list, one read after the other (pdq178_wave.h:
                // at the tail of a launch a wavefront has ONE chronically tied read)
#pragma unroll 1
                for (int hs = 0; hs < RPW; ++hs) {
                    const bool flagged = RPW == 1 || (hs ? (m_tied >> 32) != 0ull : (m_tied & 0xFFFFFFFFull) != 0ull);
                    if (!flagged) continue;
                    pdq178::wave_sort_inline<PL>(c_key + hs * HALF * N, rdlane(n_valid, hs * HALF), beam_size, &s_tie_scr.ws, lane);
                }
#ifdef FCD_LANE_TIE_PROF
                const unsigned long long tp2 = __builtin_amdgcn_s_memtime();
#endif
                for (int j = q; mine_h && j < Bn; j += HALF) s_rank[(int)(uint32_t)list[j]] = (int8_t)j;
                wave_sync();
                const uint64_t again = *reinterpret_cast<const uint64_t *>(s_rank + 8 * lane);
#pragma unroll
                for (int k = 0; k < N; ++k)
                    if (mine_h) rank[k] = (int)(int8_t)(uint8_t)(again >> (8 * k));
#pragma unroll
                for (int j = 0; j < kFifo; ++j) win[j] = __int_as_float(park[64 * j]);
                incoming = __int_as_float(park[64 * 4]);
                jump = park[64 * 5];
                lp = __int_as_float(park[64 * 6]);
                gp = __int_as_float(park[64 * 7]);
#pragma unroll
                for (int l = 0; l < NL && l < RW; ++l) child[l] = park2[64 * l];
                depth = park3[0];
                tip = park3[64];
                wave_sync();
#ifdef FCD_LANE_TIE_PROF
                if (lane == 0) {
                    const unsigned long long tp3 = __builtin_amdgcn_s_memtime();
                    atomicAdd(&pdq178::g_wave_prof[12], tp1 - tp0);  // the node-ordered list
                    atomicAdd(&pdq178::g_wave_prof[13], tp2 - tp1);  // the replay
                    atomicAdd(&pdq178::g_wave_prof[14], tp3 - tp2);  // ranks handed back
                    atomicAdd(&pdq178::g_wave_prof[15], 1ull);       // tied steps
                }
#endif
            }
        }

        // ---- survivors publish their records in rank order; old slot i says where its own candidate went ----
        // kInBeam == 64 << kSlotShift: the published value, shifted, IS the (slot, IN-BEAM) field of a child entry
        static_assert(kInBeam == (64 << kSlotShift) && kSlotMask == 63, "child-entry bit layout");
        s_fate[lane] = rank[0] >= 0 ? (rank[0] | 64) : 0;
        // a record is {label prob, gap prob, node, meta | jump, source lane, -, CRF state}: written dword by dword
        // behind an offset the compiler cannot see through, so that the stores pair up from whatever registers hold
        // the values (ds_write2_b32) instead of being moved into four consecutive ones for a 16-byte store
        int *const recw = reinterpret_cast<int *>(s_rec);
        auto publish = [&](int slot, int a, int b, int c, int d, int e, int f) {
            int off = 8 * slot;
            FCD_OPAQUE_V(off);
            recw[off + 0] = a;
            recw[off + 1] = b;
            recw[off + 2] = c;
            recw[off + 3] = d;
            recw[off + 4] = e;
            recw[off + 5] = lane;
            if (CRF) recw[off + 7] = f;
        };
        if (rank[0] >= 0) {  // the entry's own node stays in the beam
            const int meta = 0 | ((tip + 1) << 2) | (depth << 5);
            publish(hbase + rank[0], __float_as_int(slp), __float_as_int(sgp), node, meta, jump, state);
        }
        const int jumpc = (depth % kSeg == 0) ? node : jump;  // a child's nearest segment head
        const int metac = 1 | ((depth + 1) << 5);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int rk = rank[l + 1];
            if (rk >= 0) {  // the child by label l enters the beam
                // kind 1, or 2 when it has been there before (EVER: its row is in HBM)
                const int ever = (int)(((uint32_t)child[l] >> 30) & 1u);
                const int meta = (metac + ever) | ((l + 1) << 2);
                if (!ever) {
                    // first time in the beam: NOW its record exists in the arena -- (parent, label), and for a segment
                    // head where the next head up the tree is.  The best labelling is a beam entry, every node was
                    // created as the child of one, so every node on the way to the root has been through here; the
                    // 5 new nodes in 6 that are pruned at once cost the arena no write (43 -> 7 records per step at
                    // beam 32).  (Under the default tie order a tied step publishes once, after its ranks are final.)
                    *rec_at(ccand[l]) = ((node + 1) << 3) | l;
                    if ((depth + 1) % kSeg == 0) *jmp_at(ccand[l]) = jumpc;
                }
                publish(hbase + rk, __float_as_int(contrib[l]), 0, ccand[l], meta, jumpc,
                        CRF ? ((state * NL) & s_mask) + l : 0);  // :97
            }
        }
        wave_sync();

        // ---- child entries: follow a beam entry to its new slot, mark entering children, evict rows ----
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            int ch = child[l];
            if (ent && go && ch >= 0) {
                if (ch & kInBeam) {
                    int *const fp = &s_fate[hbase + ((ch >> kSlotShift) & kSlotMask)];
                    const int fate = *fp;
                    ch = (ch & kStored) | (fate << kSlotShift);
                    // the child leaves the beam while this entry -- its parent, the only reader of that slot --
                    // stays: tell the child's lane (see the row eviction below)
                    if (fate == 0 && rank[0] >= 0) *fp = kParentStays;
                } else if (rank[l + 1] >= 0) {
                    ch = (ch & kIdMask) | kEver | kInBeam | (rank[l + 1] << kSlotShift);
                }
            }
            child[l] = ch;
        }
        // word l of this entry's child row as it is stored: the entry as it is (its beam-position bits are stale by
        // the time the row is read back and are stripped THERE, on the rare path), or -1
        int row_word[RW];
#pragma unroll
        for (int l = 0; l < RW; ++l) row_word[l] = -1;
#pragma unroll
        for (int l = 0; l < NL; ++l) row_word[l] = child[l];
#pragma unroll
        for (int l = 0; l < RW; ++l) s_child[lane * RW + l] = -1;
#pragma unroll
        for (int l = 0; l < NL; ++l) s_child[lane * RW + l] = child[l];
        wave_sync();

        // ---- the new beam: lane r of the half takes record r ----
        const int me = hbase + (q < Bn ? q : 0);
        const int4 ra = s_rec[2 * me], rb = s_rec[2 * me + 1];
        const int2 r0 = *reinterpret_cast<const int2 *>(&s_rec[2 * hbase]);
        const int n_node = ra.z;
        const int n_meta = ra.w;
        const int n_kind = n_meta & 3;
        const int src = rb.y & 63;
        int n_child[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) n_child[l] = n_kind == 0 ? s_child[src * RW + l] : -1;
        const bool reload = q < Bn && n_kind == 2;
        const uint64_t m_reload = ballot(reload);
        // A node re-enters the beam only as the extension of its parent, so it needs a proper ancestor in the beam:
        // once every beam entry is at least as deep as the node none is one, and none ever will be (the minimum
        // depth of the beam never decreases).  Such a node's row is dead and is not written -- most evicted rows.
        int mind = 0x7FFFFFFF;  // smallest depth among the survivors this entry contributes ...
        if (rank[0] >= 0) mind = depth;
#pragma unroll
        for (int l = 0; l < NL; ++l)
            if (rank[l + 1] >= 0) mind = min(mind, depth + 1);
        mind = bperm(hbase + HALF - 1, half_min_in_last_lane<RPW>(mind));  // ... and in the whole new beam
        // One level further the test is still exact: a node exactly one deeper than the shallowest beam entry can
        // only come back through its PARENT (any other ancestor is shallower than every beam entry).  A parent that
        // stays in the beam has just said so (above); a parent that is only now coming back itself has not, so a
        // step in which any node re-enters the beam keeps the plain depth test.
        const bool any_reent = hmask(m_reload) != 0ull;
        const bool parent_stays = s_fate[lane] == kParentStays;
        const bool dead = depth <= mind || (depth == mind + 1 && !parent_stays && !any_reent);
        if (ent && go && rank[0] < 0 && node >= 0 && !dead) {
            // this node leaves the beam and may come back: its child row has to exist in HBM from now on
            int4 *row = reinterpret_cast<int4 *>(row_at(node));
            row[0] = make_int4(row_word[0], row_word[1], row_word[2], row_word[3]);
            if (RW == 8) row[1] = make_int4(row_word[RW - 4], row_word[RW - 3], row_word[RW - 2], row_word[RW - 1]);
        }
        if (m_reload != 0ull) {
            // a node that was in the beam before comes back: its row is in HBM, and which of its
            // children are beam entries right now has to be looked up
            int e[NL], eid[NL], eslot[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                e[l] = reload ? load_i32_l2(reinterpret_cast<int32_t *>(row_at(n_node)) + l) : -1;
                if (e[l] >= 0) e[l] &= kStored;  // (written with whatever beam-position bits the entry had at eviction)
#ifdef FCD_HIPEMU  // (lockstep emulation: device memory arrives poisoned with 0xA5)
                if (reload && e[l] == (int)0xA5A5A5A5) {  // never written: the dead-row test was wrong
                    fprintf(stderr, "beam_lane: node %d re-entered the beam but its child row was never stored\n", n_node);
                    abort();
                }
#endif
                // only a child that has been a beam entry (EVER) can be one now
                eid[l] = (e[l] >= 0 && (e[l] & kEver)) ? (e[l] & kIdMask) : -2;
                eslot[l] = -1;
            }
            // few lanes reload in a step: take them one at a time and let their read's lanes look for each
            // of their children among the new beam's nodes (one compare + ballot per child)
            for (uint64_t m_rel = ballot(reload); m_rel != 0ull; m_rel &= m_rel - 1ull) {
                const int L = __builtin_ctzll(m_rel);
                const int Lb = L & ~(HALF - 1);
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int id = rdlane(eid[l], L);
                    if (id >= 0) {
                        uint64_t m_hit = ballot(q < Bn && n_node == id);
                        if (RPW == 2) m_hit = Lb ? (m_hit >> 32) : (m_hit & 0xFFFFFFFFull);
                        if (m_hit != 0ull && lane == L) eslot[l] = __builtin_ctzll(m_hit);
                    }
                }
            }
#pragma unroll
            for (int l = 0; l < NL; ++l)
                if (reload)
                    n_child[l] = eslot[l] >= 0 ? ((e[l] & kStored) | kInBeam | (eslot[l] << kSlotShift)) : e[l];
        }
        if (CRF) {
            if (q < Bn) state = rb.w;
#pragma unroll
            for (int c = 0; c < N; ++c)   // in flight during the divisions below
                rowv[c] = t + 1 < T ? load_post(post, (int64_t)(t + 1) * st_t + (int64_t)state * st_s + c * st_n, dt) : 0.0f;
        }
        const float top = __int_as_float(r0.x) + __int_as_float(r0.y);  // beam[0].probability() :278
        if (q < Bn) {
            node = n_node;
            lp = __int_as_float(ra.x) / top;
            gp = __int_as_float(ra.y) / top;
            tip = ((n_meta >> 2) & 7) - 1;
            depth = n_meta >> 5;
            jump = rb.x;
#pragma unroll
            for (int l = 0; l < NL; ++l) child[l] = n_child[l];
        }
        if (go) B = Bn;
        wave_sync();
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
    if (p.arena.pool) slab_pool::push(p.arena.pool, pool_id, lane);
}

template <int N, bool AMB, bool CRF, bool PDQ>
hipError_t launch_nap(const LaneParams &p, int64_t n_reads, hipStream_t stream) {
    if (beam_lane_reads_per_wave(p.a.beam_size) == 2 && !p.arena.retry_counter) {  // two reads per wavefront
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

// developer instrument: the cycle counters of a -DFCD_LANE_TIE_PROF build (zeros otherwise)
hipError_t lane_tie_prof_read(unsigned long long *out16, bool reset) {
#if defined(FCD_LANE_TIE_PROF) && !defined(FCD_HIPEMU)
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(pdq178::g_wave_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    if (reset) {
        unsigned long long zero[16] = {0};
        return hipMemcpyToSymbol(HIP_SYMBOL(pdq178::g_wave_prof), zero, sizeof(zero));
    }
    return hipSuccess;
#else
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    return hipSuccess;
#endif
}

// ---- the device-side slab pool (slab_pool.h): how many wavefronts of an instantiation the chip holds, and the ring's set-up
namespace {
template <int N, bool AMB, bool CRF, bool PDQ>
const void *kernel_nap(int rpw) {
    return rpw == 2 ? reinterpret_cast<const void *>(&beam_lane_kernel<N, 2, AMB, CRF, PDQ>)
                    : reinterpret_cast<const void *>(&beam_lane_kernel<N, 1, AMB, CRF, PDQ>);
}
template <int N, bool CRF>
const void *kernel_n(int rpw, bool amb, bool pdq) {
    if (amb) return pdq ? kernel_nap<N, true, CRF, true>(rpw) : kernel_nap<N, true, CRF, false>(rpw);
    return pdq ? kernel_nap<N, false, CRF, true>(rpw) : kernel_nap<N, false, CRF, false>(rpw);
}
__global__ void slab_pool_init_kernel(unsigned long long *pool, int slabs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        pool[0] = 0ull;                          // pop tickets
        pool[1] = (unsigned long long)slabs;     // push tickets: the queue starts full
        pool[2] = (unsigned long long)slabs;
    }
    if (i < slabs)  // slab i sits in cell i as if push ticket i had stored it: sequence i + 1
        pool[slab_pool::kHeaderWords + i] = ((unsigned long long)(uint32_t)(i + 1) << 32) | (uint32_t)i;
}
}  // namespace

int beam_lane_reads_per_wave(int beam_size) { return beam_size <= 32 ? 2 : 1; }

int beam_lane_resident_waves(int beam_size, int N, int crf, bool first_pass, bool ambiguous, int tie_order) {
#ifdef FCD_HIPEMU
    (void)beam_size, (void)N, (void)crf, (void)first_pass, (void)ambiguous, (void)tie_order;
    return 3;  // (blocks run one after the other there: a small pool makes every later block reuse a slab)
#else
    const int rpw = first_pass ? beam_lane_reads_per_wave(beam_size) : 1;
    const bool pdq = tie_order == FCD_TIE_PDQ178;
    const void *k = nullptr;
    if (crf) k = kernel_n<5, true>(rpw, ambiguous, pdq);
    else switch (N) {
        case 2: k = kernel_n<2, false>(rpw, ambiguous, pdq); break;
        case 3: k = kernel_n<3, false>(rpw, ambiguous, pdq); break;
        case 4: k = kernel_n<4, false>(rpw, ambiguous, pdq); break;
        case 5: k = kernel_n<5, false>(rpw, ambiguous, pdq); break;
        case 6: k = kernel_n<6, false>(rpw, ambiguous, pdq); break;
        case 7: k = kernel_n<7, false>(rpw, ambiguous, pdq); break;
        case 8: k = kernel_n<8, false>(rpw, ambiguous, pdq); break;
    }
    int dev = 0;
    if (!k || hipGetDevice(&dev) != hipSuccess) return 256 * 4 * 8;  // (every wave slot of the chip)
    // (asked once per instantiation and device: the answer does not change, and every wide-beam call needs it)
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> known;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(k, dev);
    const auto it = known.find(key);
    if (it != known.end()) return it->second;
    int cus = 0, per_cu = 0;
    int waves = 256 * 4 * 8;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64, 0) == hipSuccess && per_cu >= 1 && cus >= 1)
        waves = cus * per_cu;
    known[key] = waves;
    return waves;
#endif
}

hipError_t slab_pool_init(unsigned long long *pool, int slabs, hipStream_t stream) {
    if (slabs < 1 || slabs > slab_pool::kMaxSlabs) return hipErrorInvalidValue;
    hipLaunchKernelGGL(slab_pool_init_kernel, dim3((unsigned)((slabs + 255) / 256)), dim3(256), 0, stream, pool, slabs);
    return hipGetLastError();
}

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

// this translation unit's copy of the replay's std-form word (pdq178.h), on the current device
FCD_PDQ178_DEFINE_STD_FORM_SETTER(beam_lane_set_pdq178_std_form)

}  // namespace fcd
