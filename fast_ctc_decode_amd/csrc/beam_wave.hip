// beam_wave.hip -- register-resident CTC prefix beam search for search::beam_search
// (/root/reference/src/search.rs:159-301).
//
// Three instantiation families of one kernel template <N, GW, RPW>:
//   RPW = 2, GW = 6 : TWO reads per wavefront (one per 32-lane half), beam_size <= 5, N <= 5
//                     -- the BASELINE headline shape (beam 5, N = 5);
//   RPW = 1, GW = 8 : one read per wavefront, beam_size <= 8, N <= 7;
//   RPW = 1, GW = 5 : one read per wavefront, beam_size 9..12, N <= 5 (twelve groups of five lanes).
//
// Lane map inside a half: q = GW*i + k.  i = beam slot (rank order, like the reference's sorted
// Vec), k = 0 the slot's own node, k = 1..NL the child reached by label k-1, and -- when the group
// has a lane to spare -- k = GW-1 a scratch target for cross-lane pushes (GW = 5, N = 5 uses the
// four lanes past the last group instead).  Every lane of group i carries slot i's (node, label_prob,
// gap_prob, tip label, depth); lane (i, k>=1) also carries node i's child entry for label k-1.
// The search state lives in VGPRs; LDS is only the cross-lane network (ds_permute/ds_bpermute)
// and a 64-entry sort-key table per wave.
//
// Per timestep:
//   * posterior row: a register FIFO holds the next kFifo*RPR rows of each half's read, element
//     (row g, column c) of the front register sitting in lane g*N+c, so the values a lane needs (its own
//     column -- the blank on a slot's own lane -- and the tip's) are two ds_bpermutes, issued one step
//     AHEAD; one coalesced load per RPR steps refills the FIFO ~40 rows ahead: no per-step memory access;
//   * child lanes evaluate the extension (:200-239).  A child entry carries IN-BEAM/slot bits, so
//     "is my target already a beam entry, and where?" costs no search: the extension is pushed to
//     that slot's lane 0 (ds_permute), which adds it to its own blank/stay terms -- the
//     reference's sort-by-node + fold (:245-260), order-independent because at most two non-zero
//     f32 addends ever meet;
//   * new tree nodes get ids in the reference's creation order via ballot + prefix popcount;
//   * pruning ranks the candidates exactly on a 64-bit key (probability desc, node asc);
//   * survivors publish (lane, depth) by rank in a small LDS table: one round trip later every lane knows
//     its source lane, where the best candidate sits and the minimum depth of the new beam; the survivors are
//     gathered into rank order with ds_bpermute and divided by the top probability (:278-282, IEEE f32
//     division: one per lane, the two quotients shared inside each group).
//
// Node ids are (time step << KS) | index among the nodes created in that step: creation order, as the reference's
// tie-break needs, with the creation time -- what `path` reports -- readable off the id itself.
// Tree arena (HBM, per read): rec[node] = (parent + 1) << 3 | label (4 bytes); rows[node] = the node's child
// entries (id | EVER, or -1), written when the node is evicted from the beam -- unless no beam entry is
// shallower, in which case the node can never come back and its row is dead --; jmp[node]
// (written only for nodes whose depth is a multiple of 64) = the nearest proper ancestor whose depth
// is a multiple of 64.  Every beam entry carries its own jump pointer in a register, so the final
// leaf -> root walk (:285-300) first hops along jump pointers to cut the labelling into 64-node
// segments and then walks all segments in parallel, one lane each, instead of chasing ~2000
// dependent pointers with a single lane.  EVER marks children that have themselves been in the
// beam: only those can own children, so only their row is re-read when they re-enter the beam
// (3.9 % of steps on BASELINE's generator) -- everything else stays in registers.
#ifdef FCD_HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif
#include <type_traits>

#include "device_utils.h"
#include "fcd_internal.h"
#include "pdq178.h"
#include "pdq178_wave.h"

namespace fcd {

namespace {

// child entry: node id in bits 0..24, beam slot in bits 25..28, IN-BEAM bit 29, EVER bit 30
constexpr int kEver = 1 << 30;
constexpr int kInBeam = 1 << 29;
constexpr int kSlotShift = 25;
constexpr int kSlotMask = 15;
constexpr int kIdMask = (1 << 25) - 1;
constexpr int kStored = kEver | kIdMask;  // what goes to HBM

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
// element `idx` of a wave-uniform array through a 32-bit BYTE offset (slabs stay below 4 GiB: capi.hip), so
// that the access is base (scalar registers) + offset (one vector register) with no 64-bit vector arithmetic
__device__ __forceinline__ int32_t *at32(int32_t *base, uint32_t idx) {
    return reinterpret_cast<int32_t *>(reinterpret_cast<char *>(base) + (idx << 2));
}
__device__ __forceinline__ int perm(int dst_lane, int v) {
    return __builtin_amdgcn_ds_permute(dst_lane << 2, v);
}

constexpr int kWavesPerBlock = 4;
constexpr int kCrfGather = -1;
constexpr int kFifo = 8;  // registers in the row FIFO
constexpr int kSeg = 64;  // nodes per traceback segment (jump-pointer spacing)

// S == 0: search::beam_search.  S > 0: search::crf_beam_search (:38-157) with S transition states:
// the row is probs[t, state, :] of the entry's state, there is no repeat-stay, and an extension
// moves to state (state * n_base) % n_state + label (:97).
// S == kCrfGather: crf_beam_search with a RUN-TIME state count (a power of two >= 4, N = 5: real
// basecaller heads have 4^k states, far too many rows for the register FIFO).  Only the rows of visited
// states are read: every lane fetches the one value it needs, probs[t+1, state', its column], as soon as
// its slot's next state is known -- before the renormalising divisions of step t -- and uses it at the top
// of step t+1 (4 B per lane, five 20-byte rows per read and step).  (state * 4) & (S - 1) + label < S
// always holds for such S, so no transition can leave the table.
//
// AMB: the same search plus the tie instrument of SURVEY.md 8a A4 (fcd_result.ambiguous, two counters per
// read; semantics in include/fcd.h).  A separate instantiation: the timed kernel pays nothing.
// PROF: the same search with a cycle stamp after each block of the step (profiles/): every
// stamp waits for the block's results, so the blocks' DEPENDENT latencies are measured, not their overlap.
// UNI: every read of the launch has the same length (no `lengths` array).  A read then stops taking part only by
// FAILING (its status is already written and its traceback skipped), so nothing has to keep its beam intact:
// the per-step "this half still runs" guards disappear -- a failed read is left with an EMPTY beam (B = 0), which
// makes every later step a no-op for it, and a half without a read starts that way.
// H16: the posteriors may be a 16-bit type (fcd_batch.dtype, converted exactly on load).  A separate instantiation:
// with the element type a run-time value the float32 kernels paid for the test on every FIFO refill -- every step
// in the CRF shape, whose rows fill a whole FIFO register (config 4: 4.2 -> 5.3 ms).
// PDQ: FCD_TIE_PDQ178 (include/fcd.h) -- equal probabilities among more than 20 candidates come out in the order
// Rust 1.78's sort_unstable_by leaves them in (:122,262).  Every step still takes its exact rank on (probability
// desc, node asc); the candidates of rank <= beam_size also leave their probability in a small table by rank, so
// one compare per slot says whether a KEPT candidate ties with its successor.  Only then (a few steps per thousand
// reads on the BASELINE generator) the half builds the node-ordered candidate list in LDS, the wavefront replays the
// quicksort on it (pdq178_wave.h) and the ranks it produces replace the exact ones.  Instantiated only for shapes
// that can hold more than 20 candidates.
template <int N, int GW, int RPW, int S, bool AMB, bool PROF = false, bool UNI = false, bool H16 = false, bool PDQ = false>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu((RPW == 2 && !AMB) ? 4 : 1))) void beam_wave_kernel(WaveParams p) {
    constexpr bool CRF = S != 0;
    constexpr bool GATHER = S == kCrfGather;
    constexpr int NL = N - 1;
    constexpr int HALF = 64 / RPW;
    constexpr int BCAP = HALF / GW;     // beam slots per read
    constexpr int E = (S > 0 ? S : 1) * N; // posterior values per timestep (register FIFO)
    constexpr int RPR = HALF / E;       // rows per FIFO register
    static_assert(RPR >= 1, "one timestep must fit the lanes of a half");
    constexpr int RW = NL <= 4 ? 4 : 8; // child-row width in the arena
    constexpr int KS = RPW == 2 ? 5 : 6;  // id slots per time step = 1 << KS >= lanes of a half >= BCAP * NL (beam_wave_id_shift)
    static_assert((1 << KS) >= HALF && (1 << KS) >= BCAP * NL, "a step's new nodes must fit its block of ids");
    // ds_permute pushes that have nothing to say need a harmless target: the group's spare lane when it
    // has one (N < GW), otherwise one of the lanes past the last group
    static_assert(N <= GW, "a group holds the node's own slot and one lane per label");
    constexpr bool HAS_SCRATCH = N <= GW - 1;
    constexpr int NIDLE = HALF - BCAP * GW;
    static_assert(HAS_SCRATCH || NIDLE >= 1, "no lane left to absorb idle pushes");
    __shared__ uint64_t s_keys[kWavesPerBlock][64];
    __shared__ int s_heads[kWavesPerBlock][64];
    // survivor table, per half: entry r = (byte address of the lane whose candidate took rank r) | that candidate's depth << 8
    // (PDQ: one entry per candidate rank, and as many tie words behind them -- per half)
    __shared__ int s_srcs[kWavesPerBlock][PDQ ? 128 : RPW * 16];
    constexpr int kTie = HALF;
    // PDQ: the node-ordered candidate list and the quicksort's tables of a tie-flagged step
    __shared__ uint64_t s_list[PDQ ? kWavesPerBlock : 1][64];
    __shared__ pdq178::WaveScratch<1> s_ws[PDQ ? kWavesPerBlock : 1];
    static_assert(!PDQ || BCAP * N > 20, "the tie order only matters above 20 candidates");
    static_assert(!PDQ || BCAP * N <= HALF - 2, "the last two entries of a half's table are never written (i_src below)");
    int n_amb = 0, n_crit = 0;
    uint32_t cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t cyc_last = 0;
    auto stamp_i = [&](int slot, int &dep) {
        if (PROF) {
            uint64_t now;
            FCD_STAMP(now, dep);
            cyc[slot] += (uint32_t)now - cyc_last;
            cyc_last = (uint32_t)now;
        }
    };
    auto stamp_f = [&](int slot, float &dep) {
        if (PROF) {
            uint64_t now;
            FCD_STAMP(now, dep);
            cyc[slot] += (uint32_t)now - cyc_last;
            cyc_last = (uint32_t)now;
        }
    };

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane & (HALF - 1);
    const int hbase = lane - q;
    const int i = q / GW, k = q - i * GW;
    const bool idle = i >= BCAP;             // e.g. lanes 30,31 of a half when GW = 6
    const bool is_self = !idle && k == 0;
    const bool is_child = !idle && k >= 1 && k <= NL;
    const int l = k - 1;
    const int grp0 = hbase + i * GW;         // lane 0 of my group
    const int dummy = HAS_SCRATCH ? (idle ? lane : grp0 + GW - 1) : hbase + BCAP * GW + (q % (NIDLE > 0 ? NIDLE : 1));
    const int beam_size = p.a.beam_size;
    const bool collapse = !CRF && p.a.collapse != 0;
    const float thr = p.a.thr;
    uint64_t *keys = s_keys[wave];
    int *srcs = s_srcs[wave] + (hbase ? (PDQ ? 2 * HALF : 16) : 0);
    // smallest candidate count from which "rank i ties with rank i + 1" is the quicksort's business: more than 20
    // candidates, rank i kept, rank i + 1 present (never, for a lane outside the beam's groups)
    int tie_lim = (!idle && i < p.a.beam_size) ? (i + 1 > 20 ? i + 1 : 20) : 0x7FFFFFFF;
    if (PDQ) FCD_OPAQUE_V(tie_lim);  // (ONE compare per step: the optimiser would take the folded limit apart again)
    // where a lane looks its group's source up: groups past the beam read an entry nobody ever writes (lane 0, kind 0)
    // -- the entries behind the kept ranks belong to dropped candidates, and a dropped candidate is often a node that
    // has been in the beam before: read as a source it would vote "re-entering" in every step
    const int i_src = (!idle && i < p.a.beam_size) ? i : HALF - 2;
    if (PDQ) {  // tie words: pairwise different, bit 31 clear -- no candidate's probability word looks like that
#pragma unroll
        for (int w = lane; w < 128; w += 64) s_srcs[wave][w] = ((w / HALF) & 1) ? (w % HALF) : 0x7FFFFF00;
    } else if (lane < RPW * 16) {
        s_srcs[wave][lane] = 0x7FFFFF00;  // never the minimum depth; points at lane 0
    }

    const int64_t local = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * RPW + (lane / HALF);
    const bool has_read = local < p.in.n_reads;  // n_reads here = reads in this launch
    const int64_t r = p.read_begin + (has_read ? local : 0);

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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o));
    Tmax = __builtin_amdgcn_readfirstlane(Tmax);

    const int dt = H16 ? p.in.dtype : (int)kF32;  // element type of the posteriors (f16 / bf16: converted exactly on load)
    // The posteriors of a wavefront's reads: the first read's address is wave-uniform (scalar registers) and the second
    // read sits one read stride further on -- formed where a block is loaded (once per FIFO rotation), not carried
    // through the time loop as a 64-bit pointer per lane.
    const int64_t r_first = p.read_begin + ((int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(wave)) * RPW;
    const float *const post_w = post_at(p.in.post, r_first * p.in.stride_read, dt);
    const int64_t half_stride = (RPW == 2 && lane >= HALF) ? p.in.stride_read : 0;
    const int64_t st_t = p.in.stride_t, st_n = p.in.stride_n, st_s = p.in.stride_s;
    // Arena addressing: a wave-uniform base (the slab of the wavefront's first read: scalar registers) plus a
    // 32-bit element offset per lane (the second read's slab starts cap_nodes elements further on), so that
    // every tree access is base + offset without 64-bit vector arithmetic.  Child rows are indexed by
    // node + 1: the root (node -1) owns row 0 and needs no special case.
    const int cap = (int)p.arena.cap_nodes;
    const int64_t wslab = ((int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(wave)) * RPW;
    int32_t *const rec_w = p.arena.rec + wslab * p.arena.cap_nodes;
    int32_t *const jmp_w = p.arena.jmp + wslab * p.arena.cap_nodes;
    int32_t *const rows_w = p.arena.rows + wslab * p.arena.cap_nodes * RW;
    const uint32_t hoff = has_read ? (uint32_t)(lane / HALF) * (uint32_t)cap : 0u;  // this half's slab, in nodes
    // (the traceback's per-lane pointers are formed AFTER the time loop: four registers the loop need not carry)

    // ---- beam state (search.rs:170-175: root, label_prob 0, gap_prob 1) ----
    int node = -1;
    float lp = 0.0f, gp = 1.0f;
    int tipf = 0;  // (tip label + 1) << 2, 0 for the root: the form meta carries, and the byte offset of the tip's column
    int depth = 0;
    int jump = -1;  // nearest proper ancestor of `node` at a depth that is a multiple of kSeg
    int child = -1;
    int B = (UNI && !has_read) ? 0 : 1;
    bool alive = has_read;
    int state = 0;
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
        if ((bad || state >= (GATHER ? p.in.S : S)) && T > 0) {
            if (q == 0) {
                p.out.status[r] = FCD_ST_BAD_STATE;
                p.out.out_len[r] = 0;
            }
            alive = false;
            state = 0;
        }
    }

    // ---- row FIFO: register j holds rows [blk*RPR, blk*RPR+RPR) of block (front + j) ----
    // this lane's (row-in-block, state, column) as a FIFO element
    const int fg = q / E, fe = q - fg * E, fs = fe / N, fc = fe - fs * N;
    const bool f_lane = q < RPR * E;
    auto load_block = [&](int blk) -> float {
        const int row = blk * RPR + fg;
        return (f_lane && row < T) ? load_post(post_w, half_stride + (int64_t)row * st_t + fs * st_s + fc * st_n, dt) : 0.0f;
    };
    float win[kFifo];
#pragma unroll
    for (int j = 0; j < kFifo; ++j) win[j] = GATHER ? 0.0f : load_block(j);
    // the block in flight joins the FIFO one rotation after its load was launched
    float incoming = GATHER ? 0.0f : load_block(kFifo);
    int g = 0;    // row within the front block (wave-uniform)
    int gE = 0;   // g * E, carried along instead of multiplied out
    int blk = 0;  // index of the front block (wave-uniform)
    // GATHER: the one value of row tt this lane needs -- column 0 (blank) on a slot's own lane, the
    // label's column on a child lane -- in the row of the slot's current state
    const int s_mask = GATHER ? (int)p.in.S - 1 : 0;
    const int kcol = is_child ? k : 0;
    auto gather_row = [&](int tt) -> float {
        return tt < T ? load_post(post_w, half_stride + (int64_t)tt * st_t + (int64_t)state * st_s + kcol * st_n, dt) : 0.0f;
    };
    float rowv = GATHER ? gather_row(0) : 0.0f;
    // Drain the prologue loads HERE, with a wait the compiler's scoreboard sees: otherwise the loop
    // header inherits "win[0] may still be in flight" and gets an s_waitcnt vmcnt(0) on EVERY step,
    // which on gfx9-family counters also waits for the previous step's tree stores to be acked.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched

    if (PROF) {  // first stamp: start of the loop
        uint64_t now;
        FCD_STAMP(now, lp);
        cyc_last = (uint32_t)now;
    }
    // The row values a lane needs -- column 0 (the blank) on a slot's own lane, the label's column on a child
    // lane, and the tip's column -- are fetched one step AHEAD, as soon as the next beam's tips are known: the two
    // ds_bpermutes then travel under the divisions instead of heading the next step's dependent chain.
    auto fetch_row = [&](float &o_pk, float &o_ptip) {
        const int rbase = hbase + gE + (S > 0 ? state * N : 0);
        o_pk = bpermf(rbase + (is_child ? k : 0), win[0]);
        o_ptip = CRF ? 0.0f : __int_as_float(__builtin_amdgcn_ds_bpermute((rbase << 2) + tipf, __float_as_int(win[0])));
    };
    float pk_next = 0.0f, ptip_next = 0.0f;
    if (!GATHER) fetch_row(pk_next, ptip_next);
    // the row FIFO moves on to step t + 1 (the values step t needs were fetched at the end of step t - 1)
    auto advance_fifo = [&]() __attribute__((always_inline)) {
        gE += E;
        if (!GATHER && ++g == RPR) {
            g = 0;
            gE = 0;
#pragma unroll
            for (int j = 0; j + 1 < kFifo; ++j) win[j] = win[j + 1];
            // wait for the block launched one rotation ago BEFORE launching the next one (vmcnt
            // counts in order: a wait placed after the new load would wait for it as well)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            win[kFifo - 1] = incoming;
            ++blk;
            incoming = load_block(blk + kFifo);
        }
    };
    {
        for (int t = 0; t < Tmax; ++t) {
            advance_fifo();
#include "beam_wave_step.inc"
        }
    }
    if (PROF && lane == 0 && p.a.prof) {
        uint32_t *o = p.a.prof + ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 8;
#pragma unroll
        for (int j = 0; j < 7; ++j) o[j] = cyc[j];
        o[7] = (uint32_t)Tmax;
    }

    // ---- walk the best labelling leaf -> root (:285-300), segment-parallel ----
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int32_t *rec = rec_w + hoff;
    const int32_t *jmp = jmp_w + hoff;
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
    int *heads = s_heads[wave];
    // beam[0] lives in group 0: every lane of the half takes ITS leaf and depth
    int h0 = bperm(hbase, node);               // current chunk's first segment head
    int d0 = bperm(hbase, alive ? depth : 0);  // its depth; 0 == nothing left
    int j0 = bperm(hbase, jump);               // the leaf's own jump pointer (from its beam entry)
    while (ballot(d0 > 0) != 0ull) {
        // phase 1: one lane hops along the jump pointers, collecting up to HALF segment heads
        int cnt = 0, nh = h0, nd = d0;
        if (q == 0) {
            while (cnt < HALF && nd > 0) {
                heads[hbase + cnt] = nh;
                // the leaf may sit at any depth and brings its jump pointer along; every later head
                // sits at a multiple of 64 and has it stored
                nh = (nd % kSeg != 0) ? j0 : jmp[nh];
                nd = ((nd - 1) / kSeg) * kSeg;
                ++cnt;
            }
        }
        cnt = bperm(hbase, cnt);
        nh = bperm(hbase, nh);
        nd = bperm(hbase, nd);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // phase 2: lane q walks segment q: depths (d_q, d_{q+1}]
        if (q < cnt) {
            const int d1 = ((d0 - 1) / kSeg) * kSeg;
            const int ds = q == 0 ? d0 : d1 - (q - 1) * kSeg;
            const int de = q == 0 ? d1 : ds - kSeg;
            int h = heads[hbase + q];
            int dd = ds;
            auto one = [&]() {  // emit position dd - 1, step to the parent
                const int e = rec[h];
                lab[dd - 1] = (uint8_t)((e & 7) + 1);
                if (pth) pth[dd - 1] = (uint32_t)(h >> KS);  // the step that created the node
                h = (e >> 3) - 1;
                --dd;
            };
            // A lane's 64 byte-sized label stores (and 64 path words), each to a line of its own, cost more than
            // the pointer chase: when the rows are aligned, four positions are collected and leave as one 4-byte and
            // one 16-byte store.  Only the leaf's segment can start off a multiple of four.
            const bool wide = (reinterpret_cast<uintptr_t>(lab) & 3) == 0 && (!pth || (reinterpret_cast<uintptr_t>(pth) & 15) == 0);
            while (dd > de && h >= 0 && (!wide || (dd & 3) != 0)) one();
            for (; dd - 4 >= de && h >= 0; dd -= 4) {  // (de is a multiple of 64: whole groups down to the segment's end)
                uint32_t lw = 0;
                uint32_t tw[4];
#pragma unroll
                for (int j = 3; j >= 0; --j) {  // positions dd-1 (j = 3) ... dd-4 (j = 0)
                    const int e = rec[h];
                    lw |= (uint32_t)((e & 7) + 1) << (8 * j);
                    tw[j] = (uint32_t)(h >> KS);
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

template <int N, int GW, int RPW, int S, bool PDQ>
hipError_t launch_tp(const WaveParams &p, int64_t n_reads, hipStream_t stream) {
    const int64_t waves = (n_reads + RPW - 1) / RPW;
    const unsigned blocks = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (p.in.dtype != kF32) {  // half-precision posteriors: the general instantiations only
        if (p.out.ambiguous)
            hipLaunchKernelGGL((beam_wave_kernel<N, GW, RPW, S, true, false, false, true, PDQ>), dim3(blocks),
                               dim3(64 * kWavesPerBlock), 0, stream, p);
        else
            hipLaunchKernelGGL((beam_wave_kernel<N, GW, RPW, S, false, false, false, true, PDQ>), dim3(blocks),
                               dim3(64 * kWavesPerBlock), 0, stream, p);
        return hipGetLastError();
    }
    if (p.out.ambiguous)
        hipLaunchKernelGGL((beam_wave_kernel<N, GW, RPW, S, true, false, false, false, PDQ>), dim3(blocks), dim3(64 * kWavesPerBlock), 0,
                           stream, p);
    else if (p.a.prof && N == 5 && GW == 6 && RPW == 2 && S == 0)   // the headline instantiation only
        hipLaunchKernelGGL((beam_wave_kernel<5, 6, 2, 0, false, true, false, false, PDQ>), dim3(blocks), dim3(64 * kWavesPerBlock), 0,
                           stream, p);
    else if (!p.in.lengths)  // reads of one length
        hipLaunchKernelGGL((beam_wave_kernel<N, GW, RPW, S, false, false, true, false, PDQ>), dim3(blocks), dim3(64 * kWavesPerBlock), 0,
                           stream, p);
    else
        hipLaunchKernelGGL((beam_wave_kernel<N, GW, RPW, S, false, false, false, false, PDQ>), dim3(blocks), dim3(64 * kWavesPerBlock), 0,
                           stream, p);
    return hipGetLastError();
}

template <int N, int GW, int RPW, int S = 0>
hipError_t launch_t(const WaveParams &p, int64_t n_reads, hipStream_t stream) {
    // the tie order of sort_unstable_by differs from the exact rank only above 20 candidates (pdq178.h): shapes
    // that cannot hold that many have no second instantiation
    constexpr bool CAN_TIE = ((64 / RPW) / GW) * N > 20;
    if (CAN_TIE && p.a.tie_order == FCD_TIE_PDQ178) return launch_tp<N, GW, RPW, S, CAN_TIE>(p, n_reads, stream);
    return launch_tp<N, GW, RPW, S, false>(p, n_reads, stream);
}

}  // namespace

int beam_wave_id_shift(int beam_size, int N, int force_one_read_per_wave) {
    return (beam_size <= 5 && N <= 5 && !force_one_read_per_wave) ? 5 : 6;  // KS of the instantiation launch_beam_wave picks
}

bool beam_wave_supported(int beam_size, int N, int crf, int S) {
    if (beam_size < 1 || beam_size > 12) return false;
    // the CRF instantiations: 5 symbols x 4 states (register FIFO) or x 8, 16, ... 2^k states (row gather)
    if (crf) return N == 5 && S >= 4 && (S & (S - 1)) == 0;
    if (beam_size > 8) return N >= 3 && N <= 5;
    return N >= 3 && N <= 7;
}

hipError_t launch_beam_wave(const BatchDesc &in, int64_t read_begin, int64_t n_reads,
                            const BeamArgs &a, const WaveArena &arena, const ResultDesc &out,
                            hipStream_t stream) {
    if (n_reads <= 0) return hipSuccess;
    WaveParams p{in, a, arena, out, read_begin};
    p.in.n_reads = n_reads;  // reads in this launch
    const bool two = a.beam_size <= 5 && in.N <= 5 && !a.force_one_read_per_wave;
    const bool wide = a.beam_size > 8;  // 9..12 beam slots: groups of five lanes
    if (a.crf) {
        if (!beam_wave_supported(a.beam_size, in.N, 1, in.S)) return hipErrorInvalidValue;
        if (in.S == 4) {
            if (wide) return launch_t<5, 5, 1, 4>(p, n_reads, stream);
            return two ? launch_t<5, 6, 2, 4>(p, n_reads, stream) : launch_t<5, 8, 1, 4>(p, n_reads, stream);
        }
        if (wide) return launch_t<5, 5, 1, kCrfGather>(p, n_reads, stream);
        return two ? launch_t<5, 6, 2, kCrfGather>(p, n_reads, stream) : launch_t<5, 8, 1, kCrfGather>(p, n_reads, stream);
    }
    if (wide) {
        switch (in.N) {
            case 3: return launch_t<3, 5, 1>(p, n_reads, stream);
            case 4: return launch_t<4, 5, 1>(p, n_reads, stream);
            case 5: return launch_t<5, 5, 1>(p, n_reads, stream);
        }
        return hipErrorInvalidValue;
    }
    if (two) {
        switch (in.N) {
            case 3: return launch_t<3, 6, 2>(p, n_reads, stream);
            case 4: return launch_t<4, 6, 2>(p, n_reads, stream);
            case 5: return launch_t<5, 6, 2>(p, n_reads, stream);
        }
    }
    switch (in.N) {
        case 3: return launch_t<3, 8, 1>(p, n_reads, stream);
        case 4: return launch_t<4, 8, 1>(p, n_reads, stream);
        case 5: return launch_t<5, 8, 1>(p, n_reads, stream);
        case 6: return launch_t<6, 8, 1>(p, n_reads, stream);
        case 7: return launch_t<7, 8, 1>(p, n_reads, stream);
    }
    return hipErrorInvalidValue;
}

// this translation unit's copy of the replay's std-form word (pdq178.h), on the current device
FCD_PDQ178_DEFINE_STD_FORM_SETTER(beam_wave_set_pdq178_std_form)

}  // namespace fcd
