// duplex.hip -- 2D pair-consensus beam search, duplex::beam_search
// (/root/reference/src/duplex.rs:443-650), one read PAIR per wavefront.
//
// The search over read 1 is the 1D prefix beam search of beam_generic.hip carried out in log
// space (LogSpace, duplex.rs:7-80) without per-step renormalisation; every tree node additionally
// owns a windowed CTC forward vector over read 2 ("SecondaryProbs", :152-210): (label, gap) log
// probabilities of the node's labelling ending at each row of read 2 inside the envelope, and
// their running maximum.  A candidate's score is prob_1 (+) max over the window of prob_2 (:146-148).
//
// Layout: the beam is a structure of arrays in LDS (as in beam_generic.hip).  Forward vectors
// live in the per-pair HBM arena as rings of Wcap = (widest envelope + 2) entries of
// {label, gap, label(+)gap}; entry for row t sits at slot t mod Wcap, so the reference's
// discard_until (:181-191) is a pure offset update.  Storing label(+)gap makes update_max
// (:193-204) a transcendental-free max-reduction done by all 64 lanes.
// RESIDENT WINDOWS (r03): while a node is a beam entry its ring ALSO lives in LDS (one buffer per beam slot,
// carried from step to step; entries keep their buffer when the beam is re-ranked, a node entering the beam has
// its ring copied in once), together with its window bounds and running maximum.  The per-step work on beam
// entries -- appending one row, the incremental update_max, serving rows to the children's builds -- then
// touches LDS only; the HBM arena is written through (it stays the truth for nodes outside the beam, which may
// come back) but is read again only when a node enters the beam.  Before r03 every step re-read every beam
// entry's window from HBM into an LDS tile (10.8 k cycles) after a chain of dependent global loads in the
// extension (30 k of the step's 175 k cycles, profiles/r03b_duplex_account.jsonl).
// Per row of read 1:
//   1. envelope check (:485-488);
//   2. if the upper bound grew (:490-522): beam re-sorted by node (parents first) and every beam
//      node's vector extended to the new bound -- sequential per node, as the recurrence is;
//   3. expansion (:526-593): slot (i,k) evaluation as in the 1D kernel; every NEW node builds its
//      whole window (:212-249), one node per lane, all new nodes of the step in parallel;
//   4. merge / prob_2_max refresh / NaN check / exact rank on (probability desc, node asc) /
//      truncate (:597-636).
//
// Log-space addition (:42-63): big + ln_1p(exp(small - big)).  The reference calls the platform
// libm through Rust's f32::exp / ln_1p, whose last bit depends on the glibc version; this kernel
// defines them as CORRECTLY ROUNDED f32 results (evaluated in f64 and rounded once), which is what
// glibc >= 2.41 (CORE-MATH) returns.  mode FCD_LOGADD_MAX reproduces builds with the reference's
// default `fastexp` feature, where exp() is identically 0 and the addition degenerates to max().
#include "device_utils.h"
#include "fcd_internal.h"
#define FCD_PDQ178_FORM0_ONLY 1  // (pdq178.h: this kernel replays the default std form only)
#include "pdq178.h"
#include "glibc235_math.h"
#include "logadd_fast.h"
#include "duplex_math.h"

namespace fcd {

namespace {

struct DuplexParams {
    const float *ln1, *ln2;   // log-space posteriors, [pair][Tcap][N] contiguous
    int64_t T1cap, T2cap;
    const int64_t *len1, *len2;  // nullable per-pair row counts
    const uint64_t *env;         // [pair][env_stride][2]
    int64_t env_stride;
    int N;
    int beam_size;
    float thr_ln;
    int collapse;
    int mode;
    int S;                       // transition states (1 for the plain search)
    int crf;                     // 1: duplex::crf_beam_search (:652-834)
    const float *init1, *init2;  // CRF: [pair * init_stride] initial state scores
    int64_t n_init1, n_init2, init1_stride, init2_stride;
    // arena (per pair slabs)
    int4 *meta;      // {parent, label, offset, end}
    float *nmax;     // running max per node
    int32_t *rlo;    // first row the running max covers (lower bound of the last update_max)
    int32_t *rows;   // NL child ids per node
    float *vec;      // Wcap * 3 floats per node
    float *rootgap;  // T2cap + 1 per pair
    int64_t cap_nodes;
    int Wcap;
    int staged;      // 1: the step's read-2 window and the beam's forward windows are tiled in LDS
    ResultDesc out;
    int64_t pair_begin;
    uint32_t *prof;  // developer instrument (fcd_debug_set_duplex_profile): [pair][8] shader cycles per phase, nullable
    int tie_order;   // FCD_TIE_PDQ178 / FCD_TIE_STABLE (include/fcd.h)
};


// (no runtime-indexed pointer arrays: they would force the struct into scratch memory)
struct DLds {
    int *beam0;  // [2][beam_stride] words: node, lp, gp, tip, par, state, buf, off, end, rlo, max (BC each), child (BC*NL)
    int beam_stride;
    int BC;
    __device__ __forceinline__ int *b_node(int b) const { return beam0 + b * beam_stride; }
    __device__ __forceinline__ float *b_lp(int b) const { return reinterpret_cast<float *>(b_node(b) + BC); }
    __device__ __forceinline__ float *b_gp(int b) const { return reinterpret_cast<float *>(b_node(b) + 2 * BC); }
    __device__ __forceinline__ int *b_tip(int b) const { return b_node(b) + 3 * BC; }
    __device__ __forceinline__ int *b_par(int b) const { return b_node(b) + 4 * BC; }
    __device__ __forceinline__ int *b_state(int b) const { return b_node(b) + 5 * BC; }
    // resident window of the entry's node: LDS buffer index, rows [off, end) present, the running maximum and the
    // first row it covers (the LDS twins of meta.z / meta.w / nmax / rlo)
    __device__ __forceinline__ int *b_buf(int b) const { return b_node(b) + 6 * BC; }
    __device__ __forceinline__ int *b_off(int b) const { return b_node(b) + 7 * BC; }
    __device__ __forceinline__ int *b_end(int b) const { return b_node(b) + 8 * BC; }
    __device__ __forceinline__ int *b_rlo(int b) const { return b_node(b) + 9 * BC; }
    __device__ __forceinline__ float *b_max(int b) const { return reinterpret_cast<float *>(b_node(b) + 10 * BC); }
    __device__ __forceinline__ int *b_child(int b) const { return b_node(b) + 11 * BC; }
    uint64_t *c_key;
    // FCD_TIE_PDQ178 (pdq178.h): the node-ordered candidate list of a tie-flagged step and the quicksort's scratch
    uint64_t *pq_list;  // C
    pdq178::Scratch *pq_scr;
    float *c_lp, *c_gp, *c_p2;
    int *c_id, *c_new;
    int *nb_src;
    int *s_off, *s_end;  // BC each: scratch of the buffer hand-over (which buffers stay taken / the free list)
    int *bt;             // 64: lane that owns the m-th new node of the current pass
    float *w2;           // Wmax*S*N: log posteriors of read 2, rows [lo, hi), all states
    float *bw;           // BC*(Wmax+2)*3: the beam entries' resident rings {label, gap, label(+)gap}, slot = row mod Wcap
};

// pdq: FCD_TIE_PDQ178 with more than 20 candidates possible -- only then the quicksort's list and scratch exist
__host__ __device__ inline size_t dlds_words(int BC, int N, int Wmax, int S, bool pdq) {
    const int NL = N - 1;
    const size_t C = (size_t)BC * N;
    return 2 * (size_t)BC * (11 + NL) + 2 * C + (pdq ? 2 * C + (sizeof(pdq178::Scratch) + 3) / 4 : 0) + 5 * C + 3 * (size_t)BC + 4 + 64 +
           (size_t)Wmax * S * N + (Wmax > 0 ? (size_t)BC * (Wmax + 2) * 3 : 0);
}

__device__ inline DLds dcarve(int *smem, int BC, int N, int Wmax, int S, bool pdq) {
    DLds L;
    const int NL = N - 1;
    const size_t C = (size_t)BC * N;
    L.c_key = reinterpret_cast<uint64_t *>(smem);
    int *p = smem + 2 * C;
    L.pq_list = reinterpret_cast<uint64_t *>(p);
    if (pdq) p += 2 * C;
    L.pq_scr = reinterpret_cast<pdq178::Scratch *>(p);
    if (pdq) p += (sizeof(pdq178::Scratch) + 3) / 4;
    L.c_lp = reinterpret_cast<float *>(p); p += C;
    L.c_gp = reinterpret_cast<float *>(p); p += C;
    L.c_p2 = reinterpret_cast<float *>(p); p += C;
    L.c_id = p; p += C;
    L.c_new = p; p += C;
    L.BC = BC;
    L.beam_stride = BC * (11 + NL);
    L.beam0 = p;
    p += 2 * (size_t)L.beam_stride;
    L.nb_src = p; p += BC;
    L.s_off = p; p += BC;
    L.s_end = p; p += BC;
    L.bt = p; p += 64;
    L.w2 = reinterpret_cast<float *>(p); p += (size_t)Wmax * S * N;
    L.bw = reinterpret_cast<float *>(p);
    return L;
}

// Forward vector access: the reference's SecondaryProbs::get (:167-179) on a ring.
struct VecRef {
    const float *base;  // node's ring (Wcap*3 floats) or the root's gap array
    int offset, end;    // rows [offset, end) are present
    bool root;
};

__device__ __forceinline__ void vec_get(const VecRef &v, int at, int Wcap, float &gap, float &sum) {
    if (at < v.offset || at >= v.end) {
        gap = kNegInf;
        sum = kNegInf;
        return;
    }
    if (v.root) {  // root_probs (:389-409): label = zero, gap = cumulative blank product
        gap = load_f32_l2(v.base + (at + 1));
        sum = gap;
        return;
    }
    const int slot = at % Wcap;
    gap = load_f32_l2(v.base + 3 * slot + 1);
    sum = load_f32_l2(v.base + 3 * slot + 2);
}

// Ring slot of row t, for t within a ring's length of the step's lower bound lo (lo_m = lo mod Wcap, computed once per
// step): an add and two conditional corrections where `t % Wcap` on a run-time Wcap is a ~25-instruction division --
// which the per-step bookkeeping used to pay a few dozen times per step.
__device__ __forceinline__ int slot_near(int t, int lo, int lo_m, int Wcap) {
    int s = lo_m + (t - lo);
    s = s < 0 ? s + Wcap : s;
    return s >= Wcap ? s - Wcap : s;
}

// Two pieces of the resident-window bookkeeping live OUT OF LINE: inlined, they pushed the logsumexp instantiations
// (whose window-building loop pins ~60 coefficient registers) over the register file and spilled 226 - 276 VGPRs to
// scratch memory.  Both run a few times per step; a call costs far less than the spills did.
//
// A node entering the beam: rows [off, off + n3 / 3) of its ring, arena -> LDS (eight loads in flight per lane).
__device__ __attribute__((noinline)) void ring_copy_in(const float *src, float *dst, int off, int n3, int Wcap, int lane) {
    // arena ring and LDS ring have the same geometry: element i of the window sits at flat index (off mod Wcap) * 3 + i,
    // wrapped once at 3 * Wcap, on both sides
    const int W3 = 3 * Wcap;
    const int s03 = (__builtin_amdgcn_readfirstlane(off) % Wcap) * 3;
    for (int base = 0; base < n3; base += 8 * kWave) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * kWave + lane;
            int f = s03 + idx;
            f = f >= W3 ? f - W3 : f;
            v[u] = idx < n3 ? load_f32_l2(src + f) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * kWave + lane;
            int f = s03 + idx;
            f = f >= W3 ? f - W3 : f;
            if (idx < n3) dst[f] = v[u];
        }
    }
}

// Up to two nodes entering the beam in the same step: their WHOLE rings (rows outside a window's bounds are never read,
// so they may come along), all loads of both in flight before the first store -- one trip to the arena for the pair,
// and none of it waits for the nodes' bounds.
__device__ __attribute__((noinline)) void ring_copy_whole2(const float *src_a, float *dst_a, const float *src_b, float *dst_b,
                                                           int W3, int lane) {
    for (int base = 0; base < W3; base += 7 * kWave) {
        float va[7], vb[7];  // (a ring of 130 rows is 390 words: one pass)
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int idx = base + u * kWave + lane;
            va[u] = idx < W3 ? load_f32_l2(src_a + idx) : 0.0f;
            vb[u] = (src_b && idx < W3) ? load_f32_l2(src_b + idx) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int idx = base + u * kWave + lane;
            if (idx < W3) {
                dst_a[idx] = va[u];
                if (src_b) dst_b[idx] = vb[u];
            }
        }
    }
}

// extend_secondary_probs' recurrence (:361-386) for ONE beam entry on its own lane: rows [end, hi) appended to the
// resident ring `mw` and, written through, to the arena ring `my`; returns the running maximum.  The parent's rows
// come from its resident ring `prg` (bounds p_off / p_end) or, when it is not a beam entry, from the arena (`pv`).
template <int MODE>
__device__ __attribute__((noinline)) float extend_rows(float *my, float *mw, const float *prg, VecRef pv, int p_off,
                                                       int p_end, const float *tile, int tile_step, int lo, int end,
                                                       int hi, int lab, bool is_rep, int Wcap, float l_lab, float l_sum,
                                                       float mx, int lo_m) {
    for (int idx = end; idx < hi; ++idx) {  // idx >= lo: the rows of read 2 are in the LDS tile
        const float *row = tile + (idx - lo) * tile_step;
        float pg, ps;
        if (prg) {
            const int at = idx - 1;
            if (at < p_off || at >= p_end) {
                pg = kNegInf;
                ps = kNegInf;
            } else {
                const int sl = slot_near(at, lo, lo_m, Wcap) * 3;  // at >= lo - 1
                pg = prg[sl + 1];
                ps = prg[sl + 2];
            }
        } else {
            vec_get(pv, idx - 1, Wcap, pg, ps);
        }
        const float g = l_sum + row[0];
        const float xx = is_rep ? pg : ps;
        const float lb = row[lab + 1] + ladd<MODE>(l_lab, xx);
        const float sm = ladd<MODE>(lb, g);
        const int sl = slot_near(idx, lo, lo_m, Wcap) * 3;
        store_row(my + sl, lb, g, sm);
        mw[sl] = lb;
        mw[sl + 1] = g;
        mw[sl + 2] = sm;
        mx = lmax(mx, sm);
        l_lab = lb;
        l_sum = sm;
    }
    return mx;
}

// PIN: the log-add coefficients stay in vector registers across the window-building loop (60 registers: the
// instantiation for batches that leave a wavefront alone on its SIMD, where a row's instruction count is
// what the loop costs); without it the kernel fits three wavefronts per SIMD, which wins once the GPU is full.
template <int MODE, bool PIN>
__global__ __launch_bounds__(64, PIN ? 2 : 3) void duplex_kernel(DuplexParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x;
    const int64_t local = blockIdx.x;
    const int64_t r = p.pair_begin + local;
    const int N = p.N, NL = N - 1, BC = p.beam_size, Wcap = p.Wcap, S = p.S;
    const bool crf = p.crf != 0;
    const bool collapse = !crf && p.collapse != 0;
    const float thr = p.thr_ln;
    const bool staged = p.staged != 0;
    const int Wmax = staged ? Wcap - 2 : 0;
    DLds L = dcarve(smem, BC, N, Wmax, S, p.tie_order == FCD_TIE_PDQ178 && (int64_t)BC * N > 20);

    int64_t T1 = p.T1cap, T2 = p.T2cap;
    if (p.len1) { int64_t t = p.len1[r]; T1 = t < 0 ? 0 : (t < T1 ? t : T1); }
    if (p.len2) { int64_t t = p.len2[r]; T2 = t < 0 ? 0 : (t < T2 ? t : T2); }
    const float *ln1 = p.ln1 + r * p.T1cap * S * N;
    const float *ln2 = p.ln2 + r * p.T2cap * S * N;
    const uint64_t *env = p.env + r * p.env_stride * 2;
    int4 *meta = p.meta + local * p.cap_nodes;
    float *nmax = p.nmax + local * p.cap_nodes;
    int32_t *rlo = p.rlo + local * p.cap_nodes;
    int32_t *rows = p.rows + local * p.cap_nodes * NL;
    float *vec = p.vec + local * p.cap_nodes * (int64_t)Wcap * 3;
    float *rootgap = p.rootgap + local * (p.T2cap + 1);
    uint8_t *lab_out = p.out.labels + r * p.out.out_stride;

    // tie instrument (fcd_result.ambiguous; the counters of the 1D kernels, include/fcd.h): the prune below is the
    // reference's sort_unstable_by (:620,:807), whose order of EQUAL probabilities above 20 candidates is pdqsort's
    const bool count_amb = p.out.ambiguous != nullptr;
    int n_amb = 0, n_crit = 0;
    auto fail = [&](int code) {
        if (lane == 0) {
            p.out.status[r] = code;
            p.out.out_len[r] = 0;
            if (count_amb) {
                p.out.ambiguous[2 * r] = (uint32_t)n_amb;
                p.out.ambiguous[2 * r + 1] = (uint32_t)n_crit;
            }
        }
    };

    // ---- root_probs (:389-409): needs envelope[(0,1)], so an empty read 1 panics in the reference
    if (T1 <= 0) return fail(FCD_ST_BAD_STATE);
    const uint64_t ub_u = env[1];
    if (ub_u > (uint64_t)T2) return fail(FCD_ST_BAD_STATE);  // slice(s![..upper_bound]) panics
    const int root_end = (int)ub_u;                            // root rows are [-1, ub)
    // CRF start states: init_state.argmax() (:679,:691; first maximum, NaN panics in the reference)
    int st1 = 0, st2 = 0;
    if (crf) {
        bool bad = false;
        for (int which = 0; which < 2; ++which) {
            const float *init = which ? p.init2 + r * p.init2_stride : p.init1 + r * p.init1_stride;
            const int64_t n_init = which ? p.n_init2 : p.n_init1;
            int arg = 0;
            float m = init[0];
            bad = bad || (m != m);
            for (int64_t j = 1; j < n_init; ++j) {
                const float e = init[j];
                bad = bad || (e != e);
                if (e > m) { m = e; arg = (int)j; }
            }
            bad = bad || arg >= S;
            if (which) st2 = arg; else st1 = arg;
        }
        if (bad) return fail(FCD_ST_BAD_STATE);
    }
    if (lane == 0) {
        // root_probs (:389-409) / crf_root_probs (:411-441): cumulative blank product
        float cur = 0.0f;
        rootgap[0] = cur;
        int st = st2;
        for (int t = 0; t < root_end; ++t) {
            cur = cur + ln2[((int64_t)t * S + st) * N];
            rootgap[t + 1] = cur;
            if (crf) st = (int)(((int64_t)st * NL) % S);  // :437
        }
        L.b_state(0)[0] = st1;
        L.b_node(0)[0] = -1;
        L.b_lp(0)[0] = kNegInf;  // label: zero
        L.b_gp(0)[0] = 0.0f;     // gap: one
        L.b_tip(0)[0] = -1;
        L.b_par(0)[0] = -2;
        L.b_buf(0)[0] = 0;       // the root's rows are staged into its buffer every step (below)
        L.b_off(0)[0] = -1;
        L.b_end(0)[0] = root_end;
        L.b_rlo(0)[0] = 0;
        L.b_max(0)[0] = 0.0f;
    }
    for (int j = lane; j < NL; j += kWave) L.b_child(0)[j] = -1;
    __syncthreads();

    int cur = 0, B = 1, nn = 0;
    int last_hi = 0;
    // cycle account (p.prof): 0 envelope + vector extension, 1 LDS tiles, 2 expansion without the window builds,
    // 3 window builds of the new nodes, 4 rank + next beam, 5 build-loop iterations, 6 new nodes, 7 steps
    const bool prof = p.prof != nullptr;
    uint64_t acc[5] = {0, 0, 0, 0, 0}, t_prev = 0, t_now = 0;
    uint32_t n_iter = 0, n_newnodes = 0, n_slow = 0, n_enter = 0, n_ext = 0;
    uint64_t sub[5] = {0, 0, 0, 0, 0}, t_sub = 0, t_sub2 = 0;  // finer stamps inside phases 0 and 4 (developer instrument)
#define FCD_SUB_BEGIN() if (prof) FCD_STAMP(t_sub, stamp_dep);
#define FCD_SUB(k) if (prof) { FCD_STAMP(t_sub2, stamp_dep); sub[k] += t_sub2 - t_sub; t_sub = t_sub2; }
    int stamp_dep = 0;
    if (prof) FCD_STAMP(t_prev, stamp_dep);
#define FCD_DUPLEX_PHASE(k)                  \
    if (prof) {                              \
        FCD_STAMP(t_now, stamp_dep);         \
        acc[k] += t_now - t_prev;            \
        t_prev = t_now;                      \
    }

    auto node_vec = [&](int node, int off, int end) {
        VecRef v;
        if (node < 0) {
            v.base = rootgap;
            v.offset = -1;
            v.end = root_end;
            v.root = true;
        } else {
            v.base = vec + (int64_t)node * Wcap * 3;
            v.offset = off;
            v.end = end;
            v.root = false;
        }
        return v;
    };

    // ---- resident windows (staged launches): LDS twins of the beam entries' rings ----
    const bool resident = staged;
    auto ring = [&](int buf) { return L.bw + (size_t)buf * Wcap * 3; };
    // node enters the beam (or its ring was rewritten by the sequential extension): bounds, maximum and rows
    // [off, end) come in from the arena, which is always written through; wave-cooperative, wave-uniform arguments
    auto load_entry = [&](int bsel, int e) {
        const int node = L.b_node(bsel)[e];
        if (node < 0) return;  // the root's rows are staged per step
        const int4 m = load_meta_l2(&meta[node]);
        const int off = m.z, end = m.w;
        ring_copy_in(vec + (int64_t)node * Wcap * 3, ring(L.b_buf(bsel)[e]), off, (end - off) * 3, Wcap, lane);
        if (lane == 0) {
            L.b_off(bsel)[e] = off;
            L.b_end(bsel)[e] = end;
            L.b_max(bsel)[e] = load_f32_l2(&nmax[node]);
            L.b_rlo(bsel)[e] = load_i32_l2(&rlo[node]);
        }
    };

    // (the envelope row of step t1 + 1 is requested during step t1: no global round trip at the head of a step)
    uint64_t env_lo_next = env[0], env_hi_next = env[1];
    for (int64_t t1 = 0; t1 < T1; ++t1) {
        // ---- envelope (:485-488) ----
        const uint64_t lo_u = env_lo_next, hi_u = env_hi_next;
        if (t1 + 1 < T1) {
            env_lo_next = env[2 * (t1 + 1)];
            env_hi_next = env[2 * (t1 + 1) + 1];
        }
        const int hi = (int)(hi_u > (uint64_t)T2 ? (uint64_t)T2 : hi_u);
        if (lo_u >= (uint64_t)hi || lo_u > (uint64_t)last_hi) return fail(FCD_ST_INVALID_ENVELOPE);
        const int lo = (int)lo_u;

        const int W = hi - lo;
        const int lo_m = lo % Wcap;  // (wave-uniform, once per step: see slot_near)
        // ---- LDS tile of read 2's rows [lo, hi) for this row of read 1 (the extension below reads it too) ----
        // Loaded AFTER the beam entries have asked for their parents' bounds (below): one trip to memory for both.
        bool tile_done = !staged;
        auto load_tile = [&]() {
            __syncthreads();
            FCD_SUB_BEGIN()
            for (int xw = lane; xw < W * S * N; xw += kWave) L.w2[xw] = ln2[(int64_t)lo * S * N + xw];
            __syncthreads();
            FCD_SUB(0)
            tile_done = true;
        };
        if (hi > last_hi) {
            FCD_SUB_BEGIN()
            // ---- :493 beam.sort_by_key(node): parents before children ----
            const int nx = cur ^ 1;
            for (int e = lane; e < B; e += kWave) {
                const int nd = L.b_node(cur)[e];
                int rk = 0;
                for (int j = 0; j < B; ++j) rk += (L.b_node(cur)[j] < nd) ? 1 : 0;
                L.b_node(nx)[rk] = nd;
                L.b_lp(nx)[rk] = L.b_lp(cur)[e];
                L.b_gp(nx)[rk] = L.b_gp(cur)[e];
                L.b_tip(nx)[rk] = L.b_tip(cur)[e];
                L.b_par(nx)[rk] = L.b_par(cur)[e];
                L.b_state(nx)[rk] = L.b_state(cur)[e];
                L.b_buf(nx)[rk] = L.b_buf(cur)[e];
                L.b_off(nx)[rk] = L.b_off(cur)[e];
                L.b_end(nx)[rk] = L.b_end(cur)[e];
                L.b_rlo(nx)[rk] = L.b_rlo(cur)[e];
                L.b_max(nx)[rk] = L.b_max(cur)[e];
                for (int l = 0; l < NL; ++l) L.b_child(nx)[rk * NL + l] = L.b_child(cur)[e * NL + l];
            }
            cur = nx;
            __syncthreads();
            // ---- extend_secondary_probs (:338-387) for every beam node ----
            // Fast path (the sliding-band case): the bound grew by exactly one row and every beam
            // node's vector already ends at the old bound, so each node appends ONE row that only
            // reads rows its parent already holds -- all nodes proceed in parallel, one per lane.
            // update_max (:193-204) is evaluated incrementally: the maximum over [lo, end) equals
            // the stored maximum over [rlo, end) unless one of the rows leaving the range attains it.
            bool fast_ok = B <= kWave;
            if (resident) {
                // ---- all beam entries in parallel, one per lane, on the resident rings ----
                // An entry appends rows [end, hi): one row when its window ends at the previous bound, several when
                // the node sat outside the beam for a while and catches up (on BASELINE config 5 two steps in three
                // bring such a node in: picked as an EXISTING child, its window is as stale as its creation).  Row idx
                // reads the parent's row idx - 1 <= hi - 2, which exists BEFORE this step whenever the parent's
                // window reaches hi - 1 -- then nothing depends on a row written in this step and the reference's
                // parents-first order (:493) is immaterial.  Only a parent that is itself behind, or a bound that
                // jumped, takes the sequential path below.
                const int e = lane;
                const bool mine = e < B && L.b_node(cur)[e] >= 0;
                int node = -1, parent = -1, lab = 0, off = 0, end = 0, rl = 0, p_off = 0, p_end = 0, p_lab = -1,
                    pslot = -1;
                float mx = kNegInf;
                bool bad = false, rescan = false, panic = false;
                int4 pm = make_int4(0, 0, 0, 0);
                if (fast_ok && mine) {
                    node = L.b_node(cur)[e];
                    parent = L.b_par(cur)[e]; lab = L.b_tip(cur)[e];
                    off = L.b_off(cur)[e]; end = L.b_end(cur)[e];
                    mx = L.b_max(cur)[e]; rl = L.b_rlo(cur)[e];
                    if (lo > off) {  // :351-359
                        const int keep = lo - 1;
                        if (keep > off) {
                            if (keep < end) off = keep;
                            else { off = keep; end = keep; }
                        }
                        if (end == off) { off = lo; end = lo; }
                        rescan = true;  // update_max(lo, hi): recomputed from the ring by the whole wavefront below
                        rl = lo;
                    }
                    panic = end >= hi;  // assert!(current_end < upper_bound) :363-366
                    if (parent >= 0) {
                        for (int j = 0; j < B; ++j)
                            if (L.b_node(cur)[j] == parent) pslot = j;
                        // (asked for whether or not the parent turns out to be a beam entry, and not looked at before the
                        // read-2 tile has been requested too: nothing here waits for it)
                        pm = load_meta_l2(&meta[parent]);
                        if (pslot >= 0) {  // the parent is a beam entry: its window is resident too
                            p_lab = L.b_tip(cur)[pslot]; p_off = L.b_off(cur)[pslot]; p_end = L.b_end(cur)[pslot];
                            bad = p_end < hi - 1;  // it has yet to write a row this entry needs: parents first
                        }
                    }
                }
                FCD_SUB(4)
                load_tile();  // (the parents' bounds above are still in flight: they arrive with the tile)
                if (parent >= 0 && pslot < 0) {
                    p_lab = pm.y; p_off = pm.z; p_end = pm.w;
                }
                if (ballot(panic) != 0ull) return fail(FCD_ST_BAD_STATE);
                fast_ok = fast_ok && ballot(bad) == 0ull;
                if (fast_ok) {
                    // update_max over the rows that stay, [max(lo, off), end), for every entry that discarded rows:
                    // 64 lanes per entry, two LDS reads each at W = 128 (NaN entries never replace the maximum).  The
                    // incremental form of r02 ("does a leaving row hold the maximum?") sent 68 % (logsumexp) / 97 %
                    // (max mode: the best single path loses probability with every row, so the maximum sits at the
                    // window's first row) of config 5's steps down the sequential path.
                    const int my_buf = mine ? L.b_buf(cur)[e] : 0;
                    for (uint64_t m = ballot(rescan); m != 0ull; m &= m - 1) {
                        // (wave-uniform entry: its bounds and buffer come by v_readlane, not through LDS)
                        const int e2 = (int)__builtin_ctzll(m);
                        const int o2 = __builtin_amdgcn_readlane(off, e2), n2 = __builtin_amdgcn_readlane(end, e2);
                        const float *rg = ring(__builtin_amdgcn_readlane(my_buf, e2));
                        float part = kNegInf;
                        for (int t0 = (lo > o2 ? lo : o2) + lane; t0 < n2; t0 += 3 * kWave) {  // three reads in flight
                            const int t1r = t0 + kWave, t2r = t0 + 2 * kWave;
                            const float v0 = rg[3 * slot_near(t0, lo, lo_m, Wcap) + 2];
                            const float v1 = t1r < n2 ? rg[3 * slot_near(t1r, lo, lo_m, Wcap) + 2] : kNegInf;
                            const float v2 = t2r < n2 ? rg[3 * slot_near(t2r, lo, lo_m, Wcap) + 2] : kNegInf;
                            part = lmax(lmax(lmax(part, v0), v1), v2);
                        }
                        part = wave_lmax(part);
                        if (lane == e2) mx = part;
                    }
                }
                FCD_SUB(2)
                if (fast_ok && mine) {
                    const bool is_rep = !crf && parent >= 0 && p_lab == lab;  // :512 (crf: :320-334, no repeat case)
                    const int tst = L.b_state(cur)[e];                        // crf: the entry's own state (:725-728)
                    float *my = vec + (int64_t)node * Wcap * 3;               // the arena is written through
                    float *mw = ring(L.b_buf(cur)[e]);
                    float l_lab = kNegInf, l_sum = kNegInf;
                    if (end > off) {
                        const int sl = slot_near(end - 1, lo, lo_m, Wcap);  // end - 1 >= off >= lo - 1 after the discard above
                        l_lab = mw[3 * sl];
                        l_sum = mw[3 * sl + 2];
                    }
                    const float *prg = pslot >= 0 ? ring(L.b_buf(cur)[pslot]) : nullptr;
                    const VecRef pv = node_vec(parent, p_off, p_end);
                    mx = extend_rows<MODE>(my, mw, prg, pv, p_off, p_end, L.w2 + tst * N, S * N, lo, end, hi, lab, is_rep,
                                           Wcap, l_lab, l_sum, mx, lo_m);  // :361-386
                    meta[node] = make_int4(parent, lab, off, hi);
                    nmax[node] = mx;
                    rlo[node] = rl;
                    L.b_off(cur)[e] = off;
                    L.b_end(cur)[e] = hi;
                    L.b_max(cur)[e] = mx;
                    L.b_rlo(cur)[e] = rl;
                }
            } else {
                fast_ok = fast_ok && hi == last_hi + 1;
                {
                    const int e = lane;
                    const bool mine = e < B && L.b_node(cur)[e] >= 0;
                    int node = -1, parent = -1, lab = 0, off = 0, end = 0, rl = 0, p_off = 0, p_end = 0,
                        p_lab = -1;
                    float mx = kNegInf;
                    bool bad = false;
                    if (fast_ok && mine) {
                        node = L.b_node(cur)[e];
                        const int4 m = load_meta_l2(&meta[node]);
                        parent = m.x; lab = m.y; off = m.z; end = m.w;
                        mx = load_f32_l2(&nmax[node]);
                        rl = load_i32_l2(&rlo[node]);
                        bad = end != last_hi;
                        if (!bad && lo > off) {
                            const int keep = lo - 1;
                            const int off_old = off;
                            if (keep > off) {
                                if (keep < end) off = keep;
                                else { off = keep; end = keep; }
                            }
                            if (end == off) { off = lo; end = lo; }
                            bad = end != last_hi;
                            if (!bad && lo < rl) bad = true;  // the range grows downwards: rescan (slow path)
                            if (!bad && lo > rl) {
                                if (lo - rl > 4) bad = true;  // a long stale range: rescan on the slow path
                                const float *my = vec + (int64_t)node * Wcap * 3;
                                for (int t = rl; t < lo && !bad; ++t) {
                                    if (t < off_old || t >= end) continue;
                                    const float sv = load_f32_l2(my + 3 * (t % Wcap) + 2);
                                    if (sv == sv && !(sv < mx)) bad = true;  // the leaving row holds the max
                                }
                                rl = lo;
                            }
                        }
                        if (!bad && parent >= 0) {
                            const int4 pm = load_meta_l2(&meta[parent]);
                            p_lab = pm.y; p_off = pm.z; p_end = pm.w;
                        }
                    }
                    fast_ok = fast_ok && ballot(bad) == 0ull;
                    if (fast_ok && mine) {
                        const VecRef pv = node_vec(parent, p_off, p_end);
                        const bool is_rep = !crf && parent >= 0 && p_lab == lab;  // :512 (crf: :320-334, no repeat case)
                        const int tst = L.b_state(cur)[e];                        // crf: the entry's own state (:725-728)
                        float *my = vec + (int64_t)node * Wcap * 3;
                        float l_lab = kNegInf, l_sum = kNegInf;
                        if (end > off) {
                            const int sl = (end - 1) % Wcap;
                            l_lab = load_f32_l2(my + 3 * sl);
                            l_sum = load_f32_l2(my + 3 * sl + 2);
                        }
                        const int idx = end;  // == last_hi == hi - 1
                        const float *row = ln2 + ((int64_t)idx * S + tst) * N;
                        float pg, ps;
                        vec_get(pv, idx - 1, Wcap, pg, ps);
                        const float g = l_sum + row[0];
                        const float xx = is_rep ? pg : ps;
                        const float lb = row[lab + 1] + ladd<MODE>(l_lab, xx);
                        const float sm = ladd<MODE>(lb, g);
                        const int sl = idx % Wcap;
                        my[3 * sl] = lb;
                        my[3 * sl + 1] = g;
                        my[3 * sl + 2] = sm;
                        mx = lmax(mx, sm);
                        meta[node] = make_int4(parent, lab, off, hi);
                        nmax[node] = mx;
                        rlo[node] = rl;
                    }
                }
            }
            __syncthreads();
            for (int e = 0; e < B && !fast_ok; ++e) {
                const int node = L.b_node(cur)[e];
                if (node < 0) continue;
                const int4 m = load_meta_l2(&meta[node]);
                const int parent = m.x, lab = m.y;
                int off = m.z, end = m.w;
                float mx = load_f32_l2(&nmax[node]);
                int p_off = 0, p_end = 0, p_lab = -1;
                if (parent >= 0) {
                    const int4 pm = load_meta_l2(&meta[parent]);
                    p_lab = pm.y;
                    p_off = pm.z;
                    p_end = pm.w;
                }
                const VecRef pv = node_vec(parent, p_off, p_end);
                const bool is_rep = !crf && parent >= 0 && p_lab == lab;  // :512, no collapse_repeats test
                const int tst = L.b_state(cur)[e];
                float *my = vec + (int64_t)node * Wcap * 3;
                if (lo > off) {  // :351-359
                    const int keep = lo - 1;
                    if (keep > off) {  // discard_until
                        if (keep < end) off = keep;
                        else { off = keep; end = keep; }
                    }
                    if (end == off) { off = lo; end = lo; }
                    // update_max(lo, hi): max of label(+)gap over rows [lo,hi) that are present
                    const int b0 = lo > off ? lo : off;
                    const int e0 = hi < end ? hi : end;
                    float part = kNegInf;
                    for (int t = b0 + lane; t < e0; t += kWave) {
                        const float s = load_f32_l2(my + 3 * (t % Wcap) + 2);
                        part = lmax(part, s);  // NaN entries never replace the running max
                    }
                    for (int o = 32; o > 0; o >>= 1) part = lmax(part, __shfl_xor(part, o));
                    mx = part;
                    if (lane == 0) rlo[node] = lo;
                }
                // `assert!(current_end < upper_bound)` (:363-366, :311-314): a window that already reaches the new
                // bound -- the envelope's upper bound moved back and then forward by less -- aborts the reference
                if (ballot(end >= hi) != 0ull) return fail(FCD_ST_BAD_STATE);  // (a vote: every lane has read `end`)
                // continue the recurrence from the stored end (:361-386); wave-uniform work
                float l_lab = kNegInf, l_gap = kNegInf, l_sum = kNegInf;
                if (end > off) {
                    const int s = (end - 1) % Wcap;
                    l_lab = load_f32_l2(my + 3 * s);
                    l_gap = load_f32_l2(my + 3 * s + 1);
                    l_sum = load_f32_l2(my + 3 * s + 2);
                }
                (void)l_gap;
                for (int idx = end; idx < hi; ++idx) {
                    const float *row = ln2 + ((int64_t)idx * S + tst) * N;
                    float pg, ps;
                    vec_get(pv, idx - 1, Wcap, pg, ps);
                    const float g = l_sum + row[0];
                    const float x = is_rep ? pg : ps;
                    const float lb = row[lab + 1] + ladd<MODE>(l_lab, x);
                    const float sm = ladd<MODE>(lb, g);
                    if (lane == 0) {
                        const int s = idx % Wcap;
                        my[3 * s] = lb;
                        my[3 * s + 1] = g;
                        my[3 * s + 2] = sm;
                    }
                    mx = lmax(mx, sm);
                    l_lab = lb;
                    l_sum = sm;
                }
                if (hi > end) end = hi;
                if (lane == 0) {
                    meta[node] = make_int4(parent, lab, off, end);
                    nmax[node] = mx;
                }
                __syncthreads();  // the next node may read this one's new rows (parents first)
            }
            if (prof) {
                ++n_ext;
                n_slow += fast_ok ? 0u : 1u;
            }
            if (!fast_ok && resident) {  // the sequential path worked on the arena: refresh the resident copies
                for (int e = 0; e < B; ++e) load_entry(cur, e);
                __syncthreads();
            }
        }
        if (!tile_done) load_tile();  // (a step whose upper bound did not move)
        last_hi = hi;
        FCD_DUPLEX_PHASE(0)

        if (staged) {
            // the root (in the beam for the first few rows only) has no ring in the arena: the rows its children's
            // builds will ask for, [lo-1, hi-1), are staged into its buffer from the cumulative blank products
            for (int e = 0; e < B; ++e) {
                if (L.b_node(cur)[e] >= 0) continue;
                float *rg = ring(L.b_buf(cur)[e]);
                for (int j = lane; j < W; j += kWave) {
                    const int at = lo - 1 + j;
                    if (at < -1 || at >= root_end) continue;
                    const float g = load_f32_l2(rootgap + (at + 1));
                    const int sl = (((at % Wcap) + Wcap) % Wcap) * 3;
                    rg[sl] = kNegInf;
                    rg[sl + 1] = g;
                    rg[sl + 2] = g;
                }
            }
            __syncthreads();
        }

        FCD_DUPLEX_PHASE(1)
        int *b_node = L.b_node(cur), *b_tip = L.b_tip(cur), *b_par = L.b_par(cur);
        int *b_child = L.b_child(cur);
        int *b_state = L.b_state(cur);
        float *b_lp = L.b_lp(cur), *b_gp = L.b_gp(cur);
        const int nslots = B * N;
        const float *frame1 = ln1 + t1 * S * N;
        int n_valid = 0;
        bool any_nan = false;

        // ---- expansion (:526-593) ----
        for (int base = 0; base < nslots; base += kWave) {
            const int c = base + lane;
            const bool act = c < nslots;
            const int i = act ? c / N : 0;
            const int k = act ? c - i * N : 0;
            const int node = b_node[i];
            const float lp = b_lp[i], gp = b_gp[i];
            const int tip = b_tip[i];
            const int state = b_state[i];
            const float *row1 = frame1 + state * N;  // crf: probs[state, :] (:749)
            bool valid = false, is_new = false, rep = false;
            float clp = kNegInf, cgp = kNegInf, p2 = 0.0f;
            int cid = -2;
            if (act) {
                if (k == 0) {
                    // The node's own candidate is the MERGE (:596-611) of up to three items that share the node -- the
                    // blank extension {zero, gap}, the repeat-stay {label, zero} and the extension that arrives from
                    // the parent's entry {label, zero} -- folded with LogSpace::add in the order the reference appends
                    // them: tips in beam order, and within a tip blank first, labels after.  The order is immaterial
                    // for ordinary numbers and for logsumexp, but max mode's add keeps a NaN only as its FIRST operand
                    // (`exp` is identically 0 there, so a NaN in the smaller-or-unordered operand never shows).
                    const float pr0 = row1[0];
                    const bool blank = pr0 > thr;  // :529
                    const float g_item = blank ? ladd<MODE>(lp, gp) + pr0 : kNegInf;
                    bool stay = collapse && tip >= 0;
                    float s_item = kNegInf;
                    if (stay) {
                        const float pt = row1[tip + 1];
                        stay = !(pt < thr);
                        if (stay) s_item = lp + pt;  // :541-544
                    }
                    bool inc = false, inc_first = false;
                    float c_item = kNegInf;
                    if (node >= 0) {
                        const int par = b_par[i];
                        for (int j = 0; j < B; ++j) {
                            if (b_node[j] == par) {
                                const float pl = frame1[b_state[j] * N + tip + 1];  // the PARENT's row
                                if (!(pl < thr)) {
                                    const bool rj = collapse && b_tip[j] == tip;
                                    const float lpj = b_lp[j], gpj = b_gp[j];
                                    c_item = rj ? gpj + pl : ladd<MODE>(lpj, gpj) + pl;
                                    inc = true;
                                    inc_first = j < i;  // the parent's entry comes earlier in the beam: its item was appended first
                                }
                                break;
                            }
                        }
                    }
                    bool have = false;
                    auto push = [&](float l_it, float g_it) {
                        if (!have) {
                            clp = l_it;
                            cgp = g_it;
                            have = true;
                        } else {
                            clp = ladd<MODE>(clp, l_it);
                            cgp = ladd<MODE>(cgp, g_it);
                        }
                    };
                    if (inc && inc_first) push(c_item, kNegInf);
                    if (blank) push(kNegInf, g_item);
                    if (stay) push(s_item, kNegInf);
                    if (inc && !inc_first) push(c_item, kNegInf);
                    valid = blank || stay || inc;
                    cid = node;
                } else {
                    const int l = k - 1;
                    const float pk = row1[k];
                    const bool pass = !(pk < thr);  // :537
                    rep = collapse && l == tip;
                    const float contrib = rep ? gp + pk : ladd<MODE>(lp, gp) + pk;
                    const int ch = b_child[i * NL + l];
                    const bool exists = ch >= 0;
                    valid = pass && (exists || !rep || gp > kNegInf);  // :546
                    if (valid && exists) {
                        for (int j = 0; j < B; ++j)
                            if (b_node[j] == ch) {
                                valid = false;
                                break;
                            }
                    }
                    is_new = valid && !exists;
                    clp = contrib;
                    cid = ch;
                }
            }
            const uint64_t m_new = ballot(is_new);
            FCD_DUPLEX_PHASE(2)
            if (prof) {
                n_newnodes += (uint32_t)popc64(m_new);
                n_iter += (uint32_t)(hi - lo + 1) * (uint32_t)((popc64(m_new) + 31) / 32);  // (max mode: rows, 16 nodes per pass)
            }
            if (MODE == FCD_LOGADD_MAX && !(staged && Wcap >= 8)) {
            // max-product mode has no transcendental in the recurrence: one lane per new node (the general form; with
            // resident windows the four-lanes-per-node loop below takes over)
            if (is_new) {
                cid = nn + popc64(m_new & lanemask_lt());
                if (cid < p.cap_nodes) {
                    // build_secondary_probs (:212-249): whole window [lo, hi), one node per lane
                    const int l = k - 1;
                    int p_off = 0, p_end = 0;
                    if (node >= 0 && !staged) {  // (staged: the parent's bounds are in LDS with its ring)
                        const int4 pm = load_meta_l2(&meta[node]);
                        p_off = pm.z;
                        p_end = pm.w;
                    }
                    const VecRef pv = node_vec(node, p_off, p_end);
                    float *my = vec + (int64_t)cid * Wcap * 3;
                    float l_lab = kNegInf, l_sum = kNegInf, mx = kNegInf;
                    if (staged) {
                        // operands of the NEXT row are requested before this row's arithmetic, slots advance
                        // incrementally: the parent's row t - 1 sits where this node's row t - 1 sits (same ring
                        // geometry), i.e. in the slot this loop has just left
                        const float *wq = L.w2 + state * N;                               // + row * S * N
                        const float *xq = ring(L.b_buf(cur)[i]) + (rep ? 1 : 2);          // the parent's resident ring
                        const int q_off = L.b_off(cur)[i], q_end = L.b_end(cur)[i];
                        const int rstep = S * N;
                        int s3 = 3 * (lo % Wcap);
                        float r0 = wq[0], rl1 = wq[l + 1];
                        float x = (lo - 1 >= q_off && lo - 1 < q_end) ? xq[3 * (((lo - 1) % Wcap + Wcap) % Wcap)] : kNegInf;
                        for (int j = 0; j < W; ++j) {
                            const int jn = j + 1 < W ? j + 1 : j;
                            const float r0n = wq[jn * rstep], rl1n = wq[jn * rstep + l + 1];
                            const float xn = (lo + j >= q_off && lo + j < q_end) ? xq[s3] : kNegInf;  // parent's row lo + j
                            const float g = l_sum + r0;
                            const float lb = rl1 + ladd<MODE>(l_lab, x);
                            const float sm = ladd<MODE>(lb, g);
                            store_row(my + s3, lb, g, sm);
                            mx = lmax(mx, sm);
                            l_lab = lb;
                            l_sum = sm;
                            s3 = s3 + 3 == 3 * Wcap ? 0 : s3 + 3;
                            r0 = r0n;
                            rl1 = rl1n;
                            x = xn;
                        }
                    } else {
                    int s = lo % Wcap;
                    for (int idx = lo; idx < hi; ++idx) {
                        float pg, ps;
                        const float *row = ln2 + ((int64_t)idx * S + state) * N;  // crf: tip.state (:772)
                        const float r0 = row[0], rl1 = row[l + 1];
                        vec_get(pv, idx - 1, Wcap, pg, ps);
                        const float g = l_sum + r0;
                        const float x = rep ? pg : ps;
                        const float lb = rl1 + ladd<MODE>(l_lab, x);
                        const float sm = ladd<MODE>(lb, g);
                        my[3 * s] = lb;
                        my[3 * s + 1] = g;
                        my[3 * s + 2] = sm;
                        mx = lmax(mx, sm);
                        l_lab = lb;
                        l_sum = sm;
                        if (++s == Wcap) s = 0;
                    }
                    }
                    meta[cid] = make_int4(node, l, lo, hi);
                    nmax[cid] = mx;
                    rlo[cid] = lo;
                    for (int j = 0; j < NL; ++j) rows[(int64_t)cid * NL + j] = -1;
                    if (node >= 0) rows[(int64_t)node * NL + l] = cid;
                    b_child[i * NL + l] = cid;
                    p2 = mx;
                }
            } else if (act && cid >= 0) {
                p2 = load_f32_l2(&nmax[cid]);  // :613-618 prob_2_max = data.max_prob (may be stale)
            }
            } else {
            // ---- new nodes: ids in creation order, then build_secondary_probs (:212-249) ----
            // The recurrence of one node is two interleaved serial chains:
            //   label_t = p_t[label] (x) (label_{t-1} (+) X_{t-1})            (needs only the label chain)
            //   sum_t   = label_t (+) (sum_{t-1} (x) p_t[blank])   [gap_t = sum_{t-1} (x) p_t[blank]]
            // so every new node gets a PAIR of lanes: the even lane runs the label chain, the odd
            // lane runs the sum chain one row behind it -- same operations in the same order as the
            // reference (exact), but one log-add per iteration instead of two.
            const int pre = popc64(m_new & lanemask_lt());
            const int n_new = popc64(m_new);
            if (is_new) {
                cid = nn + pre;
                L.bt[pre] = lane;
            }
            const bool can = is_new && cid < p.cap_nodes;
            int par_off = 0, par_end = 0;
            if (can && node >= 0 && !staged) {
                const int4 pm = load_meta_l2(&meta[node]);
                par_off = pm.z;
                par_end = pm.w;
            }
            __syncthreads();
            // lanes per new node: a pair (logsumexp: two log-add chains in lockstep) or a quad (max: see below)
            constexpr int LPN = MODE == FCD_LOGADD_MAX ? 4 : 2;
            constexpr int PER = 64 / LPN;
            for (int rd = 0; rd * PER < n_new; ++rd) {
                const int m = rd * PER + lane / LPN;
                const bool have = m < n_new;
                const int owner = have ? L.bt[m] : lane;
                const bool isA = (lane & 1) == 0;
                // the owner's parameters
                const int q_flags = __shfl((can ? 1 : 0) | (rep ? 2 : 0), owner);
                const bool work = have && (q_flags & 1);
                // (idle lanes run the loops too, on the tables of beam entry 0: every address they form is a real one)
                const int q_cid = __shfl(cid, owner);
                const int o_i = __shfl(i, owner), o_l = __shfl(k - 1, owner), o_state = __shfl(state, owner);
                const int q_i = work ? o_i : 0;
                const int q_l = work ? o_l : 0;
                const int q_state = work ? o_state : 0;
                const int q_node = __shfl(node, owner);
                const int q_poff = __shfl(par_off, owner);
                const int q_pend = __shfl(par_end, owner);
                const bool q_rep = (q_flags & 2) != 0;
                const VecRef pv = node_vec(q_node, q_poff, q_pend);
                float *my = vec + (int64_t)(work ? q_cid : 0) * Wcap * 3;
                float lb = kNegInf;   // A: label_{t-1};  B: label_{t'} received from A
                float sm = kNegInf;   // B: sum_{t'-1}
                float mx = kNegInf;
                if (MODE == FCD_LOGADD_MAX) {
                    // Max-product mode (staged, Wcap >= 8): no transcendental, so the loop is all bookkeeping -- and with
                    // one wavefront per SIMD every instruction costs 4-8 cycles whatever it does.  Four lanes per node:
                    // lane q of the quad owns rows q, q + 4, ... -- it fetches their operands, checks them, stores the
                    // finished row -- and the recurrence walks round the quad: in step u every lane computes
                    //   label = p[l] + max(label', X),  sum = max(label, sum' + p[blank])
                    // with label' / sum' taken from its LEFT neighbour (a DPP operand, quad_perm [3,0,1,2]: no extra
                    // instruction), which is the true row exactly on the diagonal lane q = u; what the other lanes
                    // compute in that step is never read.  Per four rows: one set of fetches, four 4-instruction
                    // steps, one select of the diagonal values, one store -- ~13 instructions per row instead of ~48.
                    // LogSpace::add's max flavour is v_max_f32 unless a NaN or a zero takes part (a <= b and "+ 0.0"
                    // see those differently): impossible while every operand is below zero (-inf included), which
                    // the log-probabilities are unless a posterior is 1, above 1 or NaN -- checked per group on the
                    // side, and a group that fails is redone with the exact form (its stores simply land twice).
                    const int qd = lane & 3;
                    const float *wq = L.w2 + q_state * N;
                    const float *xq = ring(L.b_buf(cur)[q_i]) + (q_rep ? 1 : 2);
                    const int q_off = L.b_off(cur)[q_i], q_end = L.b_end(cur)[q_i];
                    const unsigned q_span = q_end > q_off ? (unsigned)(q_end - q_off) : 0u;
                    const int rstep = S * N;
                    const int W3 = 3 * Wcap;
                    int jrow = qd;                               // this lane's row of the coming group
                    int s3 = 3 * slot_near(lo + qd, lo, lo_m, Wcap);  // its slot (x 3)
                    int sx = s3 == 0 ? W3 - 3 : s3 - 3;          // the slot of the row before: the parent's row it needs
                    const float *wlast = wq + (W > 0 ? W - 1 : 0) * rstep;
                    const float *wrow = wq + jrow * rstep;
                    float n0, nl, nx;
                    bool nv;
                    auto fetch = [&]() {
                        const float *wr = wrow < wlast ? wrow : wlast;  // (rows past the window: any valid address)
                        n0 = wr[0];
                        nl = wr[q_l + 1];
                        nx = xq[sx];  // (always inside the ring: read now, judge when the value is used)
                        nv = (unsigned)(lo + jrow - 1 - q_off) < q_span;
                    };
                    fetch();
                    float lab3 = kNegInf, sum3 = kNegInf;  // row - 1 of the coming group, valid in lane 3
                    const bool is0 = qd == 0, is1 = qd == 1, is2 = qd == 2;
                    bool exact_from_here = false;  // (wave-uniform)
                    auto rot = [](float v) {  // lane q <- lane q - 1 (mod 4)
                        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93 /* quad_perm [3,0,1,2] */, 0xf,
                                                                          0xf, true));
                    };
                    for (int j = 0; j < W; j += 4) {
                        const float c0 = n0, cl = nl, cx = nv ? nx : kNegInf;
                        const bool mine = work && jrow < W;
                        const int st3 = s3;
                        jrow += 4;
                        wrow += 4 * rstep;
                        s3 = s3 + 12 >= W3 ? s3 + 12 - W3 : s3 + 12;
                        sx = sx + 12 >= W3 ? sx + 12 - W3 : sx + 12;
                        if (j + 4 < W) fetch();
                        // (the incoming chain state counts as an operand too -- after a group that took the exact form it
                        // may be a NaN or a zero, and v_max_f32 would drop the NaN that LogSpace::add carries along -- so
                        // once a group has failed the check, every later group of this pass takes the exact form as well)
                        const bool special = mine && (!(cl < 0.0f) | !(c0 < 0.0f) | !(cx < 0.0f));
                        float lab0, lab1, lab2, g0, g1, g2, g3, sum0, sum1, sum2;
                        const float lab_in = lab3, sum_in = sum3;
                        lab0 = cl + vmax_raw(rot(lab_in), cx);
                        g0 = rot(sum_in) + c0;
                        sum0 = vmax_raw(lab0, g0);
                        lab1 = cl + vmax_raw(rot(lab0), cx);
                        g1 = rot(sum0) + c0;
                        sum1 = vmax_raw(lab1, g1);
                        lab2 = cl + vmax_raw(rot(lab1), cx);
                        g2 = rot(sum1) + c0;
                        sum2 = vmax_raw(lab2, g2);
                        lab3 = cl + vmax_raw(rot(lab2), cx);
                        g3 = rot(sum2) + c0;
                        sum3 = vmax_raw(lab3, g3);
                        exact_from_here = exact_from_here || ballot(special) != 0ull;
                        if (exact_from_here) {  // rare: the same four steps on LogSpace::add itself
                            lab0 = cl + ladd<MODE>(rot(lab_in), cx);
                            g0 = rot(sum_in) + c0;
                            sum0 = ladd<MODE>(lab0, g0);
                            lab1 = cl + ladd<MODE>(rot(lab0), cx);
                            g1 = rot(sum0) + c0;
                            sum1 = ladd<MODE>(lab1, g1);
                            lab2 = cl + ladd<MODE>(rot(lab1), cx);
                            g2 = rot(sum1) + c0;
                            sum2 = ladd<MODE>(lab2, g2);
                            lab3 = cl + ladd<MODE>(rot(lab2), cx);
                            g3 = rot(sum2) + c0;
                            sum3 = ladd<MODE>(lab3, g3);
                        }
                        // the diagonal: lane q keeps what step q produced
                        const float lbv = is0 ? lab0 : is1 ? lab1 : is2 ? lab2 : lab3;
                        const float gv = is0 ? g0 : is1 ? g1 : is2 ? g2 : g3;
                        const float smv = is0 ? sum0 : is1 ? sum1 : is2 ? sum2 : sum3;
                        if (mine) {
                            store_row(my + st3, lbv, gv, smv);
                            mx = lmax(mx, smv);
                        }
                    }
                    // the node's running maximum: over the quad's four partial maxima (ln never yields -0 and sums of
                    // non-positive terms never do either, so the order of the comparisons cannot show)
                    mx = lmax(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xf, 0xf, true)));
                    mx = lmax(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xf, 0xf, true)));
                } else if (staged) {
                    LogAddCoef K = logadd_coef();
                    if (PIN) {
                        FCD_OPAQUE_V(K.log2e); FCD_OPAQUE_V(K.ln2hi); FCD_OPAQUE_V(K.ln2lo); FCD_OPAQUE_V(K.two);
#pragma unroll
                        for (int u = 0; u < 12; ++u) FCD_OPAQUE_V(K.e[u]);
#pragma unroll
                        for (int u = 0; u < 15; ++u) FCD_OPAQUE_V(K.a[u]);
                    }
                    // The loop every new node runs W + 1 times -- 80 % of the kernel in logsumexp mode -- kept lean:
                    // operands of the NEXT row are requested before this row's log-add, ring slots and tile
                    // addresses advance incrementally, and label_t reaches the odd lane through a DPP move
                    // (lane pairs are adjacent) instead of an LDS round trip.
                    const float *wq = L.w2 + q_state * N + (isA ? q_l + 1 : 0);          // + row * S * N
                    // X_{t-1}: the parent's gap (repeat) or label(+)gap at row t - 1, out of its resident ring; rows
                    // outside the parent's window [q_off, q_end) read as zero (SecondaryProbs::get :167-179)
                    const float *xq = ring(L.b_buf(cur)[q_i]) + (q_rep ? 1 : 2);
                    const int q_off = L.b_off(cur)[q_i], q_end = L.b_end(cur)[q_i];
                    const int rstep = S * N;
                    int jn = isA ? 0 : -1;                      // the row this lane handles in the coming iteration
                    int slot3 = 3 * slot_near(lo + jn, lo, lo_m, Wcap);  // 3 * ((lo + jn) mod Wcap)
                    // loop-invariant forms of the per-row tests: a lane works on row j iff (unsigned)j < wlim; the parent
                    // holds row lo + j iff (unsigned)(j - jv0) < jspan (the even lane asks for X of the NEXT row, which
                    // sits in the slot this lane's own row occupies: same ring geometry)
                    const unsigned wlim = work ? (unsigned)W : 0u;
                    const int jv0 = q_off - lo;
                    const unsigned jspan = isA && q_end > q_off ? (unsigned)(q_end - q_off) : 0u;
                    const float *wnext = wq + (isA ? rstep : 0);  // the coefficient of the row after the coming one
                    float c_cur = 0.0f, x_cur = kNegInf;
                    if (isA) {
                        c_cur = wq[0];
                        x_cur = (lo - 1 >= q_off && lo - 1 < q_end) ? xq[3 * slot_near(lo - 1, lo, lo_m, Wcap)] : kNegInf;
                    }
                    for (int sidx = 0; sidx <= W; ++sidx) {
                        // A works on row t = lo + sidx (if sidx < W); B on row t' = lo + sidx - 1 (if sidx >= 1)
                        const bool on = (unsigned)jn < wlim;
                        // next row's operands (one row past the tile at the very end: inside LDS, value unused)
                        const float c_nxt = *wnext;
                        const float x_raw = xq[slot3];  // (always inside the ring: read first, judge afterwards -- no branch)
                        const float x_nxt = (unsigned)(jn - jv0) < jspan ? x_raw : kNegInf;      // the parent's row lo + jn
                        // (lanes without work carry -inf everywhere; the one idle row of a working lane -- the odd lane's
                        // first, the even lane's last -- computes a value nobody keeps)
                        const float bb = isA ? x_cur : sm + c_cur;                    // A: X_{t-1};  B: gap_{t'}
                        const float v = ladd_lockstep<MODE>(lb, bb, K);
                        // The odd lane holds the whole row t' -- label_{t'} came from the even lane -- and writes it
                        // with one 12-byte store; everything else is a select.
                        if (on && !isA) store_row(my + slot3, lb, bb, v);  // {label_{t'}, gap_{t'}, sum_{t'}}
                        // No selects on `on`: the odd lane's idle first row yields -inf (-inf (+) -inf), which changes
                        // neither sum nor maximum; the even lane never reads sum / maximum; and what the odd lane puts
                        // into label_t is overwritten by the even lane's through the DPP move below.
                        sm = v;
                        mx = lmax(mx, v);
                        const float lb_out = c_cur + v;  // label_t (even lane)
                        // hand label_t to the odd lane for the next iteration; the even lane keeps it
                        lb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(lb_out), 0xA0 /* quad_perm [0,0,2,2] */,
                                                                        0xf, 0xf, false));
                        c_cur = c_nxt;
                        x_cur = x_nxt;
                        ++jn;
                        wnext += rstep;
                        slot3 = slot3 + 3 == 3 * Wcap ? 0 : slot3 + 3;
                    }
                } else {
                for (int sidx = 0; sidx <= W; ++sidx) {
                    // A works on row t = lo + sidx (if sidx < W); B on row t' = lo + sidx - 1 (if sidx >= 1)
                    const int j = isA ? sidx : sidx - 1;
                    const bool on = work && j >= 0 && j < W;
                    float a = kNegInf, bb = kNegInf, add_after = 0.0f, r0 = 0.0f;
                    if (on) {
                        if (isA) {
                            float pg, ps, rl1;
                            rl1 = ln2[((int64_t)(lo + j) * S + q_state) * N + q_l + 1];
                            vec_get(pv, lo + j - 1, Wcap, pg, ps);
                            a = lb;
                            bb = q_rep ? pg : ps;
                            add_after = rl1;
                        } else {
                            r0 = ln2[((int64_t)(lo + j) * S + q_state) * N];
                            a = lb;          // label_{t'} from the even lane
                            bb = sm + r0;    // gap_{t'}
                        }
                    }
                    const float v = ladd<MODE>(a, bb);
                    float lb_out = lb;
                    if (on) {
                        const int slot = (lo + j) % Wcap;
                        if (isA) {
                            lb_out = add_after + v;  // label_t
                            my[3 * slot] = lb_out;
                        } else {
                            my[3 * slot + 1] = bb;   // gap_{t'}
                            my[3 * slot + 2] = v;    // sum_{t'}
                            sm = v;
                            mx = lmax(mx, v);
                        }
                    }
                    // hand label_t to the odd lane for the next iteration; the even lane keeps it
                    const float from_even = __shfl(lb_out, lane & ~1);
                    lb = isA ? lb_out : from_even;
                }
                }
                if (work && (LPN == 2 ? !isA : (lane & 3) == 0)) {
                    meta[q_cid] = make_int4(q_node, q_l, lo, hi);
                    nmax[q_cid] = mx;
                    rlo[q_cid] = lo;
                    for (int jj = 0; jj < NL; ++jj) rows[(int64_t)q_cid * NL + jj] = -1;
                    if (q_node >= 0) rows[(int64_t)q_node * NL + q_l] = q_cid;
                }
                // the owner slot needs the new node's running maximum as its prob_2_max
                const int pair_b = LPN * (pre % PER) + (LPN == 2 ? 1 : 0);
                const float got = __shfl(mx, pair_b);
                if (can && pre / PER == rd) p2 = got;
            }
            if (can) b_child[i * NL + (k - 1)] = cid;
            if (!is_new && act && cid >= 0) {
                p2 = load_f32_l2(&nmax[cid]);  // :613-618 prob_2_max = data.max_prob (may be stale)
            }
            __syncthreads();
            }
            nn += popc64(m_new);
            FCD_DUPLEX_PHASE(3)
            const float prob = ladd<MODE>(clp, cgp) + p2;  // :146-148
            if (act) {
                L.c_lp[c] = clp;
                L.c_gp[c] = cgp;
                L.c_id[c] = cid;
                L.c_new[c] = is_new ? 1 : 0;
                L.c_key[c] = valid ? (prob == prob ? make_key(prob, cid) : 1ull) : 0ull;
            }
            n_valid += popc64(__ballot(valid));
            any_nan = any_nan || (__ballot(valid && prob != prob) != 0ull);
        }
        FCD_DUPLEX_PHASE(2)
        if (nn > p.cap_nodes) return fail(FCD_ST_INTERNAL);
        if (n_valid >= 2 && any_nan) return fail(FCD_ST_INCOMPARABLE);  // :619-631
        if (n_valid == 0) return fail(FCD_ST_RAN_OUT_OF_BEAM);          // :633-636
        __syncthreads();

        // ---- rank and build the next beam (no renormalisation in log space) ----
        const int nxt = cur ^ 1;
        const int Bn = n_valid < BC ? n_valid : BC;
        const bool pdq = p.tie_order == FCD_TIE_PDQ178;  // equal probabilities above 20 candidates: Rust 1.78's order
        bool any_kept_tie = false;
        if (count_amb || (pdq && n_valid > 20)) {
            // candidates with one probability occupy ranks [gt, gt + eq): a kept one is tied when gt < BC and
            // eq >= 2; the tie can change the kept set or the best entry when the group holds rank 0 or straddles
            // the truncation boundary
            bool kept_tie = false, crit = false;
            for (int base = 0; base < nslots; base += kWave) {
                const int c = base + lane;
                const uint64_t key = c < nslots ? L.c_key[c] : 0ull;
                if (key == 0ull) continue;
                const uint32_t ph = (uint32_t)(key >> 32);
                int gt = 0, eq = 0;
                for (int j = 0; j < nslots; ++j) {
                    const uint64_t kj = L.c_key[j];
                    if (kj == 0ull) continue;
                    const uint32_t pj = (uint32_t)(kj >> 32);
                    gt += pj > ph ? 1 : 0;
                    eq += pj == ph ? 1 : 0;
                }
                kept_tie = kept_tie || (gt < BC && eq >= 2);
                crit = crit || (eq >= 2 && (gt == 0 || (gt < BC && gt + eq > BC)));
            }
            any_kept_tie = n_valid > 20 && ballot(kept_tie) != 0ull;
            if (count_amb) {
                if (any_kept_tie) ++n_amb;
                if (ballot(crit) != 0ull) ++n_crit;
            }
        }
        // candidate slot c becomes entry `rank` of the next beam
        auto emit = [&](int c, int rank) {
            const int i = c / N, k = c - i * N;
            L.b_node(nxt)[rank] = L.c_id[c];
            L.b_lp(nxt)[rank] = L.c_lp[c];
            L.b_gp(nxt)[rank] = L.c_gp[c];
            if (k == 0) {
                L.b_tip(nxt)[rank] = b_tip[i];
                L.b_par(nxt)[rank] = b_par[i];
                L.b_state(nxt)[rank] = b_state[i];
                // the entry stays: so does its resident window
                L.b_buf(nxt)[rank] = L.b_buf(cur)[i];
                L.b_off(nxt)[rank] = L.b_off(cur)[i];
                L.b_end(nxt)[rank] = L.b_end(cur)[i];
                L.b_rlo(nxt)[rank] = L.b_rlo(cur)[i];
                L.b_max(nxt)[rank] = L.b_max(cur)[i];
            } else {
                L.b_buf(nxt)[rank] = -1;  // a node entering the beam: gets a buffer and its ring below
                L.b_tip(nxt)[rank] = k - 1;
                L.b_par(nxt)[rank] = b_node[i];
                L.b_state(nxt)[rank] = crf ? (int)(((int64_t)b_state[i] * NL) % S) + (k - 1) : 0;  // :782
            }
            L.nb_src[rank] = c | (L.c_new[c] << 30);
        };
        if (pdq && any_kept_tie) {
            // sort_unstable_by's own order (src/duplex.rs:620,807): the merged candidates in ascending node order go
            // through the restated quicksort (one lane); the first BC of its result are the next beam
            for (int base = 0; base < nslots; base += kWave) {
                const int c = base + lane;
                const uint64_t key = c < nslots ? L.c_key[c] : 0ull;
                if (key == 0ull) continue;
                int pos = 0;
                for (int j = 0; j < nslots; ++j) {
                    const uint64_t kj = L.c_key[j];
                    pos += (kj != 0ull && (uint32_t)kj > (uint32_t)key) ? 1 : 0;  // low word: larger = smaller node
                }
                L.pq_list[pos] = (key & 0xFFFFFFFF00000000ull) | (uint32_t)c;
            }
            __syncthreads();
            if (lane == 0) pdq178::sort_desc(L.pq_list, n_valid, L.pq_scr);
            __syncthreads();
            for (int rank = lane; rank < Bn; rank += kWave) emit((int)(uint32_t)L.pq_list[rank], rank);
        } else {
            for (int base = 0; base < nslots; base += kWave) {
                const int c = base + lane;
                if (c >= nslots) continue;
                const uint64_t key = L.c_key[c];
                if (key == 0ull) continue;
                int rank = 0;
                for (int j = 0; j < nslots; ++j) rank += (L.c_key[j] > key) ? 1 : 0;
                if (rank < BC) emit(c, rank);
            }
        }
        __syncthreads();
        FCD_SUB_BEGIN()
        for (int item = lane; item < Bn * NL; item += kWave) {
            const int s = item / NL, l = item - s * NL;
            const int src = L.nb_src[s];
            const int c = src & 0x3FFFFFFF;
            const int i = c / N, k = c - i * N;
            int v;
            if (k == 0) v = b_child[i * NL + l];
            else if (src >> 30) v = -1;
            else v = load_i32_l2(&rows[(int64_t)L.b_node(nxt)[s] * NL + l]);
            L.b_child(nxt)[s * NL + l] = v;
        }
        FCD_SUB(3)
        // a state outside [0, S) is an ndarray index panic in the reference when the entry is next
        // expanded (:749) -- which never happens for the entries the last row leaves behind
        if (crf && t1 + 1 < T1) {
            bool bs = false;
            for (int s2 = lane; s2 < Bn; s2 += kWave) bs = bs || L.b_state(nxt)[s2] >= S;
            if (ballot(bs) != 0ull) return fail(FCD_ST_BAD_STATE);
        }
        FCD_SUB_BEGIN()
        if (resident) {
            // ---- hand the LDS buffers of the entries that left to the nodes that entered, and bring their rings in ----
            for (int j = lane; j < BC; j += kWave) L.s_off[j] = 0;
            __syncthreads();
            const bool in_next = lane < Bn;
            const int mybuf = in_next ? L.b_buf(nxt)[lane] : 0;
            if (in_next && mybuf >= 0) L.s_off[mybuf] = 1;  // stays taken
            __syncthreads();
            const bool is_free = lane < BC && L.s_off[lane] == 0;
            const uint64_t free_m = ballot(is_free);
            if (is_free) L.s_end[popc64(free_m & lanemask_lt())] = lane;  // the free list
            const bool entering = in_next && mybuf < 0;
            const uint64_t enter_m = ballot(entering);
            __syncthreads();
            if (entering) L.b_buf(nxt)[lane] = L.s_end[popc64(enter_m & lanemask_lt())];
            __syncthreads();
            // bounds and maxima of ALL entering nodes (each on its own lane) and their rings, two nodes at a time, in
            // the same trip to the arena: the loads of the bounds are issued first and used last
            int4 m4 = make_int4(0, 0, 0, 0);
            float e_mx = kNegInf;
            int e_rl = 0;
            if (entering) {
                const int nd = L.b_node(nxt)[lane];
                m4 = load_meta_l2(&meta[nd]);
                e_mx = load_f32_l2(&nmax[nd]);
                e_rl = load_i32_l2(&rlo[nd]);
            }
            if (prof) n_enter += (uint32_t)popc64(enter_m);
            for (uint64_t m = enter_m; m != 0ull;) {
                const int ea = (int)__builtin_ctzll(m);
                m &= m - 1;
                const int eb = m ? (int)__builtin_ctzll(m) : -1;
                if (eb >= 0) m &= m - 1;
                ring_copy_whole2(vec + (int64_t)L.b_node(nxt)[ea] * Wcap * 3, ring(L.b_buf(nxt)[ea]),
                                 eb >= 0 ? vec + (int64_t)L.b_node(nxt)[eb] * Wcap * 3 : nullptr,
                                 eb >= 0 ? ring(L.b_buf(nxt)[eb]) : nullptr, 3 * Wcap, lane);
            }
            FCD_SUB(1)
            if (entering) {
                L.b_off(nxt)[lane] = m4.z;
                L.b_end(nxt)[lane] = m4.w;
                L.b_max(nxt)[lane] = e_mx;
                L.b_rlo(nxt)[lane] = e_rl;
            }
        }
        B = Bn;
        cur = nxt;
        __syncthreads();
        FCD_DUPLEX_PHASE(4)
    }
    if (prof && lane == 0) {
        uint32_t *o = p.prof + 16 * r;
        for (int k = 0; k < 5; ++k) o[k] = (uint32_t)(acc[k] >> 6);  // units of 64 cycles: a pair runs ~2e8 cycles
        o[5] = n_iter;
        o[6] = n_newnodes;
        o[7] = (uint32_t)T1;
        o[8] = n_slow;   // steps whose extension took the sequential path
        o[9] = n_enter;  // nodes that entered the beam (their rings were copied into LDS)
        o[10] = n_ext;   // steps in which the envelope's upper bound grew
        for (int k = 0; k < 5; ++k) o[11 + k] = (uint32_t)(sub[k] >> 6);  // 11 read-2 tile, 12 hand-over + entering nodes, 13 update_max rescans, 14 child rows, 15 beam sort + parents' bounds
    }

    // ---- labels leaf -> root (:638-649), written in sequence order ----
    if (lane == 0) {
        int node = L.b_node(cur)[0];
        int n = 0;
        for (int q = node; q >= 0; q = load_i32_l2(reinterpret_cast<const int32_t *>(&meta[q]))) ++n;
        for (int j = n - 1; j >= 0; --j) {
            const int4 m = load_meta_l2(&meta[node]);
            lab_out[j] = (uint8_t)(m.y + 1);
            node = m.x;
        }
        p.out.out_len[r] = (uint32_t)n;
        p.out.status[r] = FCD_ST_OK;
        if (count_amb) {
            p.out.ambiguous[2 * r] = (uint32_t)n_amb;
            p.out.ambiguous[2 * r + 1] = (uint32_t)n_crit;
        }
    }
}

// ---- prepass: ln of the posteriors into contiguous log-space copies (:452-453) ----
__global__ void ln_convert_kernel(const float *x, int dtype, int64_t n_reads, int64_t T, int S, int N,
                                  int64_t s_read, int64_t s_t, int64_t s_s, int64_t s_n, float *out, int glibc235) {
    const int64_t total = n_reads * T * S * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx % N, st = (idx / N) % S, t = (idx / ((int64_t)N * S)) % T,
                      r = idx / ((int64_t)N * S * T);
        const float v = load_post(x, r * s_read + t * s_t + st * s_s + j * s_n, dtype);
        out[idx] = glibc235 ? g235::logf235(v) : ln_cr(v);  // LogSpace::new (:24-26)
    }
}

// test hook: out[i] = LogSpace::add(a[i], b[i]) and ln(a[i]) exactly as the duplex kernels compute them.  mode: the
// log-add flavour (FCD_LOGADD_*), + 4 for the form the window-building loop uses (ladd_lockstep: every lane of the
// wavefront in step, shortcuts folded into selects) instead of the general one.
__global__ void logspace_probe_kernel(const float *a, const float *b, float *out_add, float *out_ln,
                                      int64_t n, int mode) {
    const LogAddCoef K = logadd_coef();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t trips = (n + stride - 1) / stride;  // the same for every lane: the lockstep form votes
    for (int64_t k = 0; k < trips; ++k) {
        const int64_t i = k * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = i < n;
        const float x = valid ? a[i] : -1.0f, y = valid ? b[i] : -2.0f;
        float r;
        const int flavour = mode & 3;
        if (flavour == FCD_LOGADD_LOGSUMEXP_GLIBC235)
            r = (mode & 4) ? ladd_lockstep<FCD_LOGADD_LOGSUMEXP_GLIBC235>(x, y, K) : ladd<FCD_LOGADD_LOGSUMEXP_GLIBC235>(x, y);
        else if (mode & 4)
            r = flavour == FCD_LOGADD_MAX ? ladd_lockstep<FCD_LOGADD_MAX>(x, y, K) : ladd_lockstep<FCD_LOGADD_LOGSUMEXP>(x, y, K);
        else
            r = flavour == FCD_LOGADD_MAX ? ladd<FCD_LOGADD_MAX>(x, y) : ladd<FCD_LOGADD_LOGSUMEXP>(x, y);
        if (valid) {
            out_add[i] = r;
            out_ln[i] = flavour == FCD_LOGADD_LOGSUMEXP_GLIBC235 ? g235::logf235(x) : ln_cr(x);
        }
    }
}

// developer instrument: the DEPENDENT latency of one LogSpace::add as the window-building loop evaluates it
// (ladd_lockstep, coefficients in registers): every lane folds n_chain values into an accumulator, each add
// waiting for the previous one; cycles[lane] = shader cycles of the whole chain.  The duplex searches are bound by
// this latency (a tree node's window is W sequential rows), not by HBM or by the binary64 issue rate.
template <int MODE>
__global__ __launch_bounds__(64) void logadd_chain_kernel(int n_chain, uint64_t *cycles, float *sink) {
    const LogAddCoef K = logadd_coef();
    float acc = -1.0f - 0.001f * (float)threadIdx.x;
    const float v = -1.25f;
    uint64_t t0 = 0, t1 = 0;
    int dep = (int)__float_as_uint(acc);
    FCD_STAMP(t0, dep);
    for (int i = 0; i < n_chain; ++i) acc = ladd_lockstep<MODE>(acc, v, K) - 0.125f;  // stays on the full path
    dep = (int)__float_as_uint(acc);
    FCD_STAMP(t1, dep);
    cycles[threadIdx.x] = t1 - t0;
    sink[threadIdx.x] = acc;
}

// Exhaustive check of the fast paths ON THE DEVICE (the host verifier covers the host build of logadd_fast.h; the
// device build takes the hardware reciprocal in ln_1p): every f32 bit pattern in [first, last] of one domain --
// which = 0: exp on [-86, -0], 1: ln_1p on [2^-126, 1].  Wherever Ziv's test trusts the fast binary64 value, its f32
// rounding must equal the library routine's; counts[0] = arguments, [1] = sent to the slow path, [2] = mismatches.
__global__ void logadd_sweep_kernel(int which, uint32_t first, uint32_t last, unsigned long long *counts) {
    const LogAddCoef K = logadd_coef();
    unsigned long long n = 0, slow = 0, bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= (uint64_t)last; u += stride) {
        const float x = __uint_as_float((uint32_t)u);
        ++n;
        if (which == 0) {
            const double y = exp_fast((double)x, K);
            if (round_to_f32_unsafe(y)) ++slow;
            else if (__float_as_uint((float)y) != __float_as_uint((float)exp((double)x)) ||
                     round_to_f32_as_f64(y) != (double)(float)y) ++bad;
        } else {
            const double y = log1p_fast((double)x, K);
            if (round_to_f32_unsafe(y)) ++slow;
            else if (__float_as_uint((float)y) != __float_as_uint((float)log1p((double)x))) ++bad;
        }
    }
    atomicAdd(&counts[0], n);
    atomicAdd(&counts[1], slow);
    atomicAdd(&counts[2], bad);
}

// widest clamped envelope row over the whole batch -> *out (int), for sizing the rings
__global__ void env_width_kernel(const uint64_t *env, int64_t n_pairs, int64_t env_stride,
                                 int64_t T1cap, int64_t T2cap, const int64_t *len1,
                                 const int64_t *len2, int *out) {
    int best = 0;
    const int64_t total = n_pairs * T1cap;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / T1cap, t = idx % T1cap;
        int64_t T1 = T1cap, T2 = T2cap;
        if (len1) { int64_t v = len1[r]; T1 = v < 0 ? 0 : (v < T1 ? v : T1); }
        if (len2) { int64_t v = len2[r]; T2 = v < 0 ? 0 : (v < T2 ? v : T2); }
        if (t >= T1) continue;
        const uint64_t lo = env[(r * env_stride + t) * 2], hi_u = env[(r * env_stride + t) * 2 + 1];
        const uint64_t hi = hi_u > (uint64_t)T2 ? (uint64_t)T2 : hi_u;
        if (hi > lo) {
            const uint64_t w = hi - lo;
            best = max(best, (int)(w > 0x3FFFFFFFull ? 0x3FFFFFFFull : w));
        }
    }
    atomicMax(out, best);
}

}  // namespace

size_t duplex_lds_bytes(int beam_size, int N, int Wmax, int S, int tie_order) {
    return dlds_words(beam_size, N, Wmax, S, tie_order == FCD_TIE_PDQ178 && (int64_t)beam_size * N > 20) * 4 + 16;
}

hipError_t launch_ln_convert(const float *x, int dtype, int64_t n_reads, int64_t T, int S, int N, int64_t s_read,
                             int64_t s_t, int64_t s_s, int64_t s_n, float *out, int glibc235, hipStream_t stream) {
    const int64_t total = n_reads * T * S * N;
    if (total <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(ln_convert_kernel, dim3(blocks), dim3(256), 0, stream, x, dtype, n_reads, T, S, N,
                       s_read, s_t, s_s, s_n, out, glibc235);
    return hipGetLastError();
}

hipError_t launch_env_width(const uint64_t *env, int64_t n_pairs, int64_t env_stride, int64_t T1cap,
                            int64_t T2cap, const int64_t *len1, const int64_t *len2, int *out,
                            hipStream_t stream) {
    const int64_t total = n_pairs * T1cap;
    if (total <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(env_width_kernel, dim3(blocks), dim3(256), 0, stream, env, n_pairs,
                       env_stride, T1cap, T2cap, len1, len2, out);
    return hipGetLastError();
}

namespace {
__global__ void glibc235_apply_kernel(int which, const float *x, float *y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = which == 0 ? g235::expf235(x[i]) : (which == 1 ? g235::logf235(x[i]) : g235::log1pf235(x[i]));
}
}  // namespace

hipError_t launch_glibc235_apply(int which, const float *x, float *y, int64_t n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(glibc235_apply_kernel, dim3(blocks), dim3(256), 0, stream, which, x, y, n);
    return hipGetLastError();
}

hipError_t launch_logspace_probe(const float *a, const float *b, float *out_add, float *out_ln,
                                 int64_t n, int mode, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(logspace_probe_kernel, dim3(blocks), dim3(256), 0, stream, a, b, out_add, out_ln,
                       n, mode);
    return hipGetLastError();
}

hipError_t launch_logadd_sweep(int which, uint32_t first, uint32_t last, unsigned long long *counts, hipStream_t stream) {
    hipLaunchKernelGGL(logadd_sweep_kernel, dim3(256 * 16), dim3(256), 0, stream, which, first, last, counts);
    return hipGetLastError();
}

hipError_t launch_logadd_chain(int n_chain, int mode, uint64_t *cycles, float *sink, hipStream_t stream) {
    if (mode == FCD_LOGADD_MAX)
        hipLaunchKernelGGL(logadd_chain_kernel<FCD_LOGADD_MAX>, dim3(1), dim3(64), 0, stream, n_chain, cycles, sink);
    else
        hipLaunchKernelGGL(logadd_chain_kernel<FCD_LOGADD_LOGSUMEXP>, dim3(1), dim3(64), 0, stream, n_chain, cycles, sink);
    return hipGetLastError();
}

hipError_t launch_duplex(const DuplexArgs &a, int64_t pair_begin, int64_t n_pairs,
                         hipStream_t stream) {
    if (n_pairs <= 0) return hipSuccess;
    DuplexParams p;
    p.ln1 = a.ln1; p.ln2 = a.ln2; p.T1cap = a.T1cap; p.T2cap = a.T2cap;
    p.len1 = a.len1; p.len2 = a.len2; p.env = a.env; p.env_stride = a.env_stride;
    p.N = a.N; p.beam_size = a.beam_size; p.thr_ln = a.thr_ln; p.collapse = a.collapse;
    p.mode = a.mode; p.meta = a.meta; p.nmax = a.nmax; p.rows = a.rows; p.vec = a.vec;
    p.rootgap = a.rootgap; p.cap_nodes = a.cap_nodes; p.Wcap = a.Wcap; p.out = a.out;
    p.rlo = a.rlo; p.staged = a.staged;
    p.S = a.S; p.crf = a.crf; p.init1 = a.init1; p.init2 = a.init2; p.n_init1 = a.n_init1;
    p.n_init2 = a.n_init2; p.init1_stride = a.init1_stride; p.init2_stride = a.init2_stride;
    p.pair_begin = pair_begin;
    p.prof = a.prof;
    p.tie_order = a.tie_order;
    const size_t lds = duplex_lds_bytes(a.beam_size, a.N, a.staged ? a.Wcap - 2 : 0, a.S, a.tie_order);
    // up to two wavefronts per SIMD (2048 pairs on the 256 CUs): the coefficient-pinning instantiation
    const bool pin = a.staged && n_pairs <= 2048;
    if (a.mode == FCD_LOGADD_LOGSUMEXP_GLIBC235)
        hipLaunchKernelGGL((duplex_kernel<FCD_LOGADD_LOGSUMEXP_GLIBC235, false>), dim3((unsigned)n_pairs), dim3(64), lds,
                           stream, p);
    else if (a.mode == FCD_LOGADD_MAX)
        hipLaunchKernelGGL((duplex_kernel<FCD_LOGADD_MAX, false>), dim3((unsigned)n_pairs), dim3(64), lds,
                           stream, p);
    else if (pin)
        hipLaunchKernelGGL((duplex_kernel<FCD_LOGADD_LOGSUMEXP, true>), dim3((unsigned)n_pairs), dim3(64),
                           lds, stream, p);
    else
        hipLaunchKernelGGL((duplex_kernel<FCD_LOGADD_LOGSUMEXP, false>), dim3((unsigned)n_pairs), dim3(64),
                           lds, stream, p);
    return hipGetLastError();
}

}  // namespace fcd
