// duplex_slots.hip -- 2D pair-consensus beam search, duplex::beam_search / duplex::crf_beam_search
// (/root/reference/src/duplex.rs:443-650, :652-834), one read PAIR per wavefront: the SLOT-RESIDENT kernel (r06).
//
// Same search, same arithmetic and the same results as duplex.hip's kernel (which stays the any-shape fallback); what
// changed is where things live, so that a time step costs its window builds and little else:
//
//  * SLOTS.  beam_size * N <= 64.  Every tree node that is "live" in a step -- the beam entries and the nodes the step
//    creates -- owns one of P = beam_size * N slots for as long as it stays live: an LDS ring with its forward window
//    over read 2 and one word per field (node, label, parent, bounds, running maximum, children ...) in LDS arrays
//    indexed by slot.  Re-ranking the beam moves three words per entry (slot, label / gap probability); a new node
//    that survives the prune keeps the slot it was built in; nothing is copied when the beam changes.
//  * WINDOWS ARE ONE FLOAT PER ROW.  A window row of the reference is (label, gap) (ProbPair, :82-150).  Children read a
//    parent's label (+) gap, or -- the repeated label -- its gap alone (:234-240); a node's own continuation needs the
//    last row's label only.  gap_t = (label_{t-1} (+) gap_{t-1}) (x) blank_t (:232) is recomputed from the stored sum
//    of row t - 1 where it is needed (bit-identical: the same f32 addition on the same operands), so a ring holds
//    label (+) gap per row, the last label is one field, and `vfrom` remembers the row whose predecessor was "zero".
//    Rings hold Wcap4 >= widest envelope + 4 rows, slot = row mod Wcap4; the reference's discard_until (:181-191) is an
//    offset update.
//  * HBM IS WRITTEN ON EVICTION ONLY.  A node's ring and record go to its arena entry when it stops being live (a
//    new node the prune drops at once, a beam entry that falls out) -- it may come back as the child of a beam entry
//    (:546-566), and then ring and record are read back into a free slot.  12 bytes per window row written through at
//    build time became 4 bytes written once: config 5's 23 GB of writes -> ~7 GB.
//  * READ 2 IS A RING TOO.  The rows of read 2 inside the envelope live in LDS transposed ([state * N + label][row mod
//    Wcap4]); a step loads the rows the envelope gained (asked for one step ahead), not the window.
//  * The per-step bookkeeping is wave-scope: no __syncthreads(), cross-lane traffic by ds_bpermute / v_readlane,
//    exact rank by counting 64-bit keys, ties looked for in the ranked probability words.
//  * Window builds.  logsumexp: a pair of lanes per new node (label chain / sum chain, as before) on a BRANCH-FREE
//    log-add: the two "Ziv test failed, take the library routine" exits of LogSpace::add (1e-6 of the arguments) are
//    accumulated in a flag and the whole pass is redone on the exact routine when any lane raised it -- the dependent
//    chain no longer carries two divergent branches per row.  max: one lane per node, four rows per trip (16-byte LDS
//    reads of coefficients and parent rows, one 16-byte store), LogSpace::add's max flavour as v_max_f32 -- which
//    differs from it only when the label chain holds a NaN (the FIRST operand of (+); duplex.rs:57-61): detected per
//    row, and such a pass is redone on the exact routine.
//
// Everything else follows duplex.hip line for line in meaning: envelope check (:485-488), sort by node + extension
// (:490-522, :338-387) with the reference's quirks (stale windows of re-entering nodes, the extension's repeat test
// without collapse_repeats :512, assert!(current_end < upper_bound) -> FCD_ST_BAD_STATE), expansion (:526-593), merge in
// the reference's order (max mode's (+) is not commutative once a NaN takes part), prob_2_max refresh (:613-618), NaN
// check, sort_unstable_by's order of equal probabilities above 20 candidates (pdq178.h), truncate.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "device_utils.h"
#include "fcd_internal.h"
#define FCD_PDQ178_FORM0_ONLY 1  // (pdq178.h: the duplex kernels replay the default std form only)
#include "pdq178.h"
#include "duplex_math.h"

namespace fcd {

namespace {

constexpr int kSlots = 64;   // slot-indexed LDS arrays have 64 entries whatever P is: field offsets are immediates
constexpr int kNLMax = 7;    // N <= 8
constexpr int kBeamMax = 32; // beam_size * N <= 64 with N >= 2

// slot fields, in 16-byte GROUPS (one 64-entry int4 array per group: a slot's four words of a group move with one
// ds_read_b128 / ds_write_b128; field f is word f & 3 of group f >> 2)
enum {
    F_NODE = 0, F_TIP, F_PAR, F_STATE,       // group 0 (G_ID)
    F_OFF, F_END, F_VFROM, F_XREP,           // group 1 (G_WIN): the window's bounds, where its run of rows began, the :512 flag
    F_MX, F_LLAB, F_DEPTH, F_LSUM,           // group 2 (G_VAL): running maximum, last row's label, depth in the tree, last row's sum
    F_POFF, F_PEND, F_PVFROM, F_SPARE1,      // group 3 (G_PAR): the parent's window bounds as they were when it left the beam
    F_CHILD0,                                // groups 4, 5 (G_KID, G_KID + 1): kNLMax child ids (+ one spare word)
    F_COUNT = F_CHILD0 + kNLMax + 1
};
enum { G_ID = 0, G_WIN = 1, G_VAL = 2, G_PAR = 3, G_KID = 4, G_COUNT = 6 };
static_assert(F_COUNT == 4 * G_COUNT, "fields come in groups of four");

struct SlotParams {
    const float *ln1, *ln2;   // log-space posteriors, [pair][Tcap][S][N] contiguous
    int64_t T1cap, T2cap;
    const int64_t *len1, *len2;
    const uint64_t *env;
    int64_t env_stride;
    int N, beam_size;
    float thr_ln;
    int collapse, S, crf;
    const float *init1, *init2;
    int64_t n_init1, n_init2, init1_stride, init2_stride;
    // arena: ONE slab per pair, below 4 GiB (32-bit byte offsets: base in scalar registers + one vector register):
    //   meta  int4 per node {parent, label, off, end}
    //   aux   int4 per node {running maximum, vfrom, last row's label, last row's sum}
    //   ring  Wcap4 floats per node: label (+) gap of row t at [t mod Wcap4]
    //   rows  NLp child ids per node
    //   root  T2cap + 1 floats: the root's cumulative blank products
    char *slab;
    int64_t pair_stride;
    int64_t cap_nodes;
    int Wcap4, NLp;
    ResultDesc out;
    int64_t pair_begin;
    uint32_t *prof;
    int tie_order;
    int prefetch;  // 1: existing children that pass the threshold have their arena lines touched ahead of the prune
};

// LDS layout (words of the dynamic segment, which is the kernel's only LDS: offsets are immediates of the ds_ instructions):
// fixed-size tables first, then what depends on S, N and the ring capacity.
constexpr int O_KEYS = 0;                                   // 64 (+ 4 zero words of padding) 64-bit sort keys
constexpr int O_F = O_KEYS + 2 * (kSlots + 4);              // F_COUNT x 64 slot fields
constexpr int O_PW = O_F + F_COUNT * kSlots;                // 68: probability words by rank / new ranks / evicted nodes
constexpr int O_BT = O_PW + 68;                             // 64: candidate lane of the m-th new node / evicted slots
constexpr int O_FLIST = O_BT + 64;                          // 64: free slots, ascending
constexpr int O_ZERO = O_FLIST + 64;                        // 4: 0.0f (a "column" whose every row is zero)
constexpr int O_SINK = O_ZERO + 4;                          // 64: where LDS-DMA prefetches land (never read)
constexpr int O_PQL = O_SINK + 64;                          // 2 x 64: the quicksort's list
constexpr int O_PQS = O_PQL + 2 * kSlots;                   // its scratch
constexpr int O_VAR = O_PQS + (int)((sizeof(pdq178::Scratch) + 15) / 16 * 4);
// then: f1 | f1n (S * N each, rounded to 4) | tile (S * N x Wcap4) | rings ((P + 1) x Wcap4)

struct SLds {
    float *f1, *f1n; // S * N each: the current row of read 1 and the next one
    float *tile;     // S * N x Wcap4
    float *rings;    // (P + 1) x Wcap4: the slots' rings and a trash ring idle lanes may scribble on
};

__host__ __device__ inline size_t slds_words(int BC, int N, int S, int WC) {
    const size_t P = (size_t)BC * N;
    size_t w = (size_t)O_VAR + 2 * (((size_t)S * N + 3) & ~(size_t)3);
    w += (size_t)S * N * WC + (P + 1) * WC;
    return w;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int bperm_i(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float bperm_f(int src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// min(a, |b|) in one instruction (a NaN operand is dropped, as v_min_f32 does)
__device__ __forceinline__ float vmin_abs_raw(float a, float b) {
#ifdef FCD_HIPEMU
    const float c = __builtin_fabsf(b);
    return a != a ? c : (c != c ? a : (c < a ? c : a));
#else
    float r;
    asm("v_min_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

__device__ __forceinline__ int4 load_int4_l2(const int4 *p) {
    const int32_t *q = reinterpret_cast<const int32_t *>(p);
    return make_int4(load_i32_l2(q), load_i32_l2(q + 1), load_i32_l2(q + 2), load_i32_l2(q + 3));
}

// LogSpace::add for the window-building loop WITHOUT its two rare exits: where ladd_lockstep() leaves the chain for the
// library routine (exp's Ziv test failed, exp's result subnormal or the argument outside [-86, 0] with a `big` so small
// that it could show, ln_1p's Ziv test failed) this one raises `redo` and carries on with a meaningless value; the
// caller runs the whole pass again on ladd_lockstep() when any lane raised it.  Where `redo` stays clear the value is
// ladd()'s, operation for operation.
__device__ __forceinline__ float ladd_spec(float a, float b, const LogAddCoef &K, bool &redo) {
    const bool ab = a <= b;
    const float big = ab ? b : a, small = ab ? a : b;
    const float x = small - big;
    const bool sc_inf = small == kNegInf;
    const bool full = (!(x < kExpFastMin) | (__builtin_fabsf(big) < 8.0779356694631609e-28f)) & !sc_inf;
    float res = big;
    if (ballot(full) != 0ull) {
        const float xs = full ? x : -1.0f;
        const double ye = exp_fast((double)xs, K);
        const double ed = round_to_f32_as_f64(ye);
        const double yl = log1p_fast(ed, K);
        const float l = (float)yl;
        const int unsafe = (int)!(xs >= kExpFastMin) | (int)round_to_f32_unsafe(ye) |
                           (int)((uint32_t)(bits_of(ed) >> 32) < 0x38100000u) | (int)round_to_f32_unsafe(yl);
        redo = redo | (full & (unsafe != 0));
        res = full ? big + l : res;
    }
    return res;
}

#ifdef FCD_HIPEMU
static long g_emu_passes = 0, g_emu_exact = 0;
#endif

// LogSpace::add without branches or votes, for the bookkeeping that runs under divergent control flow (one row of an
// entry's extension, the merge of a candidate's items): the transcendental part is always evaluated -- on a tame argument
// where it is not needed -- and `bad` is raised where ladd() would have left its fast path; the caller then repeats the
// block on ladd().  Where `bad` stays clear the value is ladd()'s, operation for operation.  A wavefront alone on its SIMD
// pays per instruction: ladd() with its exits and the 64-bit literals it re-materialises is ~2.5 times this.
__device__ __forceinline__ float ladd_bf(float a, float b, const LogAddCoef &K, bool &bad) {
    const bool ab = a <= b;
    const float big = ab ? b : a, small = ab ? a : b;
    const float x = small - big;
    const bool have = !(small == kNegInf);
    const bool full = !(x < kExpFastMin) & have;
    const float xs = full ? x : -1.0f;
    const double ye = exp_fast((double)xs, K);
    const double ed = round_to_f32_as_f64(ye);
    const double yl = log1p_fast(ed, K);
    const uint32_t d1 = ((uint32_t)bits_of(ye) & 0x1FFFFFFFu) + 0xF0000200u;
    const uint32_t d2 = ((uint32_t)bits_of(yl) & 0x1FFFFFFFu) + 0xF0000200u;
    const uint32_t dm = d1 < d2 ? d1 : d2;
    bad = bad | (full & (dm < 1024u)) | (have & (x < kExpFastMin) & (__builtin_fabsf(big) < 8.0779356694631609e-28f));
    return full ? big + (float)yl : big;
}

template <int MODE, bool PROF>
__global__ __launch_bounds__(64, 2) void duplex_slots_kernel(SlotParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x;
    const int64_t local = blockIdx.x;
    const int64_t r = p.pair_begin + local;
    const int N = p.N, NL = N - 1, BC = p.beam_size, S = p.S, SN = S * N, WC = p.Wcap4, NLp = p.NLp;
    const int P = BC * N;
    const bool crf = p.crf != 0;
    const bool collapse = !crf && p.collapse != 0;
    const float thr = p.thr_ln;
    const bool pdq = p.tie_order == FCD_TIE_PDQ178;
    SLds L;
    {
        const int snp = (SN + 3) & ~3;
        L.f1 = reinterpret_cast<float *>(smem + O_VAR);
        L.f1n = L.f1 + snp;
        L.tile = L.f1n + snp;
        L.rings = L.tile + (size_t)SN * WC;
    }
    int *F = smem + O_F;
    uint64_t *l_keys = reinterpret_cast<uint64_t *>(smem + O_KEYS);
    int *l_pw = smem + O_PW, *l_bt = smem + O_BT, *l_flist = smem + O_FLIST, *l_sink = smem + O_SINK;
    float *l_zero = reinterpret_cast<float *>(smem + O_ZERO);
    uint64_t *l_pql = reinterpret_cast<uint64_t *>(smem + O_PQL);
    pdq178::Scratch *l_pqs = reinterpret_cast<pdq178::Scratch *>(smem + O_PQS);
    auto fi = [&](int f, int slot) -> int & { return F[(((f >> 2) * kSlots + slot) << 2) + (f & 3)]; };
    auto ff = [&](int f, int slot) -> float & { return reinterpret_cast<float *>(F)[(((f >> 2) * kSlots + slot) << 2) + (f & 3)]; };
    auto fg = [&](int g, int slot) -> int4 & { return reinterpret_cast<int4 *>(F)[g * kSlots + slot]; };  // a whole group
    auto ring = [&](int slot) { return L.rings + (size_t)slot * WC; };

    int64_t T1 = p.T1cap, T2 = p.T2cap;
    if (p.len1) { int64_t t = p.len1[r]; T1 = t < 0 ? 0 : (t < T1 ? t : T1); }
    if (p.len2) { int64_t t = p.len2[r]; T2 = t < 0 ? 0 : (t < T2 ? t : T2); }
    const float *ln1 = p.ln1 + r * p.T1cap * SN;
    const float *ln2 = p.ln2 + r * p.T2cap * SN;
    const uint64_t *env = p.env + r * p.env_stride * 2;
    char *slab = p.slab + local * p.pair_stride;
    const uint32_t cap32 = (uint32_t)p.cap_nodes;
    const uint32_t o_aux = cap32 * 16u, o_ring = cap32 * 32u, o_rows = o_ring + cap32 * (uint32_t)(WC * 4),
                   o_root = o_rows + cap32 * (uint32_t)(NLp * 4);
    auto g_meta = [&](uint32_t nd) { return reinterpret_cast<int4 *>(slab + (nd << 4)); };
    auto g_aux = [&](uint32_t nd) { return reinterpret_cast<int4 *>(slab + (o_aux + (nd << 4))); };
    auto g_rows = [&](uint32_t nd) { return reinterpret_cast<int32_t *>(slab + (o_rows + nd * (uint32_t)(NLp * 4))); };
    auto g_ring = [&](uint32_t nd) { return reinterpret_cast<float *>(slab + (o_ring + nd * (uint32_t)(WC * 4))); };
    float *rootgap = reinterpret_cast<float *>(slab + o_root);
    uint8_t *lab_out = p.out.labels + r * p.out.out_stride;

    // the log-add's coefficient table: in vector registers for the whole kernel (logsumexp flavour)
    LogAddCoef K = logadd_coef();
    if (MODE == FCD_LOGADD_LOGSUMEXP) {
        FCD_OPAQUE_V(K.log2e); FCD_OPAQUE_V(K.ln2hi); FCD_OPAQUE_V(K.ln2lo); FCD_OPAQUE_V(K.two); FCD_OPAQUE_V(K.magic);
#pragma unroll
        for (int u = 0; u < 12; ++u) FCD_OPAQUE_V(K.e[u]);
#pragma unroll
        for (int u = 0; u < 15; ++u) FCD_OPAQUE_V(K.a[u]);
    }
    // the bookkeeping's log-add: the branch-free form first (logsumexp flavour), LogSpace::add itself on a repeat
    bool bf_bad = false;
    auto la_fast = [&](float a, float b) __attribute__((always_inline)) {
        return MODE == FCD_LOGADD_LOGSUMEXP ? ladd_bf(a, b, K, bf_bad) : ladd<MODE>(a, b);
    };
    auto la_exact = [&](float a, float b) __attribute__((always_inline)) { return ladd<MODE>(a, b); };
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
    int st1 = 0, st2 = 0;
    if (crf) {  // init_state.argmax() (:679,:691; first maximum, NaN panics in the reference)
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
        // root_probs (:389-409) / crf_root_probs (:411-441): cumulative blank product, sequential as the reference's
        float cur = 0.0f;
        rootgap[0] = cur;
        int st = st2;
        for (int t = 0; t < root_end; ++t) {
            cur = cur + ln2[(uint32_t)((t * S + st) * N)];
            rootgap[t + 1] = cur;
            if (crf) st = (int)(((int64_t)st * NL) % S);  // :437
        }
        // slot 0: the root
        fi(F_NODE, 0) = -1;
        fi(F_TIP, 0) = -1;
        fi(F_PAR, 0) = -2;
        fi(F_STATE, 0) = st1;
        fi(F_OFF, 0) = -1;
        fi(F_END, 0) = root_end;
        fi(F_VFROM, 0) = -1;
        ff(F_MX, 0) = 0.0f;
        ff(F_LLAB, 0) = kNegInf;
        ff(F_LSUM, 0) = kNegInf;
        fi(F_XREP, 0) = 0;
        fi(F_DEPTH, 0) = 0;
    }
    for (int j = lane; j < kNLMax + 1; j += kWave) fi(F_CHILD0 + j, 0) = -1;
    if (lane < 4) l_keys[kSlots + lane] = 0ull;  // padding of the four-at-a-time rank loop
    if (lane < 4) l_zero[lane] = 0.0f;
    // (rows of read 2 that were never loaded read as a finite number: "zero (x) p" stays zero in the guard rows' sums)
    for (int x = lane; x < SN * WC; x += kWave) L.tile[x] = 0.0f;
    for (int x = lane; x < WC; x += kWave) ring(P)[x] = kNegInf;
    // a NaN or +inf among the posteriors of read 2 loaded so far: the window builds take their exact form from then on
    bool unclean = false;
    l_flist[lane] = lane + 1;                    // free: every slot but 0
    // lane c is candidate (ci, ck) of every step
    const int ci = lane / N, ck = lane - ci * N;
    // ... word offset (at slot 0) of the child entry candidate `lane` looks at: F_CHILD0 + ck - 1
    const int kid_off = ((((F_CHILD0 + (ck > 0 ? ck - 1 : 0)) >> 2) * kSlots) << 2) + ((F_CHILD0 + (ck > 0 ? ck - 1 : 0)) & 3);
    // ... item `lane` of a batch of rings copied 16 bytes at a time is piece g_c0 of ring g_q0; 64 items on: + (g_dq, g_dc)
    const int g_q0 = lane / (WC >> 2), g_c0 = lane - g_q0 * (WC >> 2);
    const int g_dq = kWave / (WC >> 2), g_dc = kWave - g_dq * (WC >> 2);
    // ... and element (l_rw, l_sn) of a block of read-2 rows taken one element per lane
    const int l_rw = lane / SN, l_sn = lane - l_rw * SN;
    // rank lanes: lane e < B holds entry e of the beam
    int slotE = 0, nodeE = -1;
    float lpE = kNegInf, gpE = 0.0f;  // root: label zero, gap one
    int B = 1, nn = 0, last_hi = 0;
    // read-2 tile: rows [tl_lo, tl_hi) are in the LDS ring
    int tl_lo = 0, tl_hi = 0;
    // the slot of the current lower bound (lo mod Wcap4), kept incrementally
    int cur_lo = 0, cur_lo_s = 0;
    // rows of read 2 asked for one step ahead: element (pf_row, pf_sn) in pf_val for lanes below pf_n
    float pf_val = 0.0f;
    int pf_lo = 0, pf_hi = 0;
    // the row of read 1 the coming step expands with
    if (lane < SN) L.f1[lane] = ln1[lane];
    float f1_next = 0.0f;

    const bool prof = PROF && p.prof != nullptr;
    uint64_t acc[5] = {0, 0, 0, 0, 0}, t_prev = 0, t_now = 0;
    uint32_t n_iter = 0, n_newnodes = 0, n_slow = 0, n_enter = 0, n_ext = 0, n_redo = 0;
    uint64_t sub[5] = {0, 0, 0, 0, 0}, t_sub = 0, t_sub2 = 0;
    int stamp_dep = 0;
#define FCD_S_SUB_BEGIN() if (PROF && prof) FCD_STAMP(t_sub, stamp_dep);
#define FCD_S_SUB(k) if (PROF && prof) { FCD_STAMP(t_sub2, stamp_dep); sub[k] += t_sub2 - t_sub; t_sub = t_sub2; }
#define FCD_S_PHASE(k) if (PROF && prof) { FCD_STAMP(t_now, stamp_dep); acc[k] += t_now - t_prev; t_prev = t_now; }
    // developer build (-DFCD_SLOTS_FINE, tools/dev/duplex_fine.py): the step cut into 16 consecutive intervals, written
    // behind the regular account (32 words per pair)
#ifdef FCD_SLOTS_FINE
    uint64_t fine[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_fine = 0, t_fine2 = 0;
    if (PROF && prof) FCD_STAMP(t_fine, stamp_dep);
#define FCD_S_FINE(k) if (PROF && prof) { FCD_STAMP(t_fine2, stamp_dep); fine[k] += t_fine2 - t_fine; t_fine = t_fine2; }
    constexpr int kProfWords = 32;
#else
#define FCD_S_FINE(k)
    constexpr int kProfWords = 16;
#endif
    if (PROF && prof) FCD_STAMP(t_prev, stamp_dep);
    wave_sync();

    uint64_t env_lo_next = env[0], env_hi_next = env[1];
    for (int64_t t1 = 0; t1 < T1; ++t1) {
        // ---- envelope (:485-488) ----
        const uint64_t lo_u = env_lo_next, hi_u = env_hi_next;
        const bool more = t1 + 1 < T1;
        if (more) {
            env_lo_next = env[2 * (t1 + 1)];
            env_hi_next = env[2 * (t1 + 1) + 1];
            if (lane < SN) f1_next = ln1[(t1 + 1) * SN + lane];
        }
        const int hi = (int)(hi_u > (uint64_t)T2 ? (uint64_t)T2 : hi_u);
        if (lo_u >= (uint64_t)hi || lo_u > (uint64_t)last_hi) return fail(FCD_ST_INVALID_ENVELOPE);
        const int lo = (int)lo_u;
        const int W = hi - lo;
        {
            const int d = lo - cur_lo;
            int s = cur_lo_s + d;
            if (d < -WC || d > WC) s = lo % WC;
            else {
                s = s < 0 ? s + WC : s;
                s = s >= WC ? s - WC : s;
            }
            cur_lo = lo;
            cur_lo_s = s;
        }
        const int lo_s = cur_lo_s;
        // slot of row t, |t - lo| < Wcap4
        auto slotn = [&](int t) {
            int s = lo_s + (t - lo);
            s = s < 0 ? s + WC : s;
            return s >= WC ? s - WC : s;
        };

        // ---- read-2 tile: rows [max(lo - 1, 0), hi) ----
        FCD_S_SUB_BEGIN()
        {
            const int need_lo = lo > 0 ? lo - 1 : 0;
            int from;
            if (tl_hi > tl_lo && need_lo >= tl_lo && need_lo <= tl_hi) {
                from = tl_hi;  // rows the envelope gained (none when hi <= tl_hi)
            } else {
                from = need_lo;
                tl_lo = need_lo;
                tl_hi = need_lo;
            }
            if (hi > from) {
                if (from == pf_lo && hi == pf_hi) {  // asked for during the previous step: one element per lane
                    const int n = (hi - from) * SN;
                    if (lane < n) L.tile[(size_t)l_sn * WC + slotn(from + l_rw)] = pf_val;
                    unclean = unclean || ballot(lane < n && !(pf_val < __builtin_huge_valf())) != 0ull;
                } else {
                    const int n = (hi - from) * SN;
                    bool bad_v = false;
                    for (int x = lane; x < n; x += kWave) {
                        const int rw = x / SN, sn = x - rw * SN;
                        const int row = from + rw;
                        const float v = ln2[(uint32_t)(row * SN + sn)];
                        L.tile[(size_t)sn * WC + (row % WC)] = v;
                        bad_v = bad_v || !(v < __builtin_huge_valf());
                    }
                    unclean = unclean || ballot(bad_v) != 0ull;
                }
                tl_hi = hi;
                if (tl_hi - tl_lo > WC) tl_lo = tl_hi - WC;
            }
            // ask for the rows the NEXT step will gain
            pf_lo = pf_hi = 0;
            if (more) {
                const int hi_n = (int)(env_hi_next > (uint64_t)T2 ? (uint64_t)T2 : env_hi_next);
                const int n = (hi_n - tl_hi) * SN;
                if (hi_n > tl_hi && n <= kWave) {
                    pf_lo = tl_hi;
                    pf_hi = hi_n;
                    if (lane < n) pf_val = ln2[(uint32_t)(tl_hi * SN + lane)];  // (row-major: element `lane` of the block of rows)
                }
            }
        }
        FCD_S_SUB(0)
        FCD_S_FINE(0)  // envelope + tile

        // the root (in the beam for the first few rows only) has no ring of its own: the rows its children's extensions
        // and its children's builds will ask for, [lo - 1, hi - 1), are staged into slot 0's ring from the cumulative
        // blank products -- before the extension, which then reads a root parent like any other parent in the beam
        const bool root_in = ballot(lane < B && nodeE < 0) != 0ull;
        if (root_in) {
            float *rg = ring(0);
            for (int j = lane; j < W; j += kWave) {
                const int at = lo - 1 + j;
                if (at < -1) continue;
                rg[slotn(at)] = at < root_end ? load_f32_l2(rootgap + (at + 1)) : kNegInf;
            }
            wave_sync();
        }
        const bool grew = hi > last_hi;
        if (grew) {
            FCD_S_SUB_BEGIN()
            // ---- :493 beam.sort_by_key(node): parents before children ----
            {
                int rk = 0;
                for (int j = 0; j < B; ++j) {
                    const int nj = rl_i(nodeE, j);
                    rk += nj < nodeE ? 1 : 0;
                }
                const int dst = lane < B ? rk : 63;  // (B <= 32: lane 63 is never a rank lane)
                const int s2 = __builtin_amdgcn_ds_permute(dst << 2, slotE);
                const int n2 = __builtin_amdgcn_ds_permute(dst << 2, nodeE);
                const int l2 = __builtin_amdgcn_ds_permute(dst << 2, __float_as_int(lpE));
                const int g2 = __builtin_amdgcn_ds_permute(dst << 2, __float_as_int(gpE));
                slotE = s2; nodeE = n2; lpE = __int_as_float(l2); gpE = __int_as_float(g2);
            }
        }
        // which rank holds my parent (rank lanes) -- also used by the expansion below
        const bool mineE = lane < B;
        int parE = -2, pslotE = -1, prankE = -1;
        if (mineE) parE = fi(F_PAR, slotE);
        for (int j = 0; j < B; ++j) {
            const int nj = rl_i(nodeE, j), sj = rl_i(slotE, j);
            const bool hit = mineE && nodeE >= 0 && parE == nj;
            pslotE = hit ? sj : pslotE;
            prankE = hit ? j : prankE;
        }
        if (grew) {
            // ---- extend_secondary_probs (:338-387) for every beam node, one per lane ----
            // An entry appends rows [end, hi).  Row idx reads the parent's row idx - 1 <= hi - 2, which exists BEFORE this
            // step whenever the parent's window reaches hi - 1 -- then nothing depends on a row written in this step and
            // the reference's parents-first order (:493) is immaterial.  Only a parent that is itself behind takes the
            // one-entry-at-a-time path (entries are in node order: parents first).
            const bool mine = mineE && nodeE >= 0;
            int off = 0, end = 0, vfrom = 0, lab = 0, tst = 0, p_off = 0, p_end = 0, p_vfrom = 0, xrep = 0, e_depth = 0;
            float mx = kNegInf, llab = kNegInf, lsum = kNegInf;
            bool rescan = false, panic = false, behind = false;
            if (mine) {
                const int4 g_id = fg(G_ID, slotE), g_win = fg(G_WIN, slotE), g_val = fg(G_VAL, slotE);
                off = g_win.x; end = g_win.y; vfrom = g_win.z; xrep = g_win.w;
                mx = __int_as_float(g_val.x); llab = __int_as_float(g_val.y); e_depth = g_val.z; lsum = __int_as_float(g_val.w);
                lab = g_id.y; tst = g_id.w;
                if (lo > off) {  // :351-359
                    const int keep = lo - 1;
                    if (keep > off) {
                        if (keep < end) off = keep;
                        else { off = keep; end = keep; }
                    }
                    if (end == off) {  // emptied: a new run of rows starts at lo, below it two guard rows of "zero"
                        off = lo; end = lo; vfrom = lo; llab = kNegInf; lsum = kNegInf;
                        float *mw0 = ring(slotE);
                        mw0[slotn(lo - 1)] = kNegInf;
                        mw0[slotn(lo - 2)] = kNegInf;
                    }
                    rescan = true;  // update_max(lo, hi)
                }
                panic = end >= hi;  // assert!(current_end < upper_bound) :363-366
                if (parE >= 0 || pslotE >= 0) {
                    if (pslotE >= 0) {
                        const int4 pw_ = fg(G_WIN, pslotE);
                        p_off = pw_.x; p_end = pw_.y; p_vfrom = pw_.z;
                        behind = p_end < hi - 1;
                    } else {
                        const int4 pp_ = fg(G_PAR, slotE);
                        p_off = pp_.x; p_end = pp_.y; p_vfrom = pp_.z;
                    }
                } else {  // the root's window
                    p_off = -1; p_end = root_end; p_vfrom = -1;
                }
            }
            FCD_S_SUB(4)
            FCD_S_FINE(1)  // sort, parents, extension prologue
            if (ballot(panic) != 0ull) return fail(FCD_ST_BAD_STATE);
            // update_max over the rows that stay, [max(lo, off), end), for every entry that discarded rows: eight lanes per
            // entry, 16-byte reads, all entries at once (NaN rows never replace the maximum).  One LDS round trip where a
            // loop over the entries (64 lanes each, a wait per trip) was 3.9 k cycles per step.
            if (ballot(rescan) != 0ull) {
                const int r8 = lane & 7, G = WC >> 2;
                for (int e0 = 0; e0 < B; e0 += 8) {
                    const int e = e0 + (lane >> 3);
                    const bool on = e < B;
                    const int ec = on ? e : 0;
                    const int rs = bperm_i(ec, rescan ? 1 : 0), sl = bperm_i(ec, slotE), o2 = bperm_i(ec, off), n2 = bperm_i(ec, end);
                    const int a0 = lo > o2 ? lo : o2;
                    float part = kNegInf;
                    if (on && rs && a0 < n2) {
                        const float4 *rg4 = reinterpret_cast<const float4 *>(ring(sl));
                        const int gb = (n2 - 1) >> 2;
                        const unsigned span = (unsigned)(n2 - a0);
                        for (int g = (a0 >> 2) + r8; g <= gb; g += 40) {  // five 16-byte reads in flight per lane: one trip up to 160 rows
                            float4 v[5];
#pragma unroll
                            for (int u = 0; u < 5; ++u) {
                                const int gu = g + 8 * u;
                                int sg = (lo_s >> 2) + (gu - (lo >> 2));
                                sg = sg < 0 ? sg + G : sg;
                                sg = sg >= G ? sg - G : sg;
                                v[u] = rg4[gu <= gb ? sg : 0];
                            }
#pragma unroll
                            for (int u = 0; u < 5; ++u) {
                                const int d0 = ((g + 8 * u) << 2) - a0;  // (rows before a0 wrap round to huge, rows past n2 fail too)
                                part = vmax_raw(part, (unsigned)d0 < span ? v[u].x : kNegInf);
                                part = vmax_raw(part, (unsigned)(d0 + 1) < span ? v[u].y : kNegInf);
                                part = vmax_raw(part, (unsigned)(d0 + 2) < span ? v[u].z : kNegInf);
                                part = vmax_raw(part, (unsigned)(d0 + 3) < span ? v[u].w : kNegInf);
                            }
                        }
                    }
                    // the eight partial maxima of an entry: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
                    part = lmax(part, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0xB1, 0xf, 0xf, true)));
                    part = lmax(part, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0x4E, 0xf, 0xf, true)));
                    part = lmax(part, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0x141, 0xf, 0xf, true)));
                    const float got = bperm_f(((lane - e0) & 7) << 3, part);
                    if (rescan && lane >= e0 && lane < e0 + 8) mx = got;
                }
            }
            FCD_S_SUB(2)
            FCD_S_FINE(2)  // rescans
            const bool seq = ballot(behind) != 0ull;
            // the recurrence (:361-386) for one entry on its own lane
            const float mx_in = mx;
            auto extend = [&](auto LA) __attribute__((always_inline)) {
                mx = mx_in;
                float *mw = ring(slotE);
                const float *prg = pslotE >= 0 ? ring(pslotE) : nullptr;
                const float *parena = g_ring((uint32_t)(parE < 0 ? 0 : parE));
                const bool rootpar = parE < 0 && pslotE < 0;  // (the root has left the beam: its products are in the arena only)
                float l_lab = llab, l_sum = lsum;  // (the last row's label and sum are fields: no look into the ring)
                const float *tb = L.tile + (size_t)(tst * N) * WC;        // blank of the entry's state (:725-728)
                const float *tl = L.tile + (size_t)(tst * N + lab + 1) * WC;
                for (int idx = end; idx < hi; ++idx) {
                    const int sl = slotn(idx);
                    const int at = idx - 1;
                    // the parent's row idx - 1: label (+) gap, or -- repeated label -- the gap alone, recomputed from
                    // its stored sum of row idx - 2 and the blank of row idx - 1
                    float x = kNegInf;
                    if (at >= p_off && at < p_end) {
                        if (rootpar) {
                            x = load_f32_l2(rootgap + (at + 1));
                        } else if (!xrep) {
                            x = prg ? prg[slotn(at)] : load_f32_l2(parena + (at % WC));
                        } else {
                            float ps = kNegInf;
                            if (at != p_vfrom) ps = prg ? prg[slotn(at - 1)] : load_f32_l2(parena + ((at - 1) % WC));
                            // (xrep only without transition states: the blank column of row `at` is state 0's)
                            const float bl = (at >= tl_lo && at < tl_hi) ? L.tile[slotn(at)] : ln2[(uint32_t)(at * SN)];
                            x = ps + bl;
                        }
                    }
                    const float g = l_sum + tb[sl];
                    const float lb = tl[sl] + LA(l_lab, x);
                    const float sm = LA(lb, g);
                    mw[sl] = sm;
                    mx = lmax(mx, sm);
                    l_lab = lb;
                    l_sum = sm;
                }
                fg(G_WIN, slotE) = make_int4(off, hi, vfrom, xrep);
                fg(G_VAL, slotE) = make_int4(__float_as_int(mx), __float_as_int(l_lab), e_depth, __float_as_int(l_sum));
            };
            if (!seq) {
                bf_bad = false;
                if (mine) extend(la_fast);
                if (MODE == FCD_LOGADD_LOGSUMEXP && ballot(bf_bad && mine) != 0ull) {  // (1e-6 of the log-adds: once more, on LogSpace::add)
                    if (mine) extend(la_exact);
                }
            } else {
                for (int e = 0; e < B; ++e) {
                    if (mine && lane == e) {
                        // the parent may have moved in an earlier trip
                        if (pslotE >= 0) {
                            const int4 pw_ = fg(G_WIN, pslotE);
                            p_off = pw_.x; p_end = pw_.y; p_vfrom = pw_.z;
                        }
                        extend(la_exact);
                    }
                    wave_sync();
                }
            }
            if (PROF && prof) {
                ++n_ext;
                n_slow += seq ? 1u : 0u;
            }
        }
        last_hi = hi;
        wave_sync();
        FCD_S_PHASE(0)
        FCD_S_FINE(3)  // extension rows

        FCD_S_PHASE(1)
        FCD_S_FINE(4)  // root staging

        // ---- expansion (:526-593): lane c = candidate (ci, ck) ----
        const bool act = ci < B;
        const int slot_i = bperm_i(ci, slotE);
        const int node = bperm_i(ci, nodeE);
        const float lp = bperm_f(ci, lpE), gp = bperm_f(ci, gpE);
        const int prank_i = bperm_i(ci, prankE);
        int tip = -1, state = 0, ch = -1, depth = 0;
        if (act) {
            const int4 g_id = fg(G_ID, slot_i);
            tip = g_id.y;
            state = g_id.w;
            depth = fi(F_DEPTH, slot_i);
            if (ck > 0) ch = F[kid_off + (slot_i << 2)];
        }
        // is the child a beam entry already?  (then the extension is folded into that entry's own candidate)
        bool ch_inbeam = false;
        for (int j = 0; j < B; ++j) {
            const int nj = rl_i(nodeE, j);
            ch_inbeam = ch_inbeam | (ch >= 0 && ch == nj);
        }
        // the entry of my parent, if it is in the beam (own candidates)
        const int pj = (act && ck == 0 && node >= 0) ? prank_i : -1;
        const int pjc = pj >= 0 ? pj : 0;
        const int slot_j = bperm_i(pjc, slotE);
        const float gpj = bperm_f(pjc, gpE);
        // running maximum of an existing child that is outside the beam: in its arena record (asked for here, used
        // after the window builds)
        float p2_stale = 0.0f;
        const bool stale_c = act && ck > 0 && ch >= 0 && !ch_inbeam;
        if (stale_c) p2_stale = __int_as_float(load_i32_l2(reinterpret_cast<const int32_t *>(g_aux((uint32_t)ch))));
        const float *row1 = L.f1 + state * N;  // crf: probs[state, :] (:749)
        bool valid = false, is_new = false, rep = false;
        float clp = kNegInf, cgp = kNegInf, p2 = 0.0f;
        int cid = -2;
        // Everything about a candidate but its probabilities: which items it has, which node it is.
        bool blank = false, stay = false, inc = false, inc_first = false, rj = false;
        float pr0 = kNegInf, pt = kNegInf, pl = kNegInf, pk = kNegInf;
        if (act) {
            if (ck == 0) {
                // The node's own candidate is the MERGE (:596-611) of up to three items that share the node -- the blank
                // extension {zero, gap}, the repeat-stay {label, zero} and the extension that arrives from the parent's
                // entry {label, zero} -- folded with LogSpace::add in the order the reference appends them: tips in
                // beam order, and within a tip blank first, labels after (max mode's add keeps a NaN only as its
                // FIRST operand).
                pr0 = row1[0];
                blank = pr0 > thr;  // :529
                stay = collapse && tip >= 0;
                if (stay) {
                    pt = row1[tip + 1];
                    stay = !(pt < thr);  // :541-544
                }
                if (pj >= 0) {
                    pl = L.f1[fi(F_STATE, slot_j) * N + tip + 1];  // the PARENT's row
                    if (!(pl < thr)) {
                        rj = collapse && fi(F_TIP, slot_j) == tip;
                        // (the parent's own test `gap > zero` (:546) only guards the CREATION of this node: it exists)
                        inc = true;
                        inc_first = pj < ci;  // the parent's entry comes earlier in the beam: its item was appended first
                    }
                }
                valid = blank || stay || inc;
                cid = node;
                if (node >= 0) p2 = ff(F_MX, slot_i);  // :613-618 (the root keeps one)
            } else {
                const int l = ck - 1;
                pk = row1[ck];
                const bool pass = !(pk < thr);  // :537
                rep = collapse && l == tip;
                const bool exists = ch >= 0;
                valid = pass && (exists || !rep || gp > kNegInf) && !ch_inbeam;  // :546
                is_new = valid && !exists;
                cid = ch;
                if (stale_c) p2 = p2_stale;
            }
        }
        // ... and the probabilities.  An entry's label (+) gap is wanted by all of its candidates (the blank extension,
        // every label's extension) and by the child entry it extends into: ONE log-add per lane, handed to the child's
        // lane by ds_bpermute.  logsumexp flavour: LogSpace::add with a zero operand returns the other operand whatever
        // it is (duplex.rs:45-47), so the merge is the labels' items folded in the reference's order and the gap item
        // alone -- three log-adds per step where the literal fold took nine.  (Max flavour: not commutative once a NaN
        // takes part, folded literally; it has no transcendental to save.)
        auto probabilities = [&](auto LA) __attribute__((always_inline)) {
            const float lg = LA(lp, gp);                  // tip.prob_1.probability()
            const float lgj = bperm_f(pjc * N, lg);       // ... of the parent's entry (own candidates with an incoming item)
            if (ck == 0) {
                const float g_item = blank ? lg + pr0 : kNegInf;
                const float s_item = stay ? lp + pt : kNegInf;
                const float c_item = inc ? (rj ? gpj + pl : lgj + pl) : kNegInf;
                if (MODE == FCD_LOGADD_LOGSUMEXP) {
                    const bool both = inc && stay;
                    const float first = inc_first ? c_item : s_item, second = inc_first ? s_item : c_item;
                    const float m2 = LA(both ? first : kNegInf, both ? second : kNegInf);
                    clp = both ? m2 : (inc ? c_item : s_item);
                    cgp = g_item;
                } else {
                    bool have = false;
                    auto push = [&](float l_it, float g_it) {
                        if (!have) {
                            clp = l_it;
                            cgp = g_it;
                            have = true;
                        } else {
                            clp = LA(clp, l_it);
                            cgp = LA(cgp, g_it);
                        }
                    };
                    clp = kNegInf;
                    cgp = kNegInf;
                    if (inc && inc_first) push(c_item, kNegInf);
                    if (blank) push(kNegInf, g_item);
                    if (stay) push(s_item, kNegInf);
                    if (inc && !inc_first) push(c_item, kNegInf);
                }
            } else {
                clp = rep ? gp + pk : lg + pk;
                cgp = kNegInf;
            }
        };
        FCD_S_FINE(5)  // expansion: entry fields, candidates
        bf_bad = false;
        probabilities(la_fast);
        if (MODE == FCD_LOGADD_LOGSUMEXP && ballot(bf_bad && act) != 0ull) probabilities(la_exact);
        FCD_S_FINE(6)  // expansion: probabilities
#ifndef FCD_HIPEMU
        // An existing child outside the beam that passed the threshold may be in the next beam, and then its ring and
        // record come back from the arena -- evicted tens of steps ago, i.e. from HBM (2-4 k cycles on the critical path
        // of the hand-over when asked for after the prune).  Touch their cache lines now, with LDS-DMA loads into a
        // scratch row nobody reads (no register is held, nothing waits): by the prune they sit in L2.
        if (p.prefetch && valid && ck > 0 && !is_new) {
            typedef __attribute__((address_space(1))) const void gptr_t;
            typedef __attribute__((address_space(3))) void lptr_t;
            lptr_t *sink = (lptr_t *)l_sink;
            const char *rb = reinterpret_cast<const char *>(g_ring((uint32_t)ch));
            const int nb = WC * 4;
            for (int b = 0; b < nb; b += 64) __builtin_amdgcn_global_load_lds((gptr_t *)(rb + b), sink, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)(rb + nb - 4), sink, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)g_meta((uint32_t)ch), sink, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)g_rows((uint32_t)ch), sink, 4, 0, 0);
        }
#endif
        const uint64_t m_new = ballot(is_new);
        const int n_new = popc64(m_new);
        const int pre = popc64(m_new & lanemask_lt());
        int nbuf = 0;
        if (is_new) {
            cid = nn + pre;
            l_bt[pre] = lane;
            nbuf = l_flist[pre];
        }
        const bool can = is_new && cid < p.cap_nodes;
        if (can) {  // add_node (tree.rs:125-145): the new node's slot
            const int l = ck - 1;
            fg(G_ID, nbuf) = make_int4(cid, l, node, crf ? (int)(((int64_t)state * NL) % S) + l : 0);  // :782
            fg(G_WIN, nbuf) = make_int4(lo, hi, lo, (!crf && node >= 0 && tip == l) ? 1 : 0);  // :512 (no collapse_repeats test there)
            fi(F_DEPTH, nbuf) = depth + 1;  // (running maximum and last label: by the lane that builds the window)
            fg(G_KID, nbuf) = make_int4(-1, -1, -1, -1);
            if (NL > 4) fg(G_KID + 1, nbuf) = make_int4(-1, -1, -1, -1);
            F[kid_off + (slot_i << 2)] = cid;
        }
        FCD_S_PHASE(2)
        FCD_S_FINE(7)  // new-node slots
        if (PROF && prof) {
            n_newnodes += (uint32_t)n_new;
            n_iter += (uint32_t)(W + 1) * (uint32_t)((n_new + 31) / 32);
        }
        wave_sync();

        // ---- new nodes: build_secondary_probs (:212-249), whole window [lo, hi) ----
        // One lane per node, both chains, every test per row: the exact form (glibc 2.35 flavour always; the other
        // flavours when a fast pass asked for it).  W + 1 trips.
        auto build_exact = [&](int m0) {
            const int m = m0 + lane;
            const bool have = m < n_new;
            const int owner = have ? l_bt[m] : 0;
            const int o_flags = bperm_i(owner, (can ? 1 : 0) | (rep ? 2 : 0));
            const bool work = have && (o_flags & 1);
            const bool q_rep = (o_flags & 2) != 0;
            const int q_buf = bperm_i(owner, nbuf), q_ps = bperm_i(owner, slot_i), q_l = bperm_i(owner, ck - 1);
            const int q_state = bperm_i(owner, state), q_node = bperm_i(owner, node);
            float lb = kNegInf, sm = kNegInf, mx = kNegInf;
            if (work) {
                float *my = ring(q_buf);
                const float *prg = ring(q_ps);
                const int4 pw_ = fg(G_WIN, q_ps);
                const int p_off = q_node < 0 ? -1 : pw_.x, p_end = q_node < 0 ? root_end : pw_.y;
                const int p_vfrom = q_node < 0 ? -1 : pw_.z;
                const float *tb = L.tile + (size_t)(q_state * N) * WC;  // crf: tip.state (:772)
                const float *tl = L.tile + (size_t)(q_state * N + q_l + 1) * WC;
                for (int idx = lo; idx < hi; ++idx) {
                    const int sl = slotn(idx), at = idx - 1;
                    float x = kNegInf;
                    if (at >= p_off && at < p_end) {
                        if (!q_rep) x = prg[slotn(at)];
                        else {
                            float ps = kNegInf;
                            if (at != p_vfrom) ps = prg[slotn(at - 1)];
                            const float bl = (at >= tl_lo && at < tl_hi) ? L.tile[slotn(at)] : ln2[(uint32_t)(at * SN)];
                            x = ps + bl;
                        }
                    }
                    const float g = sm + tb[sl];
                    lb = tl[sl] + ladd<MODE>(lb, x);
                    sm = ladd<MODE>(lb, g);
                    my[sl] = sm;
                    mx = lmax(mx, sm);
                }
                ff(F_MX, q_buf) = mx;
                ff(F_LLAB, q_buf) = lb;
                ff(F_LSUM, q_buf) = sm;
                my[slotn(lo - 1)] = kNegInf;  // the guard rows
                my[slotn(lo - 2)] = kNegInf;
            }
        };
        if (n_new > 0) {
            // Who builds what: the m-th new node's parameters come from its candidate lane (l_bt).  Shared by the passes below.
            struct NodeParams {
                bool work, rep;
                int buf, ps, l, state;
                int p_off, p_end, p_vfrom;
            };
            auto node_params = [&](int m) {
                NodeParams q;
                const bool have = m < n_new;
                const int owner = have ? l_bt[m] : 0;
                const int o_flags = bperm_i(owner, (can ? 1 : 0) | (rep ? 2 : 0));
                q.work = have && (o_flags & 1);
                q.rep = q.work && (o_flags & 2) != 0;
                const int o_buf = bperm_i(owner, nbuf), o_ps = bperm_i(owner, slot_i), o_l = bperm_i(owner, ck - 1);
                const int o_state = bperm_i(owner, state), o_node = bperm_i(owner, node);
                // (idle lanes run the loops too, on the trash ring / slot 0's tables: every address they form is a real one)
                q.buf = q.work ? o_buf : P;
                q.ps = q.work ? o_ps : P;
                q.l = q.work ? o_l : 0;
                q.state = q.work ? o_state : 0;
                const bool root = !q.work || o_node < 0;
                const int4 pw_ = fg(G_WIN, q.ps < kSlots ? q.ps : 0);
                q.p_off = root ? -1 : pw_.x;
                q.p_end = root ? root_end : pw_.y;
                q.p_vfrom = root ? -1 : pw_.z;
                return q;
            };
            // The lean passes read a parent's row t - 1 (t - 2 for a repeated label) for every t in [lo, hi) straight
            // out of its ring, without asking whether the row is inside the parent's window: true when the window reaches
            // hi - 1 and starts at or before lo - 1 -- or starts where the parent's run of rows began (vfrom), below which
            // every ring holds two rows of "zero" (the guard rows; root: staged).  Receding envelopes, and special
            // posteriors whose "zero (x) p" is not zero, take the general form.
            auto lean_ok = [&](const NodeParams &q) {
                return !q.work || (q.p_end >= hi - 1 && (q.p_off <= lo - 1 || q.p_off == q.p_vfrom) && q.p_vfrom <= lo);
            };
            bool redo_pass = MODE == FCD_LOGADD_LOGSUMEXP_GLIBC235 || unclean;
            if (MODE == FCD_LOGADD_LOGSUMEXP && !redo_pass) {
                // The recurrence of one node is two interleaved serial chains:
                //   label_t = p_t[label] (x) (label_{t-1} (+) X_{t-1})            (needs only the label chain)
                //   sum_t   = label_t (+) (sum_{t-1} (x) p_t[blank])   [gap_t = sum_{t-1} (x) p_t[blank]]
                // so every new node gets a PAIR of lanes: the even lane runs the label chain, the odd lane runs the sum
                // chain one row behind it -- same operations in the same order as the reference, one log-add per trip.
                // A lone wavefront pays ~6 cycles per instruction whatever the instruction is, so the trip is kept to
                // the log-add and a dozen instructions: every operand comes through a pointer that advances by one row
                // (the three trips in which some pointer wraps round its ring are fix-up points BETWEEN runs of trips,
                // not tests inside them), the rare exits of LogSpace::add are folded into two accumulators (the smallest
                // distance of a binary64 result from an f32 rounding boundary; the largest `big`), and nothing is masked:
                // idle trips and idle lanes compute on "zero" or on the trash ring.
                const bool isA = (lane & 1) == 0;
                // slots of rows lo - 2 .. lo + 1 and the first trip in which a pointer wraps (wave-uniform)
                const int s_m2 = slotn(lo - 2), s_m1 = slotn(lo - 1), s_p1 = slotn(lo + 1);
                const int first_wrap = WC - (s_m1 > lo_s ? s_m1 : (s_p1 > lo_s ? s_p1 : lo_s));  // (the largest of three consecutive slots mod WC wraps first)
                const int n_trips = W + 1;
                for (int rd = 0; rd * 32 < n_new && !redo_pass; ++rd) {
                    const NodeParams q = node_params(rd * 32 + (lane >> 1));
                    if (ballot(!lean_ok(q)) != 0ull) { redo_pass = true; break; }
                    float *trash = ring(P);
                    const float *col = L.tile + (size_t)(q.state * N + (isA ? q.l + 1 : 0)) * WC;  // even: the label's column, odd: the blank's
                    const float *prg = ring(q.ps);
                    // pa: this lane's coefficient of the NEXT trip's row; pb (+ pz): X of the next trip's row -- the
                    // parent's sum one row up, or (repeated label) its sum two rows up plus the blank one row up; pw: where
                    // the odd lane stores.  Even lane: rows lo, lo + 1, ...; odd lane: one row behind.
                    const float *pa = col + (isA ? s_p1 : lo_s);
                    const float *pb = isA ? prg + (q.rep ? s_m1 : lo_s) : trash + lo_s;
                    const float *pz = q.rep && isA ? L.tile + lo_s : l_zero;
                    const int zstep = q.rep && isA ? 1 : 0;
                    float *pw = (isA ? trash : ring(q.buf)) + (isA ? lo_s : s_m1);
                    int wa = WC - (isA ? s_p1 : lo_s), wb = WC - (isA && q.rep ? s_m1 : lo_s), wz = zstep ? WC - lo_s : 0x7FFFFFFF,
                        ww = WC - (isA ? lo_s : s_m1);
                    float c_cur = isA ? col[lo_s] : 0.0f;
                    float x_cur = kNegInf;
                    if (isA) x_cur = q.rep ? prg[s_m2] + L.tile[s_m1] : prg[s_m1] + 0.0f;
                    float lb = kNegInf;     // A: label_{t-1};  B: label_{t'} received from A
                    float sm = kNegInf;     // B: sum_{t'-1}
                    float mx = kNegInf, lkeep = kNegInf;
                    uint32_t zacc = 0xFFFFFFFFu;  // min over the trips of (dropped bits - (2^28 - 512)) mod 2^32: < 1024 = a Ziv test failed
                    float bmin = __builtin_huge_valf();  // min over the trips of |big|: below 2^-90 = exp(x) under -86 could show in big + ln_1p(exp(x))
                    // (two copies of the trips: passes without a repeated-label child -- most -- read X with one load)
                    auto trips = [&](auto with_rep) __attribute__((always_inline)) {
                    int i = 0;
                    for (int seg = 0; seg < 4; ++seg) {
                        const int stop_raw = seg < 3 ? first_wrap + seg : n_trips;
                        const int stop = stop_raw < n_trips ? stop_raw : n_trips;
                        // fix-ups due before trip i (no pointer wraps before trip 1)
                        pa -= i == wa ? WC : 0;
                        pb -= i == wb ? WC : 0;
                        if (with_rep.value) pz -= i == wz ? WC : 0;
                        pw -= i == ww ? WC : 0;
                        for (; i < stop; ++i) {
                            const float c_nxt = *pa;
                            const float x_nxt = with_rep.value ? *pb + *pz : *pb + 0.0f;
                            const float b = isA ? x_cur : sm + c_cur;  // A: X_{t-1};  B: gap_{t'}
                            // ---- LogSpace::add(lb, b), exits folded ----
                            const bool ab = lb <= b;
                            const float big = ab ? b : lb, small = ab ? lb : b;
                            const float x = small - big;
                            const bool full = !(x < kExpFastMin) & !(small == kNegInf);
                            float v = big;
                            if (ballot(full) != 0ull) {
                                const float xs = full ? x : -1.0f;
                                const double ye = exp_fast((double)xs, K);
                                const double ed = round_to_f32_as_f64(ye);
                                const double yl = log1p_fast(ed, K);
                                const uint32_t d1 = ((uint32_t)bits_of(ye) & 0x1FFFFFFFu) + 0xF0000200u;
                                const uint32_t d2 = ((uint32_t)bits_of(yl) & 0x1FFFFFFFu) + 0xF0000200u;
                                zacc = zacc < d1 ? zacc : d1;
                                zacc = zacc < d2 ? zacc : d2;
                                const float r = big + (float)yl;
                                v = full ? r : big;
                            }
                            bmin = vmin_abs_raw(bmin, big);
                            *pw = v;  // (odd lane: sum_{t'}; even lane: the trash ring)
                            sm = v;
                            mx = vmax_raw(mx, v);  // (LogSpace::max keeps the accumulator against a NaN, as v_max_f32 does)
                            lkeep = lb;
                            const int lb_out = __float_as_int(c_cur + v);  // label_t (even lane)
                            // hand label_t to the odd lane for the next trip; the even lane keeps it
                            lb = __int_as_float(__builtin_amdgcn_update_dpp(lb_out, lb_out, 0xA0 /* quad_perm [0,0,2,2] */, 0xf, 0xf, false));
                            c_cur = c_nxt;
                            x_cur = x_nxt;
                            ++pa; ++pb; ++pw;
                            if (with_rep.value) pz += zstep;
                        }
                    }
                    };
                    if (ballot(q.rep) != 0ull) trips(std::true_type{});
                    else trips(std::false_type{});
                    if (q.work && !isA) {
                        float *my = ring(q.buf);
                        ff(F_MX, q.buf) = mx;
                        ff(F_LLAB, q.buf) = lkeep;  // label_{hi-1}: what the odd lane combined in its last trip
                        ff(F_LSUM, q.buf) = sm;      // sum_{hi-1}
                        my[s_m1] = kNegInf;         // the guard rows
                        my[s_m2] = kNegInf;
                    }
                    if (ballot(q.work && (zacc < 1024u || bmin < 8.0779356694631609e-28f)) != 0ull) redo_pass = true;
                }
                if (PROF && prof) n_redo += redo_pass ? 1u : 0u;
            } else if (MODE == FCD_LOGADD_MAX && !redo_pass) {
                // Max-product mode has no transcendental in the recurrence, so a row costs its instructions: one lane per
                // node, four rows per trip -- 16-byte LDS reads of the coefficients and of the parent's rows, one 16-byte
                // store.  LogSpace::add's max flavour is v_max_f32 unless its FIRST operand is a NaN (the label chain,
                // duplex.rs:57-61), which takes a NaN or a +inf among the posteriors of read 2 seen so far: such pairs
                // (`unclean`) build on the exact form from then on.
                const NodeParams q = node_params(lane);
                if (ballot(!lean_ok(q)) != 0ull) {
                    redo_pass = true;
                } else {
                    const bool anyrep = ballot(q.rep) != 0ull;
                    const int G = WC >> 2;
                    const float4 *pcb = reinterpret_cast<const float4 *>(L.tile + (size_t)(q.state * N) * WC);
                    const float4 *pcl = reinterpret_cast<const float4 *>(L.tile + (size_t)(q.state * N + q.l + 1) * WC);
                    const float4 *ppx = reinterpret_cast<const float4 *>(ring(q.ps));
                    const float4 *pb0 = reinterpret_cast<const float4 *>(L.tile);  // blank column of state 0 (repeat children)
                    float4 *pmy = reinterpret_cast<float4 *>(ring(q.buf));
                    const int g0 = lo >> 2, g1 = (hi - 1) >> 2;
                    int sg = lo_s >> 2;
                    float lab = kNegInf, sum = kNegInf, mx = kNegInf;
                    float4 pprev, bprev;  // the group before: the parent's rows 4g - 1, 4g - 2 and the blank of 4g - 1
                    {
                        const int sgm = sg == 0 ? G - 1 : sg - 1;
                        pprev = ppx[sgm];
                        bprev = pb0[sgm];
                    }
                    // one group of four rows at ring group `sgx`: its operands are asked for one group ahead (a wavefront
                    // alone on its SIMD waits ~100 cycles on every LDS round trip it does not cover itself);
                    // masked: rows outside [lo, hi) are skipped (first / last group)
                    struct Ops { float4 cb, cl, px, b0; };
                    auto gload = [&](int sgx, bool with_rep, Ops &o) __attribute__((always_inline)) {
                        o.cb = pcb[sgx];
                        o.cl = pcl[sgx];
                        o.px = ppx[sgx];
                        if (with_rep) o.b0 = pb0[sgx];
                    };
                    auto gcomp = [&](int gx, int sgx, bool masked, bool with_rep, const Ops &o) __attribute__((always_inline)) {
                        const float4 cb = o.cb, cl = o.cl, px = o.px;
                        float x0 = pprev.w, x1 = px.x, x2 = px.y, x3 = px.z;
                        if (with_rep) {
                            const float4 b0 = o.b0;
                            x0 = q.rep ? pprev.z + bprev.w : x0;
                            x1 = q.rep ? pprev.w + b0.x : x1;
                            x2 = q.rep ? px.x + b0.y : x2;
                            x3 = q.rep ? px.y + b0.z : x3;
                            bprev = b0;
                        }
                        float s0 = kNegInf, s1 = kNegInf, s2 = kNegInf, s3 = kNegInf;
                        const int t0 = gx << 2;
                        if (!masked || (t0 >= lo && t0 < hi)) { lab = cl.x + vmax_raw(lab, x0); sum = vmax_raw(lab, sum + cb.x); s0 = sum; }
                        if (!masked || (t0 + 1 >= lo && t0 + 1 < hi)) { lab = cl.y + vmax_raw(lab, x1); sum = vmax_raw(lab, sum + cb.y); s1 = sum; }
                        if (!masked || (t0 + 2 >= lo && t0 + 2 < hi)) { lab = cl.z + vmax_raw(lab, x2); sum = vmax_raw(lab, sum + cb.z); s2 = sum; }
                        if (!masked || (t0 + 3 >= lo && t0 + 3 < hi)) { lab = cl.w + vmax_raw(lab, x3); sum = vmax_raw(lab, sum + cb.w); s3 = sum; }
                        mx = vmax_raw(vmax_raw(mx, vmax_raw(s0, s1)), vmax_raw(s2, s3));
                        pmy[sgx] = make_float4(s0, s1, s2, s3);
                        pprev = px;
                    };
                    auto run = [&](bool with_rep) __attribute__((always_inline)) {
                        int g = g0;
                        Ops A, B2;
                        const bool head = (lo & 3) != 0 || g0 == g1;
                        if (head) {
                            gload(sg, with_rep, A);
                            gcomp(g, sg, true, with_rep, A);
                            ++g;
                            sg = sg + 1 == G ? 0 : sg + 1;
                        }
                        const int g_full_end = (hi & 3) != 0 ? g1 : g1 + 1;  // groups [g, g_full_end) are whole
                        while (g < g_full_end) {
                            // a run of groups up to the end of the ring or of the window, two per trip
                            int n = g_full_end - g;
                            n = n < G - sg ? n : G - sg;
                            int u = 0;
                            gload(sg, with_rep, A);
                            for (; u + 2 <= n; u += 2) {
                                gload(sg + u + 1, with_rep, B2);
                                gcomp(0, sg + u, false, with_rep, A);
                                if (u + 2 < n) gload(sg + u + 2, with_rep, A);
                                gcomp(0, sg + u + 1, false, with_rep, B2);
                            }
                            if (u < n) gcomp(0, sg + u, false, with_rep, A);
                            g += n;
                            sg = sg + n == G ? 0 : sg + n;
                        }
                        if (g <= g1) {
                            gload(sg, with_rep, A);
                            gcomp(g, sg, true, with_rep, A);
                        }
                    };
                    if (anyrep) run(true);
                    else run(false);
                    if (q.work) {
                        ff(F_MX, q.buf) = mx;
                        ff(F_LLAB, q.buf) = lab;
                        ff(F_LSUM, q.buf) = sum;
                        float *my = ring(q.buf);
                        my[slotn(lo - 1)] = kNegInf;  // the guard rows (the first group's store may have covered them)
                        my[slotn(lo - 2)] = kNegInf;
                    }
                }
                if (PROF && prof) n_redo += redo_pass ? 1u : 0u;
            }
#ifdef FCD_HIPEMU  // (developer aid under the emulator: FCD_EMU_SLOTS_STATS=1 counts how the passes were built)
            if (lane == 0) { ++g_emu_passes; g_emu_exact += redo_pass ? 1 : 0; }
#endif
            if (redo_pass) {
                wave_sync();
                for (int m0 = 0; m0 < n_new; m0 += kWave) build_exact(m0);
            }
            wave_sync();
        }
        // What was asked for at the head of the step -- the rows of read 2 the NEXT step gains, the next row of read 1 --
        // goes to LDS here, long after it arrived and BEFORE the hand-over's stores are issued: a wait on these loads behind
        // the stores would wait for the stores' acknowledgements too (the memory counter is in order).
        if (pf_hi > pf_lo && pf_hi - (lo > 0 ? lo - 1 : 0) <= WC && pf_lo == tl_hi) {
            const int n = (pf_hi - pf_lo) * SN;
            if (lane < n) L.tile[(size_t)l_sn * WC + slotn(pf_lo + l_rw)] = pf_val;
            unclean = unclean || ballot(lane < n && !(pf_val < __builtin_huge_valf())) != 0ull;
            tl_hi = pf_hi;
            if (tl_hi - tl_lo > WC) tl_lo = tl_hi - WC;
            pf_lo = pf_hi = 0;
        }
        if (more && lane < SN) L.f1n[lane] = f1_next;
        if (is_new && can) p2 = ff(F_MX, nbuf);
        nn += n_new;
        FCD_S_PHASE(3)
        FCD_S_FINE(8)  // builds
        if (nn > p.cap_nodes) return fail(FCD_ST_INTERNAL);

        // ---- merge is done (own candidates folded the three items); probability (:146-148), keys, exact rank ----
        bf_bad = false;
        float prob = la_fast(clp, cgp) + p2;
        if (MODE == FCD_LOGADD_LOGSUMEXP && ballot(bf_bad && valid) != 0ull) prob = la_exact(clp, cgp) + p2;
        const uint64_t key = valid ? (prob == prob ? make_key(prob, cid) : 1ull) : 0ull;
        l_keys[lane] = key;
        const uint64_t m_valid = ballot(valid);
        const int n_valid = popc64(m_valid);
        const bool any_nan = ballot(valid && prob != prob) != 0ull;
        if (n_valid >= 2 && any_nan) return fail(FCD_ST_INCOMPARABLE);  // :619-631
        if (n_valid == 0) return fail(FCD_ST_RAN_OUT_OF_BEAM);          // :633-636
        wave_sync();
        int rank;
        {
            int r0, r1, r2, r3;
            FCD_RANK4_FIRST(key, l_keys[0], l_keys[1], l_keys[2], l_keys[3], r0, r1, r2, r3);
            for (int u = 4; u < P; u += 4) FCD_RANK4(key, l_keys[u], l_keys[u + 1], l_keys[u + 2], l_keys[u + 3], r0, r1, r2, r3);
            rank = (r0 + r1) + (r2 + r3);
        }
        const int Bn = n_valid < BC ? n_valid : BC;
        FCD_S_FINE(9)  // probability, keys, exact rank
        // equal probabilities: candidates with one probability occupy consecutive ranks, so a KEPT candidate is tied
        // when ranks i and i + 1 hold one probability word for some i < beam_size; the tie can change the kept set or
        // the best entry when it sits at ranks 0 / 1 or across the truncation boundary
        bool any_kept_tie = false;
        if (count_amb || (pdq && n_valid > 20)) {
            if (valid) l_pw[rank] = (int)(uint32_t)(key >> 32);
            wave_sync();
            const bool pair_eq = lane + 1 < n_valid && l_pw[lane] == l_pw[lane + 1];
            const uint64_t m_eq = ballot(pair_eq);
            const uint64_t keptm = BC >= 64 ? ~0ull : ((1ull << BC) - 1ull);
            any_kept_tie = n_valid > 20 && (m_eq & keptm) != 0ull;
            if (count_amb) {
                if (any_kept_tie) ++n_amb;
                if ((m_eq & 1ull) != 0ull || (BC < 64 && ((m_eq >> (BC - 1)) & 1ull) != 0ull)) ++n_crit;
            }
            wave_sync();
        }
        if (pdq && any_kept_tie) {
            // sort_unstable_by's own order (src/duplex.rs:620,807): the merged candidates in ascending node order go
            // through the restated quicksort (one lane); the position of a candidate in its result is its rank
            int pos = 0;
            for (int j = 0; j < P; ++j) {
                const uint64_t kj = l_keys[j];
                pos += (kj != 0ull && (uint32_t)kj > (uint32_t)key) ? 1 : 0;  // low word: larger = smaller node
            }
            if (valid) l_pql[pos] = (key & 0xFFFFFFFF00000000ull) | (uint32_t)lane;
            wave_sync();
            if (lane == 0) pdq178::sort_desc(l_pql, n_valid, l_pqs);
            wave_sync();
            if (lane < n_valid) l_pw[(int)(uint32_t)l_pql[lane]] = lane;
            wave_sync();
            if (valid) rank = l_pw[lane];
            wave_sync();
        }
        FCD_S_SUB_BEGIN()
        FCD_S_FINE(10)  // ties

        // ---- the next beam: survivors keep (or get) a slot, everything else that was live leaves ----
        // All of it batched: a step evicts ~7 nodes and the wavefront is alone on its SIMD, so a loop over the evicted
        // slots (an LDS round trip, a wait and a few dozen instructions each) was 12-15 k cycles of a 20 k step (r06).
        const bool surv = valid && rank < Bn;
        const bool stale_in = surv && ck > 0 && !is_new;  // an existing child comes (back) into the beam
        const uint64_t m_stale = ballot(stale_in);
        const int n_stale = popc64(m_stale);
        int myslot = ck == 0 ? slot_i : nbuf;
        if (stale_in) myslot = l_flist[n_new + popc64(m_stale & lanemask_lt())];
        // records of the nodes coming back: asked for now, looked at after the evictions have been issued
        int4 s_meta = make_int4(0, 0, 0, 0), s_aux = make_int4(0, 0, 0, 0);
        int s_rows[kNLMax];
#pragma unroll
        for (int j = 0; j < kNLMax; ++j) s_rows[j] = -1;
        if (stale_in) {
            s_meta = load_int4_l2(g_meta((uint32_t)cid));
            s_aux = load_int4_l2(g_aux((uint32_t)cid));
#pragma unroll
            for (int j = 0; j < kNLMax; ++j)
                if (j < NL) s_rows[j] = load_i32_l2(g_rows((uint32_t)cid) + j);
        }
        // ... and their rings, the first two of them (a third one and later: after the evictions)
        float sr0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sr1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int st_a = n_stale > 0 ? (int)__builtin_ctzll(m_stale) : 0;
        const uint64_t m_stale2 = m_stale & (m_stale - 1);
        const int st_b = n_stale > 1 ? (int)__builtin_ctzll(m_stale2) : 0;
        const bool small_ring = WC <= 4 * kWave;
        if (n_stale > 0 && small_ring) {
            const float *aa = g_ring((uint32_t)rl_i(cid, st_a));
            const float *ab_ = g_ring((uint32_t)rl_i(cid, st_b));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = u * kWave + lane;
                if (x < WC) {
                    sr0[u] = load_f32_l2(aa + x);
                    if (n_stale > 1) sr1[u] = load_f32_l2(ab_ + x);
                }
            }
        }
        if (PROF && prof) n_enter += (uint32_t)n_stale;
        FCD_S_FINE(11)  // survivors, returning nodes' loads issued
        // what leaves: the beam entries that did not survive and the new nodes that did not make it.  A node comes back
        // only as the child of a beam entry, so it needs a proper ancestor in the beam -- none exists once every entry
        // of the next beam is at least as deep as the node, and then none ever will (its ancestors have left for good,
        // top-down from the root: the argument of beam_wave_step.inc's dead-row test).  Such a node leaves its
        // (parent, label) behind -- the final walk to the root may pass through it -- and nothing else: no ring, no
        // window record, no child row.
        const bool ev = (act && ck == 0 && !surv && node >= 0) || (is_new && can && !surv);  // (the root has no arena entry)
        const int my_depth = ck == 0 ? depth : depth + 1;
        int min_depth = surv ? my_depth : 0x7FFFFFFF;
#define FCD_DPP_IMIN(CTRL, RM)                                                                                  \
        {                                                                                                       \
            const int o_ = __builtin_amdgcn_update_dpp(min_depth, min_depth, CTRL, RM, 0xf, false);             \
            min_depth = o_ < min_depth ? o_ : min_depth;                                                        \
        }
        FCD_DPP_IMIN(0x111, 0xf) FCD_DPP_IMIN(0x112, 0xf) FCD_DPP_IMIN(0x114, 0xf) FCD_DPP_IMIN(0x118, 0xf)
        FCD_DPP_IMIN(0x142, 0xa) FCD_DPP_IMIN(0x143, 0xc)
#undef FCD_DPP_IMIN
        min_depth = rl_i(min_depth, 63);
        // ... unless one of its children stays: an entry whose parent has left reads the parent's last rows out of the
        // arena when it is next extended
        const int pr = ck == 0 ? prank_i : ci;  // rank of my parent in this step's beam (own candidates: if it is there)
        const int has_kid = __builtin_amdgcn_ds_permute(((surv && pr >= 0) ? pr * N : 63) << 2, 1);
        const bool dead = ev && my_depth <= min_depth && !(ck == 0 && has_kid != 0);
        const bool ev_ring = ev && !dead;
        const uint64_t m_ev = ballot(ev_ring);
        const int n_ev = popc64(m_ev);
        if (ev_ring) {
            const int qi = popc64(m_ev & lanemask_lt());
            l_bt[qi] = myslot;
            l_pw[qi] = cid;  // (ck == 0: cid is the entry's node)
        }
        // a survivor whose parent sat in this step's beam and is not in the next one remembers the parent's bounds
        {
            const int prc = (surv && pr >= 0) ? pr : 0;
            const int p_surv = bperm_i(prc * N, surv ? 1 : 0);
            const int p_slot = bperm_i(prc * N, slot_i);
            const int p_node = bperm_i(prc * N, node);
            if (surv && pr >= 0 && !p_surv && p_node >= 0) fg(G_PAR, myslot) = fg(G_WIN, p_slot);  // (word 3 rides along unused)
            if (stale_in) {  // (its slot's fields are all new)
                const int l = ck - 1;
                fg(G_ID, myslot) = make_int4(cid, l, node, crf ? (int)(((int64_t)state * NL) % S) + l : 0);  // :782
            }
        }
        // rank lanes of the next beam
        {
            const int dst = surv ? rank : 63;
            const int s2 = __builtin_amdgcn_ds_permute(dst << 2, myslot);
            const int n2 = __builtin_amdgcn_ds_permute(dst << 2, cid);
            const int l2 = __builtin_amdgcn_ds_permute(dst << 2, __float_as_int(clp));
            const int g2 = __builtin_amdgcn_ds_permute(dst << 2, __float_as_int(cgp));
            slotE = s2; nodeE = n2; lpE = __int_as_float(l2); gpE = __int_as_float(g2);
        }
        wave_sync();
        // a state outside [0, S) is an ndarray index panic in the reference when the entry is next expanded (:749) --
        // which never happens for the entries the last row leaves behind
        if (crf && more) {
            const bool bs = lane < Bn && fi(F_STATE, slotE) >= S;
            if (ballot(bs) != 0ull) return fail(FCD_ST_BAD_STATE);
        }
        FCD_S_FINE(12)  // eviction lists, parents' bounds, rank lanes
        // ---- nodes coming back: ring and record from the arena into their slot ----
        // (BEFORE the evictions' stores are issued: a wait on these loads behind the stores would wait for the stores too)
        if (n_stale > 0 && small_ring) {
            float *da = ring(rl_i(myslot, st_a)), *db = ring(rl_i(myslot, st_b));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = u * kWave + lane;
                if (x < WC) {
                    da[x] = sr0[u];
                    if (n_stale > 1) db[x] = sr1[u];
                }
            }
        }
        for (uint64_t m = small_ring ? (m_stale2 & (m_stale2 - 1)) : m_stale; m != 0ull; m &= m - 1) {
            const int src = (int)__builtin_ctzll(m);
            const int s_ = rl_i(myslot, src), nd = rl_i(cid, src);
            float *dst = ring(s_);
            const float *a_ = g_ring((uint32_t)nd);
            for (int x0 = 0; x0 < WC; x0 += 4 * kWave) {  // four loads in flight per lane
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = x0 + u * kWave + lane;
                    v[u] = x < WC ? load_f32_l2(a_ + x) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = x0 + u * kWave + lane;
                    if (x < WC) dst[x] = v[u];
                }
            }
        }
        if (stale_in) {
            fg(G_WIN, myslot) = make_int4(s_meta.z, s_meta.w, s_aux.y, (!crf && node >= 0 && tip == ck - 1) ? 1 : 0);
            fg(G_VAL, myslot) = make_int4(s_aux.x, s_aux.z, depth + 1, s_aux.w);
            fg(G_KID, myslot) = make_int4(s_rows[0], s_rows[1], s_rows[2], s_rows[3]);
            if (NL > 4) fg(G_KID + 1, myslot) = make_int4(s_rows[4], s_rows[5], s_rows[6], -1);
        }
        FCD_S_FINE(13)  // returning nodes landed
        // ---- evictions: records, by the candidate lane whose node leaves ----
        if (ev) {
            const int s_ = myslot, nd = cid;
            const int4 e_id = fg(G_ID, s_), e_win = fg(G_WIN, s_);
            *g_meta((uint32_t)nd) = make_int4(e_id.z, e_id.y, e_win.x, e_win.y);
            if (!dead) {
                const int4 e_val = fg(G_VAL, s_);
                *g_aux((uint32_t)nd) = make_int4(e_val.x, e_win.z, e_val.y, e_val.w);
                int4 *rw = reinterpret_cast<int4 *>(g_rows((uint32_t)nd));  // (NLp = 4 or 8 words, 16-byte aligned)
                rw[0] = fg(G_KID, s_);
                if (NLp > 4) rw[1] = fg(G_KID + 1, s_);
            }
        }
        // ---- evictions: rings, 16 bytes per lane and item, four items in flight per lane (one LDS round trip for the
        // lists, one for the rings, then the stores -- which nothing in this step waits for) ----
        {
            const int G = WC >> 2;
            const int total = n_ev * G;
            int q = g_q0, c = g_c0;  // ring and 16-byte piece of item `lane`
            for (int x0 = 0; x0 < total; x0 += 4 * kWave) {
                int qs[4], cs[4], ss[4], ns[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    qs[u] = q;
                    cs[u] = c;
                    on[u] = x0 + u * kWave + lane < total;
                    q += g_dq;
                    c += g_dc;
                    if (c >= G) { c -= G; ++q; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ss[u] = on[u] ? l_bt[qs[u]] : P;
                    ns[u] = on[u] ? l_pw[qs[u]] : 0;
                }
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4 *>(ring(ss[u]))[cs[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (on[u]) reinterpret_cast<float4 *>(g_ring((uint32_t)ns[u]))[cs[u]] = v[u];
            }
        }
        FCD_S_SUB(1)
        FCD_S_FINE(14)  // evictions + returning nodes
        // ---- free list of the coming step: the slots no entry of the next beam sits in ----
        {
            bool used = false;
            for (int j = 0; j < Bn; ++j) {
                const int sj = rl_i(slotE, j);
                used = used | (sj == lane);
            }
            const bool is_free = lane < P && !used;
            const uint64_t fm = ballot(is_free);
            if (is_free) l_flist[popc64(fm & lanemask_lt())] = lane;
        }
        { float *t_ = L.f1; L.f1 = L.f1n; L.f1n = t_; }
        B = Bn;
        wave_sync();
        FCD_S_PHASE(4)
        FCD_S_FINE(15)  // free list, end of step
    }
    if (PROF && prof && lane == 0) {
        uint32_t *o = p.prof + kProfWords * r;
#ifdef FCD_SLOTS_FINE
        for (int k = 0; k < 16; ++k) o[16 + k] = (uint32_t)(fine[k] >> 6);
#endif
        for (int k = 0; k < 5; ++k) o[k] = (uint32_t)(acc[k] >> 6);  // units of 64 cycles
        o[5] = n_iter;
        o[6] = n_newnodes;
        o[7] = (uint32_t)T1;
        o[8] = n_slow;
        o[9] = n_enter;
        o[10] = n_ext;
        for (int k = 0; k < 5; ++k) o[11 + k] = (uint32_t)(sub[k] >> 6);
    }
    (void)n_redo;

    // ---- labels leaf -> root (:638-649), written in sequence order ----
    // entries still in the beam have no arena record yet: the walk needs (parent, label) of the best node's ancestors
    if (lane < B && nodeE >= 0) *g_meta((uint32_t)nodeE) = make_int4(fi(F_PAR, slotE), fi(F_TIP, slotE), fi(F_OFF, slotE), fi(F_END, slotE));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    {
        const int s0 = rl_i(slotE, 0);
        const int n = fi(F_DEPTH, s0);
        if (lane == 0) {
            int q = rl_i(nodeE, 0);
            for (int j = n - 1; j >= 0; --j) {
                const int4 mq = load_int4_l2(g_meta((uint32_t)q));
                lab_out[j] = (uint8_t)(mq.y + 1);
                q = mq.x;
            }
            p.out.out_len[r] = (uint32_t)n;
            p.out.status[r] = FCD_ST_OK;
            if (count_amb) {
                p.out.ambiguous[2 * r] = (uint32_t)n_amb;
                p.out.ambiguous[2 * r + 1] = (uint32_t)n_crit;
            }
        }
    }
}

}  // namespace

bool duplex_slots_supported(int beam_size, int N, int S, int width, int tie_order) {
    if (beam_size < 1 || N < 2 || N > 8 || beam_size > kBeamMax || (int64_t)beam_size * N > kSlots) return false;
    const int WC = duplex_slots_ring_rows(width);
    return duplex_slots_lds_bytes(beam_size, N, S, WC, tie_order) <= 64 * 1024;
}

size_t duplex_slots_pair_bytes(int64_t cap_nodes, int N, int ring_rows, int64_t T2cap) {
    const int NLp = (N - 1 + 3) & ~3;
    const size_t b = (size_t)cap_nodes * (32 + (size_t)ring_rows * 4 + (size_t)NLp * 4) + (size_t)(T2cap + 1) * 4;
    return (b + 255) & ~(size_t)255;
}

int duplex_slots_ring_rows(int width) { return ((width > 1 ? width : 1) + 4 + 3) & ~3; }

size_t duplex_slots_lds_bytes(int beam_size, int N, int S, int WC, int tie_order) {
    (void)tie_order;  // (the quicksort's list and scratch are part of the fixed tables: under 1 KiB)
    return slds_words(beam_size, N, S, WC) * 4 + 16;
}

hipError_t launch_duplex_slots(const DuplexArgs &a, int64_t pair_begin, int64_t n_pairs, hipStream_t stream) {
    if (n_pairs <= 0) return hipSuccess;
    SlotParams p;
    p.ln1 = a.ln1; p.ln2 = a.ln2; p.T1cap = a.T1cap; p.T2cap = a.T2cap;
    p.len1 = a.len1; p.len2 = a.len2; p.env = a.env; p.env_stride = a.env_stride;
    p.N = a.N; p.beam_size = a.beam_size; p.thr_ln = a.thr_ln; p.collapse = a.collapse;
    p.S = a.S; p.crf = a.crf; p.init1 = a.init1; p.init2 = a.init2; p.n_init1 = a.n_init1;
    p.n_init2 = a.n_init2; p.init1_stride = a.init1_stride; p.init2_stride = a.init2_stride;
    p.slab = reinterpret_cast<char *>(a.meta); p.pair_stride = a.pair_stride;
    p.cap_nodes = a.cap_nodes; p.Wcap4 = a.Wcap; p.NLp = a.NLp;
    p.out = a.out; p.pair_begin = pair_begin; p.prof = a.prof; p.tie_order = a.tie_order;
    static const int env_prefetch = [] {
        // "1": on.  Off by default: on BASELINE config 5 the touch costs the expansion 0.7 k cycles per step and saves the
        // hand-over 0.3 k (profiles/r06g_duplex_account_*.jsonl) -- the arena lines of a node evicted a few steps ago are
        // still in L2 / MALL more often than not
        const char *e = getenv("FCD_DUPLEX_PREFETCH");
        return e && !strcmp(e, "1") ? 1 : 0;
    }();
    p.prefetch = env_prefetch;
    const size_t lds = duplex_slots_lds_bytes(a.beam_size, a.N, a.S, a.Wcap, a.tie_order);
    const dim3 grid((unsigned)n_pairs), block(64);
#define FCD_SLOTS_LAUNCH(MODE)                                                                                     \
    do {                                                                                                           \
        if (a.prof) hipLaunchKernelGGL((duplex_slots_kernel<MODE, true>), grid, block, lds, stream, p);            \
        else hipLaunchKernelGGL((duplex_slots_kernel<MODE, false>), grid, block, lds, stream, p);                  \
    } while (0)
    if (a.mode == FCD_LOGADD_LOGSUMEXP_GLIBC235) FCD_SLOTS_LAUNCH(FCD_LOGADD_LOGSUMEXP_GLIBC235);
    else if (a.mode == FCD_LOGADD_MAX) FCD_SLOTS_LAUNCH(FCD_LOGADD_MAX);
    else FCD_SLOTS_LAUNCH(FCD_LOGADD_LOGSUMEXP);
#undef FCD_SLOTS_LAUNCH
#ifdef FCD_HIPEMU
    if (getenv("FCD_EMU_SLOTS_STATS")) fprintf(stderr, "duplex_slots: %ld build passes so far, %ld on the exact form\n", g_emu_passes, g_emu_exact);
#endif
    return hipGetLastError();
}

}  // namespace fcd
