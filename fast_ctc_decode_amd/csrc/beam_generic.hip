// beam_generic.hip -- LDS-resident CTC prefix beam search, one read per wavefront.
//
// Covers search::beam_search (/root/reference/src/search.rs:159-301) and
// search::crf_beam_search (:38-157) for ANY beam_size / alphabet / CRF state count that fits
// the per-wave LDS budget.  It is the fallback and the correctness workhorse; the
// register-resident kernel in beam_wave.hip is the fast path for beam_size <= 8, N <= 7.
//
// Mapping (CDNA4, wave64): one 64-thread workgroup == one wavefront == one read, so all
// synchronisation is wave-local.  The beam lives in LDS as a structure of arrays, double
// buffered.  Every timestep the wave evaluates B*N "slots" (beam entry i, k): k == 0 is the
// entry's own node (blank + repeat-stay + the extension arriving from its parent if that parent
// is in the beam too), k >= 1 is the child reached by label k-1.  This yields each distinct tree
// node exactly once, already merged, which is what the reference gets by sorting the raw
// candidates by node and folding duplicates (:245-260); at most two non-zero f32 addends meet
// in any sum, so the merged value is independent of the reference's fold order.
// Pruning ranks the slots by a 64-bit key (probability desc, node asc) -- exact, no sort.
//
// The prefix tree is an append-only per-read arena in HBM (tree.rs:125-161): node ids are
// handed out in the reference's creation order (beam order, then label order) with a
// ballot/popcount prefix sum, so ties resolve exactly as in the reference.
#include "device_utils.h"
#include "fcd_internal.h"
#include "pdq178.h"

namespace fcd {

namespace {

struct GenericParams {
    BatchDesc in;
    BeamArgs a;
    GenericArena arena;
    ResultDesc out;
    int64_t read_begin;
};

// LDS carve-up (all 4-byte words unless noted).  BC = beam_size, NL = N-1, C = BC*N.
// NB: no runtime-indexed pointer arrays in here -- `ptr[cur]` with a runtime `cur` would push the
// whole struct into scratch memory (every access a vector-memory round trip).  The two beam buffers
// are addressed arithmetically instead: buffer b starts at beam0 + b * beam_stride.
struct Lds {
    int *beam0;       // [2][beam_stride] words: node, lp, gp, tip, par, state, depth (BC each), child (BC*NL)
    int beam_stride;
    int BC;
    __device__ __forceinline__ int *b_node(int b) const { return beam0 + b * beam_stride; }
    __device__ __forceinline__ float *b_lp(int b) const { return reinterpret_cast<float *>(b_node(b) + BC); }
    __device__ __forceinline__ float *b_gp(int b) const { return reinterpret_cast<float *>(b_node(b) + 2 * BC); }
    __device__ __forceinline__ int *b_tip(int b) const { return b_node(b) + 3 * BC; }
    __device__ __forceinline__ int *b_par(int b) const { return b_node(b) + 4 * BC; }
    __device__ __forceinline__ int *b_state(int b) const { return b_node(b) + 5 * BC; }
    __device__ __forceinline__ int *b_depth(int b) const { return b_node(b) + 6 * BC; }
    __device__ __forceinline__ int *b_child(int b) const { return b_node(b) + 7 * BC; }  // BC*NL
    uint64_t *c_key;  // C, 8-byte aligned
    float *c_lp;
    float *c_gp;
    int *c_id;
    uint64_t *c_newmask;  // ceil(C/64) ballots: slot c created its node in this step
    int *nb_src;   // BC
    int *b_pslot;  // BC: beam slot of the entry's parent, or -1
    // m_flag, hist and l_key are never live at the same time and share one region:
    int *m_flag;   // C: 1 when slot (i,k)'s child is itself a beam entry (its own slot absorbs the extension)
    float *row;    // N (non-CRF staging of the current posterior row)
    float *top;    // 1
    int *hist;          // kBuckets: candidates per probability bucket (prune pre-selection)
    uint64_t *l_key;    // list_cap(BC): keys of the candidates that can still reach the beam
    int *l_c;           // list_cap(BC): their slot index
    // FCD_TIE_PDQ178 (pdq178.h): the node-ordered candidate list of a tie-flagged step and the quicksort's scratch
    uint64_t *pq_list;  // C
    pdq178::Scratch *pq_scr;
};

// Prune pre-selection (phase B): candidates are bucketed by how far their probability lies below the
// step's maximum, in units of 2^kBucketShift steps of the orderable f32 bit pattern (1/32 binade);
// everything 8 binades or more below the maximum shares the last bucket.
constexpr int kBuckets = 256;
constexpr int kBucketShift = 18;
__host__ __device__ inline int list_cap(int BC) { return BC + 64; }

__host__ __device__ inline size_t shared_region_words(int BC, int N) {
    size_t w = (size_t)BC * N;                                   // m_flag
    if (w < (size_t)kBuckets) w = kBuckets;                      // hist
    if (w < 2 * (size_t)list_cap(BC)) w = 2 * (size_t)list_cap(BC);  // l_key
    return (w + 3) & ~(size_t)3;
}

// pdq: the tie order is FCD_TIE_PDQ178 and a step can hold more than 20 candidates -- only then the node-ordered list and
// the quicksort's scratch exist (reserved unconditionally they cost the stable order a sixth of its largest beam)
__host__ __device__ inline size_t lds_words(int BC, int N, bool pdq) {
    int NL = N - 1;
    size_t C = (size_t)BC * N;
    size_t w = 0;
    w += 2 * (size_t)BC * (7 + NL);
    w += 2 * C;  // keys (u64)
    w += 3 * C;  // lp, gp, id
    w += 2 * ((C + 63) / 64) + 2;  // newmask (u64)
    w += BC;     // nb_src
    w += BC;     // b_pslot
    w += N;      // row
    w += 2;      // top + pad
    w += shared_region_words(BC, N) + 3;  // m_flag / hist / l_key, 16-byte aligned
    w += (size_t)list_cap(BC);            // l_c
    if (pdq) w += 2 * C + 2 + (sizeof(pdq178::Scratch) + 3) / 4;  // pq_list (u64, 8-byte aligned) + pq_scr
    return w;
}

__device__ inline Lds carve(int *smem, int BC, int N) {
    Lds L;
    int NL = N - 1;
    size_t C = (size_t)BC * N;
    uint64_t *k = reinterpret_cast<uint64_t *>(smem);
    L.c_key = k;
    int *p = smem + 2 * C;
    L.c_lp = reinterpret_cast<float *>(p); p += C;
    L.c_gp = reinterpret_cast<float *>(p); p += C;
    L.c_id = p; p += C;
    if ((p - smem) & 1) ++p;
    L.c_newmask = reinterpret_cast<uint64_t *>(p); p += 2 * ((C + 63) / 64);
    L.BC = BC;
    L.beam_stride = BC * (7 + NL);
    L.beam0 = p;
    p += 2 * (size_t)L.beam_stride;
    L.nb_src = p; p += BC;
    L.b_pslot = p; p += BC;
    L.row = reinterpret_cast<float *>(p); p += N;
    L.top = reinterpret_cast<float *>(p); p += 2;
    while ((p - smem) & 3) ++p;  // smem is 16-byte aligned: hist is cleared / scanned as int4
    L.m_flag = p;
    L.hist = p;
    L.l_key = reinterpret_cast<uint64_t *>(p);
    p += shared_region_words(BC, N);
    L.l_c = p;
    p += list_cap(BC);
    if ((p - smem) & 1) ++p;
    L.pq_list = reinterpret_cast<uint64_t *>(p);
    p += 2 * C;
    L.pq_scr = reinterpret_cast<pdq178::Scratch *>(p);
    return L;
}

// One workgroup == one wavefront, and a wave's LDS instructions execute in issue order: ordering LDS
// traffic between phases only needs the COMPILER to keep it in order.  __syncthreads() would also
// drain vmcnt -- every outstanding tree store and posterior load -- several times per timestep.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void fail(const GenericParams &p, int64_t r, int code, int n_amb, int n_crit) {
    if (threadIdx.x == 0) {
        p.out.status[r] = code;
        p.out.out_len[r] = 0;
        if (p.out.ambiguous) {
            p.out.ambiguous[2 * r] = (uint32_t)n_amb;
            p.out.ambiguous[2 * r + 1] = (uint32_t)n_crit;
        }
    }
}

__global__ __launch_bounds__(64) void beam_generic_kernel(GenericParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x;
    const int64_t r = p.read_begin + blockIdx.x;
    const int N = p.in.N, NL = N - 1, S = p.in.S;
    const int BC = p.a.beam_size;
    const bool crf = p.a.crf != 0;
    const bool collapse = p.a.collapse != 0;
    const float thr = p.a.thr;
    Lds L = carve(smem, BC, N);

    int64_t T = p.in.T;
    if (p.in.lengths) {
        int64_t t = p.in.lengths[r];
        T = t < 0 ? 0 : (t < T ? t : T);
    }
    const int dt = p.in.dtype;
    const float *post = post_at(p.in.post, r * p.in.stride_read, dt);
    const int64_t st_t = p.in.stride_t, st_s = p.in.stride_s, st_n = p.in.stride_n;
    int4 *rec = p.arena.rec + (int64_t)blockIdx.x * p.arena.cap_nodes;
    int32_t *rows = p.arena.rows + (int64_t)blockIdx.x * p.arena.cap_nodes * NL;

    // ---- initial beam: search.rs:170-175 / :54-59 ----
    int cur = 0;
    int B = 1;
    if (lane == 0) {
        int st0 = 0;
        float lp0 = 0.0f, gp0 = 1.0f;
        bool bad = false;
        if (crf) {
            // ndarray-stats argmax/max: first maximum wins, NaN -> Err -> unwrap panics
            const float *init = p.a.init + r * p.a.init_stride;
            float m = init[0];
            bad = (p.a.n_init <= 0) || (m != m);
            for (int64_t j = 1; j < p.a.n_init && !bad; ++j) {
                float e = init[j];
                if (e != e) bad = true;
                if (e > m) {
                    m = e;
                    st0 = (int)j;
                }
            }
            lp0 = m;
            gp0 = init[0];
        }
        L.b_node(0)[0] = -1;
        L.b_lp(0)[0] = lp0;
        L.b_gp(0)[0] = gp0;
        L.b_tip(0)[0] = -1;
        L.b_par(0)[0] = -2;
        L.b_state(0)[0] = bad ? -1 : st0;
        L.b_depth(0)[0] = 0;
    }
    for (int j = lane; j < NL; j += kWave) L.b_child(0)[j] = -1;
    if (!crf && T > 0)
        for (int j = lane; j < N; j += kWave) L.row[j] = load_post(post, j * st_n, dt);
    // row t+1 travels in a register while step t runs, so its HBM latency is never waited for
    // (alphabets above 64 labels fall back to loading it when it is needed)
    const bool row_in_reg = !crf && N <= kWave;
    float next_row = (row_in_reg && lane < N && T > 1) ? load_post(post, st_t + lane * st_n, dt) : 0.0f;
    wave_sync();

    int nn = 0;  // nodes in this read's tree (wave-uniform)
    // tie instrument (fcd_result.ambiguous; two counters, semantics in include/fcd.h)
    const bool count_amb = p.out.ambiguous != nullptr;
    int n_amb = 0, n_crit = 0;
    const bool pdq = p.a.tie_order == FCD_TIE_PDQ178;  // equal probabilities above 20 candidates: Rust 1.78's order

    for (int64_t t = 0; t < T; ++t) {
        int *b_node = L.b_node(cur), *b_tip = L.b_tip(cur), *b_par = L.b_par(cur);
        int *b_state = L.b_state(cur), *b_depth = L.b_depth(cur), *b_child = L.b_child(cur);
        float *b_lp = L.b_lp(cur), *b_gp = L.b_gp(cur);
        const int nslots = B * N;
        const float *frame = post_at(post, t * st_t, dt);

        int n_valid = 0;
        bool any_nan = false, bad_state = false;

        // ---- phase P: where is each entry's parent in the beam?  O(B^2/64) once per step instead
        // of a beam search per slot: entry e with parent slot j marks slot (j, tip_e + 1) as merged
        for (int c = lane; c < nslots; c += kWave) L.m_flag[c] = 0;
        wave_sync();
        for (int e = lane; e < B; e += kWave) {
            int ps = -1;
            if (b_node[e] >= 0) {
                const int par = b_par[e];
                for (int j = 0; j < B; ++j)
                    if (b_node[j] == par) {
                        ps = j;
                        break;
                    }
                if (ps >= 0) L.m_flag[ps * N + b_tip[e] + 1] = 1;
            }
            L.b_pslot[e] = ps;
        }
        wave_sync();

        // ---- phase A: evaluate every (entry, k) slot; number the new nodes ----
        for (int base = 0; base < nslots; base += kWave) {
            const int c = base + lane;
            const bool act = c < nslots;
            const int i = act ? c / N : 0;
            const int k = act ? c - i * N : 0;
            const int node = b_node[i];
            const float lp = b_lp[i], gp = b_gp[i];
            const int tip = b_tip[i];
            const int st = b_state[i];
            bool valid = false, is_new = false;
            float clp = 0.0f, cgp = 0.0f;
            int cid = -2;
            if (act) {
                bool ok_state = !crf || (st >= 0 && st < S);
                if (!ok_state) {
                    bad_state = true;
                } else if (k == 0) {
                    // the entry's own node: blank (:191-198) + repeat-stay (:206-211)
                    const float pr0 = crf ? load_post(frame, st * st_s, dt) : L.row[0];
                    const bool blank = pr0 > thr;
                    cgp = blank ? (lp + gp) * pr0 : 0.0f;
                    bool stay = !crf && collapse && tip >= 0;
                    if (stay) {
                        const float pt = L.row[tip + 1];
                        stay = !(pt < thr);
                        clp = stay ? lp * pt : 0.0f;
                    }
                    // extension arriving from this node's parent, if the parent is in the beam
                    bool inc = false;
                    const int j = L.b_pslot[i];
                    if (j >= 0) {
                        const int stj = b_state[j];
                        if (crf && (stj < 0 || stj >= S)) {
                            bad_state = true;
                        } else {
                            const float pl = crf ? load_post(frame, stj * st_s + (tip + 1) * st_n, dt) : L.row[tip + 1];
                            if (!(pl < thr)) {  // :201 skip only if pr_b < thr
                                const bool rep = !crf && collapse && b_tip[j] == tip;
                                const float lpj = b_lp[j], gpj = b_gp[j];
                                const float contrib = rep ? gpj * pl : (lpj + gpj) * pl;
                                clp = clp + contrib;
                                inc = true;
                            }
                        }
                    }
                    valid = blank || stay || inc;
                    cid = node;
                } else {
                    // child by label k-1 (:200-239 / crf :84-99)
                    const int l = k - 1;
                    const float pk = crf ? load_post(frame, st * st_s + k * st_n, dt) : L.row[k];
                    const bool pass = !(pk < thr);
                    const bool rep = !crf && collapse && l == tip;
                    const float contrib = rep ? gp * pk : (lp + gp) * pk;
                    const int ch = b_child[i * NL + l];
                    const bool exists = ch >= 0;
                    valid = pass && (exists || !rep || gp > 0.0f);  // :212-218
                    // child already in the beam: its own slot absorbs this extension
                    if (valid && exists && L.m_flag[c]) valid = false;
                    is_new = valid && !exists;
                    clp = contrib;
                    cid = ch;
                }
            }
            // tree.rs:125-145 add_node, ids in (beam order, label order)
            const uint64_t m_new = __ballot(is_new);
            if (is_new) {
                cid = nn + popc64(m_new & lanemask_lt());
                if (cid < p.arena.cap_nodes) {
                    const int l = k - 1;
                    rec[cid] = make_int4(node, (int)t, l, b_depth[i] + 1);
                    for (int j = 0; j < NL; ++j) rows[(int64_t)cid * NL + j] = -1;
                    if (node >= 0) rows[(int64_t)node * NL + l] = cid;
                    b_child[i * NL + l] = cid;
                }
            }
            nn += popc64(m_new);
            if (lane == 0) L.c_newmask[base >> 6] = m_new;
            const float prob = clp + cgp;
            if (act) {
                L.c_lp[c] = clp;
                L.c_gp[c] = cgp;
                L.c_id[c] = cid;
                // a NaN key is only ever ranked when it is the lone candidate (never compared, :262)
                L.c_key[c] = valid ? (prob == prob ? make_key(prob, cid) : 1ull) : 0ull;
            }
            n_valid += popc64(__ballot(valid));
            any_nan = any_nan || (__ballot(valid && prob != prob) != 0ull);
        }
        bad_state = __ballot(bad_state) != 0ull;
        if (bad_state) return fail(p, r, FCD_ST_BAD_STATE, n_amb, n_crit);
        if (nn > p.arena.cap_nodes) return fail(p, r, FCD_ST_INTERNAL, n_amb, n_crit);
        // search.rs:261-277: any NaN among >= 2 candidates -> IncomparableValues, then empty -> RanOutOfBeam
        if (n_valid >= 2 && any_nan) return fail(p, r, FCD_ST_INCOMPARABLE, n_amb, n_crit);
        if (n_valid == 0) return fail(p, r, FCD_ST_RAN_OUT_OF_BEAM, n_amb, n_crit);
        wave_sync();

        // ---- phase B: the top beam_size candidates, in exact key order, build the next beam ----
        // Ranking all C = B*N candidates against each other costs O(C^2/64) per step.  Instead a
        // histogram over "distance below the maximum probability" finds the bucket holding the
        // beam_size-th largest candidate; only the candidates in that bucket or above (Lc of them,
        // usually beam_size + 1 or 2) are compacted into a list and ranked exactly on the 64-bit key.
        // Equal probabilities share a bucket, so ties still resolve by node index.  If the list would
        // not fit (heavily tied or extremely spread probabilities) the step falls back to all-pairs.
        const int nxt = cur ^ 1;
        const int Bn = n_valid < BC ? n_valid : BC;
        const int cap = list_cap(BC);
        int bstar = kBuckets - 1;  // accept every valid candidate (n_valid <= BC)
        int Lc = n_valid;
        uint32_t mx = 0;
        bool tie = false, crit = false;
        if (n_valid > BC) {
            for (int c = lane; c < nslots; c += kWave) {
                const uint32_t hi = (uint32_t)(L.c_key[c] >> 32);
                mx = hi > mx ? hi : mx;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t other = (uint32_t)__shfl_xor((int)mx, o);
                mx = other > mx ? other : mx;
            }
            *reinterpret_cast<int4 *>(L.hist + 4 * lane) = make_int4(0, 0, 0, 0);
            wave_sync();
            for (int c = lane; c < nslots; c += kWave) {
                const uint64_t key = L.c_key[c];
                if (key != 0ull) {
                    const uint32_t d = (mx - (uint32_t)(key >> 32)) >> kBucketShift;
                    atomicAdd(&L.hist[d < (uint32_t)(kBuckets - 1) ? d : (uint32_t)(kBuckets - 1)], 1);
                }
            }
            wave_sync();
            const int4 h = *reinterpret_cast<const int4 *>(L.hist + 4 * lane);
            const int s0 = h.x, s1 = s0 + h.y, s2 = s1 + h.z, s3 = s2 + h.w;
            int incl = s3;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const int excl = incl - s3;
            // exactly one lane owns the bucket where the running count reaches BC (the total is n_valid > BC)
            const bool cross = excl < BC && incl >= BC;
            const int kk = (excl + s0 >= BC) ? 0 : (excl + s1 >= BC) ? 1 : (excl + s2 >= BC) ? 2 : 3;
            const int cum = excl + (kk == 0 ? s0 : kk == 1 ? s1 : kk == 2 ? s2 : s3);
            const int owner = __builtin_ctzll(__ballot(cross));
            bstar = __shfl(4 * lane + kk, owner);
            Lc = __shfl(cum, owner);
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
                L.b_depth(nxt)[rank] = b_depth[i];
            } else {
                L.b_tip(nxt)[rank] = k - 1;
                L.b_par(nxt)[rank] = b_node[i];
                L.b_state(nxt)[rank] = crf ? (int)(((int64_t)b_state[i] * NL) % S) + (k - 1) : 0;
                L.b_depth(nxt)[rank] = b_depth[i] + 1;
            }
            L.nb_src[rank] = c | ((int)((L.c_newmask[c >> 6] >> (c & 63)) & 1ull) << 30);
            if (rank == 0) *L.top = L.c_lp[c] + L.c_gp[c];
        };
        // a kept candidate tied with another one: counted by the tie instrument, and re-ranked below under PDQ178
        const bool want_tie = count_amb || (pdq && n_valid > 20);
        if (Lc <= cap) {
            // compaction in slot order (the list overwrites the histogram), then exact rank inside the list
            wave_sync();
            int base = 0;
            for (int base0 = 0; base0 < nslots; base0 += kWave) {
                const int c = base0 + lane;
                const uint64_t key = c < nslots ? L.c_key[c] : 0ull;
                bool in = key != 0ull;
                if (in && n_valid > BC) {
                    const uint32_t d = (mx - (uint32_t)(key >> 32)) >> kBucketShift;
                    in = (int)(d < (uint32_t)(kBuckets - 1) ? d : (uint32_t)(kBuckets - 1)) <= bstar;
                }
                const uint64_t m_in = __ballot(in);
                if (in) {
                    const int pos = base + popc64(m_in & lanemask_lt());
                    L.l_key[pos] = key;
                    L.l_c[pos] = c;
                }
                base += popc64(m_in);
            }
            wave_sync();
            for (int e = lane; e < Lc; e += kWave) {
                const uint64_t key = L.l_key[e];
                int rank = 0;
                for (int j = 0; j < Lc; ++j) rank += (L.l_key[j] > key) ? 1 : 0;
                if (want_tie) {
                    // equal probabilities share a bucket: every candidate tied with a kept one is in the list,
                    // and so is every candidate of greater probability
                    int n_eq = 0, n_gt = 0;
                    for (int j = 0; j < Lc; ++j) {
                        const uint32_t hj = (uint32_t)(L.l_key[j] >> 32), he = (uint32_t)(key >> 32);
                        n_eq += hj == he ? 1 : 0;
                        n_gt += hj > he ? 1 : 0;
                    }
                    tie = tie || (rank < BC && n_valid > 20 && n_eq >= 2);
                    crit = crit || (n_eq >= 2 && (n_gt == 0 || (n_gt < BC && n_gt + n_eq > BC)));
                }
                if (rank < BC) emit(L.l_c[e], rank);
            }
        } else {
        // every lane owns slots lane, lane+64, ...: rank up to four of them in one sweep over the keys
        for (int base0 = 0; base0 < nslots; base0 += 4 * kWave) {
            uint64_t myk[4];
            int myr[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = base0 + u * kWave + lane;
                myk[u] = c < nslots ? L.c_key[c] : 0ull;
            }
            for (int j = 0; j < nslots; ++j) {
                const uint64_t kj = L.c_key[j];
#pragma unroll
                for (int u = 0; u < 4; ++u) myr[u] += (kj > myk[u]) ? 1 : 0;
            }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = base0 + u * kWave + lane;
            const uint64_t key = myk[u];
            const int rank = myr[u];
            if (c >= nslots || key == 0ull) continue;
            if (want_tie) {
                int n_eq = 0, n_gt = 0;
                for (int j = 0; j < nslots; ++j) {
                    const uint64_t kj = L.c_key[j];
                    n_eq += (kj != 0ull && (uint32_t)(kj >> 32) == (uint32_t)(key >> 32)) ? 1 : 0;
                    n_gt += ((uint32_t)(kj >> 32) > (uint32_t)(key >> 32)) ? 1 : 0;
                }
                tie = tie || (rank < BC && n_valid > 20 && n_eq >= 2);
                crit = crit || (n_eq >= 2 && (n_gt == 0 || (n_gt < BC && n_gt + n_eq > BC)));
            }
            if (rank < BC) emit(c, rank);
          }
        }
        }
        wave_sync();
        const bool any_tie = __ballot(tie) != 0ull;
        if (count_amb) {
            n_amb += any_tie ? 1 : 0;
            n_crit += __ballot(crit) != 0ull ? 1 : 0;
        }
        if (pdq && any_tie) {
            // sort_unstable_by's own order (src/search.rs:122,262): the merged candidates in ascending node order go
            // through the restated quicksort, one lane, and the first BC of its result ARE the next beam (every
            // entry written above is written again)
            for (int base0 = 0; base0 < nslots; base0 += kWave) {
                const int c = base0 + lane;
                const uint64_t key = c < nslots ? L.c_key[c] : 0ull;
                if (key == 0ull) continue;
                int pos = 0;
                for (int j = 0; j < nslots; ++j) {
                    const uint64_t kj = L.c_key[j];
                    pos += (kj != 0ull && (uint32_t)kj > (uint32_t)key) ? 1 : 0;  // low word: larger = smaller node
                }
                L.pq_list[pos] = (key & 0xFFFFFFFF00000000ull) | (uint32_t)c;
            }
            wave_sync();
            if (lane == 0) pdq178::sort_desc(L.pq_list, n_valid, L.pq_scr);
            wave_sync();
            for (int rank = lane; rank < Bn; rank += kWave) emit((int)(uint32_t)L.pq_list[rank], rank);
            wave_sync();
        }

        // ---- phase C: child rows of the new beam + renormalise by the top entry (:278-282) ----
        for (int item = lane; item < Bn * NL; item += kWave) {
            const int s = item / NL, l = item - s * NL;
            const int src = L.nb_src[s];
            const int c = src & 0x3FFFFFFF;
            const int i = c / N, k = c - i * N;
            int v;
            if (k == 0)
                v = b_child[i * NL + l];
            else if (src >> 30)
                v = -1;
            else
                v = load_i32_l2(&rows[(int64_t)L.b_node(nxt)[s] * NL + l]);
            L.b_child(nxt)[s * NL + l] = v;
        }
        const float top = *L.top;
        for (int s = lane; s < Bn; s += kWave) {
            L.b_lp(nxt)[s] = L.b_lp(nxt)[s] / top;
            L.b_gp(nxt)[s] = L.b_gp(nxt)[s] / top;
        }
        if (row_in_reg) {
            if (lane < N) L.row[lane] = next_row;
            next_row = (lane < N && t + 2 < T) ? load_post(post, (t + 2) * st_t + lane * st_n, dt) : 0.0f;
        } else if (!crf && t + 1 < T) {
            for (int j = lane; j < N; j += kWave) L.row[j] = load_post(post, (t + 1) * st_t + j * st_n, dt);
        }
        B = Bn;
        cur = nxt;
        wave_sync();
    }

    // ---- walk the best labelling leaf -> root (:285-300), writing it in sequence order ----
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop L1 lines older than our own stores
    if (lane == 0) {
        int node = L.b_node(cur)[0];
        const int n = L.b_depth(cur)[0];
        uint8_t *lab = p.out.labels + r * p.out.out_stride;
        uint32_t *pth = p.out.path ? p.out.path + r * p.out.out_stride : nullptr;
        for (int j = n - 1; j >= 0 && node >= 0; --j) {
            const int4 q = rec[node];
            lab[j] = (uint8_t)(q.z + 1);
            if (pth) pth[j] = (uint32_t)q.y;
            node = q.x;
        }
        p.out.out_len[r] = (uint32_t)n;
        p.out.status[r] = FCD_ST_OK;
        if (count_amb) {
            p.out.ambiguous[2 * r] = (uint32_t)n_amb;
            p.out.ambiguous[2 * r + 1] = (uint32_t)n_crit;
        }
    }
}

}  // namespace

bool beam_generic_replays_ties(int beam_size, int N, int tie_order) {
    return tie_order == FCD_TIE_PDQ178 && (int64_t)beam_size * N > 20;
}

size_t beam_generic_lds_bytes(int beam_size, int N, int tie_order) {
    return lds_words(beam_size, N, beam_generic_replays_ties(beam_size, N, tie_order)) * 4 + 16;
}

hipError_t launch_beam_generic(const BatchDesc &in, int64_t read_begin, int64_t n_reads,
                               const BeamArgs &a, const GenericArena &arena, const ResultDesc &out,
                               hipStream_t stream) {
    if (n_reads <= 0) return hipSuccess;
    GenericParams p{in, a, arena, out, read_begin};
    const size_t lds = beam_generic_lds_bytes(a.beam_size, in.N, a.tie_order);
    hipLaunchKernelGGL(beam_generic_kernel, dim3((unsigned)n_reads), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// this translation unit's copy of the replay's std-form word (pdq178.h), on the current device
FCD_PDQ178_DEFINE_STD_FORM_SETTER(beam_generic_set_pdq178_std_form)

}  // namespace fcd
