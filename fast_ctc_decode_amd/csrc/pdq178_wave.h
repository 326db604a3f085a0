// pdq178_wave.h -- pdq178.h's replay of Rust 1.78's sort_unstable_by, run by a whole wavefront on ONE list.
//
// Why: two beam entries with equal probabilities ("twins": prefixes that differ in one symbol of equal posterior)
// produce equal children step after step, so at wide beams a read that has met one tie hands the sort a tied list
// at EVERY later step (BASELINE config 3: 17 % of the reads, 21 of 8192 on more than half their steps).  Such a read's
// wavefront is the launch's straggler: it runs alone on its SIMD, one instruction per ~4 cycles, so what the replay
// costs is its INSTRUCTION COUNT.  Rounds 3 and 4 walked the partition tree level by level, all segments of up to
// two lists at once (a leader lane per segment, elements fetching their segment's parameters with ds_bpermute, five
// planes of positions): ~2900 instructions per level, 39 k cycles per tied step at beam 32.  At the tail there is
// one tied list per wavefront and one or two segments per level, so the generality bought nothing.  This routine
// takes ONE segment at a time off a small stack, and everything about the segment is wave-uniform -- scalar
// registers and scalar branches, no leaders, no parameter fetches:
//   * choose_pivot's samples are fetched by every lane from the same LDS addresses (a broadcast read) and the
//     sorting network with its swap count is evaluated redundantly; the scans of `partition` and the block split
//     of partition_in_blocks are bit scans and population counts over the votes of the classification;
//   * the O(len) parts stay data-parallel: one lane per element (per plane of 64 positions, and only the planes the
//     segment spans) classifies it against the pivot; its index among the misplaced elements of its block -- which
//     IS partition_in_blocks' offsets_l / offsets_r entry -- is a masked population count; the cyclic permutation of
//     the first `count` pairs is one scatter through two small position tables;
//   * what is inherently serial and short (the swaps that park the left-over misplaced elements, break_patterns,
//     the pivot swaps) is lane 0's; the rare heavy cases (heapsort after too many bad partitions,
//     partial_insertion_sort shifting on a long segment) are pdq178.h's serial routines, run by lane 0;
//   * segments of 20 elements or fewer end in an insertion sort, i.e. a STABLE sort: they are only marked (a bit per
//     boundary, in scalar registers) and every element of every such leaf ranks itself inside its leaf in one last
//     pass; a segment that lies wholly behind the first `keep` positions is dropped (segments never exchange
//     elements, so the kept prefix cannot tell).
// From 64 elements down a segment is handed, with its recursion state, to pdq178_reg.h: one element per lane, every
// partition a handful of cross-lane operations, nothing in memory.
// The stack lives in two vector registers (frame k in lane k): a push is a masked move, a pop two v_readlane.
// A list longer than kWaveMaxLen (its first partition could need more than one round of 128-element blocks) is
// sorted by pdq178::sort_desc on one lane instead.  tests/test_pdq178.py compares this routine with sort_desc --
// and both with the oracle's restatement -- element for element.
//
// The algorithm replayed is core::slice::sort of Rust 1.78.0 (library/core/src/slice/sort.rs; the Rust project,
// MIT OR Apache-2.0), restated -- see pdq178.h for what that claim rests on.
#pragma once

#include "pdq178.h"
#include "pdq178_reg.h"

namespace fcd {
namespace pdq178 {

constexpr int kWaveMaxLen = 2 * kBlock + 1;  // the pivot + at most 2 * BLOCK elements: partition_in_blocks is done in one round

template <int P>  // planes of 64 positions: a list lives in v[0 .. n), n <= 64 * P
struct alignas(16) WaveScratch {
    static constexpr int kPos = 64 * P;
    uint16_t pos_l[kPos];  // position of the k-th misplaced element of the left block
    uint16_t pos_r[kPos];  // ... of the right block, counted from the right end
};

#ifdef FCD_HIPEMU
#define FCD_LDS_AS
#else
#define FCD_LDS_AS __attribute__((address_space(3)))  // ds_* instructions instead of flat_* ones: a third of the latency
#endif

#if defined(FCD_WAVE_PROF) && !defined(FCD_HIPEMU)
// developer instrument (tools/dev/time_coop.py): shader cycles per phase, summed over the calls of lane 0's wavefronts
static __device__ unsigned long long g_wave_prof[16];  // (one per translation unit that asks for the stamps)
#define FCD_WAVE_STAMP(slot)                                                             \
    do {                                                                                 \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();                   \
        prof_acc__[slot] += (unsigned)(now__ - prof_last__);                             \
        prof_last__ = now__;                                                             \
    } while (0)
#define FCD_WAVE_STAMP_INIT                                              \
    unsigned prof_acc__[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      \
    unsigned long long prof_last__ = __builtin_amdgcn_s_memtime()
#define FCD_WAVE_COUNT_ROUND ++prof_acc__[10]
#define FCD_WAVE_COUNT_CALL                                                                   \
    do {                                                                                      \
        prof_acc__[11] = 1;                                                                   \
        if (lane == 0)                                                                        \
            for (int i__ = 0; i__ < 12; ++i__) atomicAdd(&g_wave_prof[i__], (unsigned long long)prof_acc__[i__]); \
    } while (0)
#else
#define FCD_WAVE_COUNT_CALL ((void)0)
#define FCD_WAVE_COUNT_ROUND ((void)0)
#define FCD_WAVE_STAMP(slot) ((void)0)
#define FCD_WAVE_STAMP_INIT ((void)0)
#endif

namespace wave_detail {

__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ uint64_t bits_below(int n) {  // n in [0, 64]
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}

// a bit per position of the P planes, in wave-uniform (scalar) registers
template <int P>
struct Bits {
    uint64_t m[P];
    // ones at positions < x
    __device__ __forceinline__ int ones_below(int x) const {
        int c = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int lo = x - 64 * j;
            if (lo > 0) c += __builtin_popcountll(m[j] & bits_below(lo));
        }
        return c;
    }
    // first position in [a, b) whose bit is `want`, or b
    __device__ __forceinline__ int first(int a, int b, bool want) const {
        int r = b;
#pragma unroll
        for (int j = P - 1; j >= 0; --j) {
            const int lo = a - 64 * j, hi = b - 64 * j;
            if (hi <= 0 || lo >= 64) continue;
            uint64_t z = want ? m[j] : ~m[j];
            if (lo > 0) z &= ~bits_below(lo);
            if (hi < 64) z &= bits_below(hi);
            if (z) r = 64 * j + __builtin_ctzll(z);
        }
        return r;
    }
    // last position in [a, b) whose bit is 1, or a - 1
    __device__ __forceinline__ int last_one(int a, int b) const {
        int r = a - 1;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int lo = a - 64 * j, hi = b - 64 * j;
            if (hi <= 0 || lo >= 64) continue;
            uint64_t o = m[j];
            if (lo > 0) o &= ~bits_below(lo);
            if (hi < 64) o &= bits_below(hi);
            if (o) r = 64 * j + 63 - __builtin_clzll(o);
        }
        return r;
    }
};

}  // namespace wave_detail

// Sorts the list v[0 .. n) (n <= 64 * P) into the order sort_unstable_by leaves it in -- as far as its first `keep`
// positions go.  Called by all 64 lanes of a wavefront with the same arguments; `lane` = the caller's lane; v and s
// are LDS, and whatever the lanes wrote to v must be visible (a wave-scope release / acquire) before the call.
template <int P>
__device__ __forceinline__ void wave_sort_inline(elem_t *v_generic, int n_in, int keep_in, WaveScratch<P> *s_generic, const int lane) {
    using namespace wave_detail;
    constexpr int kPos = 64 * P;
    static_assert(kPos <= kWaveMaxLen || sizeof(Scratch) <= sizeof(WaveScratch<P>), "the serial fall-back borrows the position tables as its scratch");
    typedef FCD_LDS_AS elem_t *vptr;
    const vptr v = (vptr)v_generic;
    FCD_LDS_AS WaveScratch<P> *const s = (FCD_LDS_AS WaveScratch<P> *)s_generic;
    auto key_at = [&](int idx) -> uint32_t {  // the sort key of v[idx]: the element's upper word
        return reinterpret_cast<FCD_LDS_AS const uint32_t *>(v + idx)[1];
    };
    const int n = uni(n_in), keep = uni(keep_in);
    if (n < 2) return;
    FCD_WAVE_STAMP_INIT;
    if (kPos > kWaveMaxLen && n > kWaveMaxLen) {
        if (lane == 0) sort_desc(v_generic, n, reinterpret_cast<Scratch *>(s_generic));
        sync();
        return;
    }
    // finished boundaries: a leaf / pivot / the list starts at every set bit
    Bits<P + 1> cut;
#pragma unroll
    for (int j = 0; j <= P; ++j) cut.m[j] = 0ull;
    auto set_cut = [&](int x) {  // (selects, not conditional stores: those the optimiser turns into cut.m[x >> 6], an array in scratch memory)
        const uint64_t b = 1ull << (x & 63);
#pragma unroll
        for (int j = 0; j <= P; ++j) cut.m[j] |= (x >> 6) == j ? b : 0ull;
    };
    set_cut(0);
    set_cut(n);

    // the segment in hand (wave-uniform) and the stack of the ones still to do: frame k in lane k of two registers
    int base = 0, len = n, pred = -1, limit = 0;
    bool wbal = true, wpar = true;
    for (uint32_t b = (uint32_t)n; b; b >>= 1) ++limit;  // usize::BITS - len.leading_zeros()
    int sp = 0;
    int fr_a = 0, fr_b = 0;  // base | len << 16, (pred + 1) | flags << 16
    bool busy = n > 20;
    FCD_WAVE_STAMP(0);
    while (busy) {
        FCD_WAVE_COUNT_ROUND;
        const vptr w = v + base;
        bool finished = false;  // this segment needs nothing more
        // what is left of the segment after this round, if anything (NORMAL: two children, EQUAL: one)
        int c_base[2] = {0, 0}, c_len[2] = {0, 0}, c_pred[2] = {-1, -1}, c_flag[2] = {0, 0};
        if (P == 1 || len <= 64) {
            // the segment fits the lanes of the wavefront: the rest of ITS quicksort -- every partition down to the leaves --
            // runs in registers (pdq178_reg.h) and comes back sorted as far as the kept prefix reaches
            uint32_t rk = 0, rt = 0;
            if (lane < len) {
                const elem_t e = w[lane];
                rk = (uint32_t)(e >> 32);
                rt = (uint32_t)e;
            }
            const uint32_t ppk = (uint32_t)uni((int)(pred >= 0 ? key_at(pred) : 0u));
            reg_sort(rk, rt, len, keep - base, pred >= 0, ppk, limit, wbal, wpar, v_generic + base, lane);
            if (lane < len) w[lane] = ((elem_t)rk << 32) | rt;
            sync();
            finished = true;
        } else if (limit == 0) {
            if (lane == 0) heapsort(w, len);
            sync();
            finished = true;
        } else {
            if (!wbal) {
                if (lane == 0) break_patterns(w, len);
                --limit;
                sync();
            }
            // ---- choose_pivot: the adjacent triples around len/4, len/2, 3 len/4 (from 50 elements: Tukey's ninther),
            // every lane reading the same addresses -- all nine loads in flight together (below 50 elements the outer
            // two of a triple are read and ignored: c - 1 >= base + 4 and c + 1 < base + len either way); the network
            // runs on (index, key) pairs ----
            const bool ninther = len >= 50;
            const int ia = len / 4, ib = ia * 2, ic = ia * 3;
            uint32_t e[9];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int c = base + (t == 0 ? ia : (t == 1 ? ib : ic));
                e[3 * t] = key_at(c - 1);
                e[3 * t + 1] = key_at(c);
                e[3 * t + 2] = key_at(c + 1);
            }
            int swaps = 0;
            auto srt2 = [&](int &a, uint32_t &ea, int &b, uint32_t &eb) {  // sort2: the smaller (in sort order) index first
                const bool sw = eb > ea;  // less(v[b], v[a])
                const int ta = a, tb = b;
                const uint32_t tea = ea, teb = eb;
                a = sw ? tb : ta;
                b = sw ? ta : tb;
                ea = sw ? teb : tea;
                eb = sw ? tea : teb;
                swaps += sw ? 1 : 0;
            };
            int ix[3];
            uint32_t ev[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int c = t == 0 ? ia : (t == 1 ? ib : ic);
                int lo = c - 1, mi = c, hi = c + 1;
                uint32_t elo = e[3 * t], emi = e[3 * t + 1], ehi = e[3 * t + 2];
                if (ninther) {  // sort_adjacent: the median of (c - 1, c, c + 1) replaces c
                    srt2(lo, elo, mi, emi);
                    srt2(mi, emi, hi, ehi);
                    srt2(lo, elo, mi, emi);
                }
                ix[t] = mi;
                ev[t] = emi;
            }
            srt2(ix[0], ev[0], ix[1], ev[1]);
            srt2(ix[1], ev[1], ix[2], ev[2]);
            srt2(ix[0], ev[0], ix[1], ev[1]);
            int pivot = uni(ix[1]);
            swaps = uni(swaps);
            bool likely_sorted = swaps == 0;
            FCD_WAVE_STAMP(1);
            if (swaps >= 12) {  // v.reverse()
                elem_t tmp[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int p = 64 * j + lane;
                    tmp[j] = 0;
                    if (base < 64 * (j + 1) && base + len > 64 * j && p >= base && p < base + len) tmp[j] = v[p];
                }
                sync();
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int p = 64 * j + lane;
                    if (base < 64 * (j + 1) && base + len > 64 * j && p >= base && p < base + len) v[base + (len - 1 - (p - base))] = tmp[j];
                }
                sync();
                pivot = len - 1 - pivot;
                likely_sorted = true;
            }
            if (wbal && wpar && likely_sorted) {
                // partial_insertion_sort: is the segment sorted already?  (every element against its predecessor)
                Bits<P> desc;
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int p = 64 * j + lane;
                    bool d = false;
                    if (base + 1 < 64 * (j + 1) && base + len > 64 * j && p > base && p < base + len) d = key_at(p) > key_at(p - 1);
                    desc.m[j] = __builtin_amdgcn_ballot_w64(d);
                }
                const int first_desc = desc.first(base + 1, base + len, true);
                if (first_desc == base + len) {
                    finished = true;
                } else if (len >= 50) {  // it goes on to shift elements about: the serial routine (rare)
                    int done = 0;
                    if (lane == 0) done = partial_insertion_sort(w, len) ? 1 : 0;
                    sync();
                    finished = __builtin_amdgcn_readlane(done, 0) != 0;
                }
            }
            if (!finished) {
                // ---- ONE round of loads: every element of the segment (a lane per element and plane), the pivot, the
                // first element (it changes places with the pivot: swap(0, pivot)) and the predecessor.  The swap
                // happens in registers: the lane of the pivot's position takes the first element, position `base` is
                // not classified, and what LDS holds at the two positions is put right by the scatter below. ----
                const int wb = base + 1, we = base + len;
                const int ppos = base + pivot;
                elem_t val[P];
                bool in[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int p = 64 * j + lane;
                    val[j] = 0;
                    in[j] = false;
                    if (wb < 64 * (j + 1) && we > 64 * j) {  // (wave-uniform: the segment reaches this plane)
                        in[j] = p >= wb && p < we;
                        if (in[j]) val[j] = v[p];
                    }
                }
                const elem_t pval = w[pivot], first = w[0];
                const uint32_t pe = pred >= 0 ? key_at(pred) : 0u;
                const uint32_t pk = (uint32_t)uni((int)(uint32_t)(pval >> 32));
                // which partition: the pivot equals the predecessor (nothing in the segment is "less" than it) -> partition_equal
                const bool equal = pred >= 0 && !((uint32_t)uni((int)pe) > pk);  // !less(v[pred], v[pivot])
                FCD_WAVE_STAMP(2);
                // ---- one lane per element: which side of the pivot does it belong to? ----
                bool bit[P];
                Bits<P> mk;
                int ones_upto[P];  // ones in the planes below plane j
                {
                    int run = 0;
#pragma unroll
                    for (int j = 0; j < P; ++j) {
                        const int p = 64 * j + lane;
                        if (in[j] && p == ppos) val[j] = first;
                        const uint32_t k = (uint32_t)(val[j] >> 32);
                        // NORMAL: less(e, pivot) -- the element belongs left.  EQUAL: less(pivot, e) -- it belongs right.
                        bit[j] = in[j] && (equal ? pk > k : k > pk);
                        mk.m[j] = __builtin_amdgcn_ballot_w64(bit[j]);
                        ones_upto[j] = run;
                        run += __builtin_popcountll(mk.m[j]);
                    }
                }
                FCD_WAVE_STAMP(3);
                // ---- the scans of `partition`, the block split of partition_in_blocks, the counts (all scalar) ----
                int a0, a1, a2, p0, p2, count, cL = 0, cR = 0;
                if (!equal) {
                    a0 = mk.first(wb, we, false);                      // while l < r && is_less(v[l], pivot)
                    const int last1 = mk.last_one(a0, we);
                    a2 = last1 + 1 > a0 ? last1 + 1 : a0;              // while l < r && !is_less(v[r - 1], pivot)
                    const int rem = a2 - a0;                           // <= 2 * BLOCK: one round, is_done at once
                    a1 = a0 + rem / 2;                                 // block_l = rem / 2, block_r = rem - block_l
                    p0 = mk.ones_below(a0);
                    const int p1 = mk.ones_below(a1);
                    p2 = mk.ones_below(a2);
                    cL = (a1 - a0) - (p1 - p0);                        // left block: elements that are NOT less than the pivot
                    cR = p2 - p1;                                      // right block: elements that are
                    count = cL < cR ? cL : cR;
                } else {
                    p0 = 0;                                            // (no bits below wb)
                    p2 = mk.ones_below(we);
                    const int nE = (we - wb) - p2;                     // elements equal to the pivot: they end up on the left
                    a0 = wb;
                    a1 = wb + nE;
                    a2 = we;
                    count = mk.ones_below(a1);                         // greater ones inside the left zone == equal ones outside it
                }
                // ---- every misplaced element's index among the misplaced ones of its side -> position tables ----
                int role[P], kk[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    role[j] = 0;
                    kk[j] = 0;
                    if (!(wb < 64 * (j + 1) && we > 64 * j) || !in[j]) continue;
                    const int p = 64 * j + lane;
                    const int ones_before = ones_upto[j] + reg_detail::ones_below_lane(mk.m[j], lane);
                    if (!equal) {
                        if (p >= a0 && p < a1 && !bit[j]) {         // offsets_l, in tracing order (left to right)
                            role[j] = 1;
                            kk[j] = (p - a0) - (ones_before - p0);
                            s->pos_l[kk[j]] = (uint16_t)p;
                        } else if (p >= a1 && p < a2 && bit[j]) {   // offsets_r, in tracing order (right to left)
                            role[j] = 2;
                            kk[j] = p2 - ones_before - 1;
                            s->pos_r[kk[j]] = (uint16_t)p;
                        }
                    } else {
                        if (p < a1 && bit[j]) {                     // a greater element inside the left zone, from the left
                            role[j] = 1;
                            kk[j] = ones_before;
                            s->pos_l[kk[j]] = (uint16_t)p;
                        } else if (p >= a1 && !bit[j]) {            // an equal element outside it, from the right
                            role[j] = 2;
                            kk[j] = (a2 - p - 1) - (p2 - ones_before);
                            s->pos_r[kk[j]] = (uint16_t)p;
                        }
                    }
                }
                sync();
                FCD_WAVE_STAMP(4);
                // ---- where every element ends up, in ONE scatter: every mover still holds its value in a register.
                // NORMAL, as partition_in_blocks and `partition` do it:
                //  1. the cyclic permutation L0 <- R0 <- L1 <- R1 ... <- R(count-1) <- L0 of the first `count` misplaced pairs;
                //  2. the left-over misplaced elements of the longer side are parked against the block boundary by
                //     swaps taken from the far end: left-over hole j of the left block ends at a1 - cL + j (of the right
                //     block: a1 + cR - 1 - j), and the well-placed element the swap met at that spot goes to the hole --
                //     which may itself be a spot a later swap visits, and so on: it follows the chain of holes until one
                //     lies outside the parking zone (each hop moves strictly away from the boundary);
                //  3. swap(0, mid): whoever ends at the last position of the left part goes to `base`, the pivot there.
                // EQUAL: the k-th greater element from the left swaps with the k-th equal one from the right; the pivot
                // stays in front. ----
                const int m_left = equal ? 0 : cL - count, m_right = equal ? 0 : cR - count;  // (one of them is 0)
                const int pm = a1 - m_left + m_right - 1;  // NORMAL: where the pivot belongs
                const int zlo = m_left > 0 ? a1 - m_left : a1, zhi = m_left > 0 ? a1 : a1 + m_right;  // the parking zone
                const int org_l = a1 - cL, org_r = a1 + cR - 1;  // left-over hole j parks at org_l + j / org_r - j
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    if (!in[j]) continue;
                    const int p = 64 * j + lane;
                    int z = p;             // where the element sits after step 1
                    bool parked = false;   // a left-over misplaced element: its spot is known outright
                    if (role[j] != 0) {
                        if (kk[j] < count) {
                            if (!equal) z = role[j] == 1 ? s->pos_r[kk[j] == 0 ? count - 1 : kk[j] - 1] : s->pos_l[kk[j]];
                            else z = role[j] == 1 ? s->pos_r[kk[j]] : s->pos_l[kk[j]];
                        } else {
                            z = role[j] == 1 ? org_l + kk[j] : org_r - kk[j];
                            parked = true;
                        }
                    }
                    if (!parked && z >= zlo && z < zhi) {  // met by a parking swap: on to that swap's hole, and again
                        if (m_left > 0) {
                            do z = s->pos_l[z - org_l]; while (z >= zlo);
                        } else {
                            do z = s->pos_r[org_r - z]; while (z < zhi);
                        }
                    }
                    if (!equal && z == pm) z = base;
                    // (the pivot's old position still holds the pivot in LDS: whoever sits there writes even if it stays)
                    if (z != p || p == ppos) v[z] = val[j];
                }
                if (lane == 0) {
                    if (equal || pm == base) w[0] = pval;
                    else v[pm] = pval;
                }
                sync();
                FCD_WAVE_STAMP(5);
                if (!equal) {
                    const int mid = pm - base;
                    const int smaller = mid < len - mid ? mid : len - mid;
                    const bool nb = smaller >= len / 8, np = a0 >= a2;
                    const int nl = mid, nr = len - mid - 1;
                    set_cut(pm);
                    set_cut(pm + 1);
                    // recurse into the shorter side (a fresh call: balanced, partitioned), carry on with the longer one
                    const int fresh = limit | 256 | 512, cont = limit | (nb ? 256 : 0) | (np ? 512 : 0);
                    c_base[0] = base;
                    c_len[0] = nl;
                    c_pred[0] = pred;
                    c_flag[0] = nl < nr ? fresh : cont;
                    c_base[1] = pm + 1;
                    c_len[1] = nr;
                    c_pred[1] = pm;
                    c_flag[1] = nl < nr ? cont : fresh;
                } else {
                    const int mid = (a1 - wb) + 1;  // the elements equal to the pivot (and the pivot) are done
                    set_cut(base + mid);
                    c_base[1] = base + mid;
                    c_len[1] = len - mid;
                    c_pred[1] = pred;
                    c_flag[1] = limit | (wbal ? 256 : 0) | (wpar ? 512 : 0);
                }
                FCD_WAVE_STAMP(6);
            }
        }
        // ---- what next: a child that still needs partitioning (longer than a leaf, not wholly behind the kept
        // prefix), else the top of the stack ----
        const bool need0 = !finished && c_len[0] > 20 && c_base[0] < keep;
        const bool need1 = !finished && c_len[1] > 20 && c_base[1] < keep;
        if (need0 && need1) {  // push child 1
            if (lane == sp) {
                fr_a = c_base[1] | (c_len[1] << 16);
                fr_b = (c_pred[1] + 1) | (c_flag[1] << 16);
            }
            ++sp;
        }
        if (need0 || need1) {
            const int c = need0 ? 0 : 1;
            base = c_base[c];
            len = c_len[c];
            pred = c_pred[c];
            limit = c_flag[c] & 255;
            wbal = (c_flag[c] & 256) != 0;
            wpar = (c_flag[c] & 512) != 0;
        } else if (sp > 0) {
            --sp;
            const int fa = __builtin_amdgcn_readlane(fr_a, sp), fb = __builtin_amdgcn_readlane(fr_b, sp);
            base = fa & 0xFFFF;
            len = fa >> 16;
            pred = (fb & 0xFFFF) - 1;
            limit = (fb >> 16) & 255;
            wbal = ((fb >> 16) & 256) != 0;
            wpar = ((fb >> 16) & 512) != 0;
        } else {
            busy = false;
        }
        FCD_WAVE_STAMP(7);
    }
    FCD_WAVE_STAMP(8);

    // ---- leaves: what is left between two boundaries and holds 20 elements or fewer ends in pdqsort's insertion
    // sort -- a stable sort: every element ranks itself inside its leaf.  (Longer stretches are finished regions:
    // runs of equal elements, slices partial_insertion_sort or heapsort completed.) ----
    elem_t lval[P];
    int dest[P];
    const int reach = keep < kPos ? keep + 20 : kPos;  // (20 positions of slack cover every leaf that reaches into the kept prefix)
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int p = 64 * j + lane;
        dest[j] = -1;
        lval[j] = 0;
        if (64 * j >= n || 64 * j >= reach) continue;  // (wave-uniform: nothing of this plane matters)
        // nearest boundary at or below p and nearest one above it, as far as 63 positions away: two 64-bit windows of the
        // boundary bits around p (the list starts and ends with a boundary; a leaf spans at most 20 positions)
        const uint64_t below_w = (cut.m[j] << (63 - lane)) | (j > 0 ? ((cut.m[j > 0 ? j - 1 : 0] >> 1) >> lane) : 0ull);  // bit 63 = position p
        const uint64_t above_w = ((cut.m[j] >> 1) >> lane) | (cut.m[j + 1] << (63 - lane));                          // bit 0 = position p + 1
        const int lo = p - (below_w ? __builtin_clzll(below_w) : 64);
        const int hi = p + 1 + (above_w ? __builtin_ctzll(above_w) : 64);
        const int ln = hi - lo;
        const bool mine = p < n && ln >= 2 && ln <= 20 && lo < keep;  // (whole leaves: in or out)
        if (__builtin_amdgcn_ballot_w64(mine) == 0ull) continue;      // (wave-uniform)
        if (mine) lval[j] = v[p];
        const uint32_t key = (uint32_t)(lval[j] >> 32);
        // the leaf's keys, ten loads in flight together
        int rank = 0;
#pragma unroll
        for (int u0 = 0; u0 < 20; u0 += 10) {
            uint32_t kq[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) kq[u] = mine ? key_at(lo + u0 + u < n ? lo + u0 + u : n - 1) : 0u;
#pragma unroll
            for (int u = 0; u < 10; ++u)
                rank += (lo + u0 + u < hi && (kq[u] > key || (kq[u] == key && lo + u0 + u < p))) ? 1 : 0;
        }
        if (mine) dest[j] = lo + rank;
    }
    sync();
#pragma unroll
    for (int j = 0; j < P; ++j)
        if (dest[j] >= 0) v[dest[j]] = lval[j];
    sync();
    FCD_WAVE_STAMP(9);
    FCD_WAVE_COUNT_CALL;
}

// The same as a call (kernels whose register budget cannot hold the routine next to their own state).
template <int P>
__device__ __attribute__((noinline)) void wave_sort(elem_t *v, int n, int keep, WaveScratch<P> *s, int lane) {
    wave_sort_inline<P>(v, n, keep, s, lane);
}

}  // namespace pdq178
}  // namespace fcd
