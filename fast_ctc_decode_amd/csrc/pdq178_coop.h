// pdq178_coop.h -- pdq178.h's replay of Rust 1.78's sort_unstable_by, run by a whole wavefront.
//
// Why: two beam entries with equal probabilities ("twins": prefixes that differ in one symbol of equal
// posterior) produce equal children step after step, so at wide beams a read that has met one tie hands the sort
// a tied list at EVERY later step (BASELINE config 3: 17 % of the reads).  One lane replaying the quicksort on
// ~130 elements takes ~20x a normal step, and a launch is as slow as its slowest wavefront.  The permutation
// pdqsort produces is a deterministic function of the list, but nothing forces its replay to be serial:
//   * the partition tree is walked level by level; every segment of a level is independent of the others, so all
//     of them -- of up to two lists (the two reads of a wavefront) -- are processed in the same pass;
//   * the O(len) parts of a partition are data-parallel: one lane per element classifies it against its
//     segment's pivot, wave-wide ballots give every element its index among the misplaced elements of its block
//     (partition_in_blocks' offsets_l / offsets_r ARE those indices), and the cyclic permutation of the first
//     `count` pairs is one scatter through two small position tables;
//   * what is inherently serial and short -- choose_pivot's sorting network, the swaps that park the left-over
//     misplaced elements, break_patterns, the pivot swaps -- is done by one "leader" lane per segment, all leaders
//     at once; the rare heavy cases (heapsort after too many bad partitions, partial_insertion_sort on a long
//     segment) stay pdq178.h's serial routines, called by the leader;
//   * segments of 20 elements or fewer end in an insertion sort, i.e. a STABLE sort: every element of every such
//     leaf ranks itself inside its leaf in one last pass.
// A list longer than kCoopMaxLen (its first partition could need more than one round of 128-element blocks) is
// sorted by pdq178::sort_desc on one lane instead.  tests/test_pdq178.py compares this routine with sort_desc --
// and both with the oracle's restatement -- element for element.
#pragma once

#include "pdq178.h"

namespace fcd {
namespace pdq178 {

constexpr int kCoopMaxLen = 2 * kBlock + 1;  // the pivot + at most 2 * BLOCK elements: partition_in_blocks is done in one round

template <int MAXP>  // planes of 64 positions: the lists live in v[0 .. 64 * MAXP)
struct alignas(16) CoopScratch {
    static constexpr int kPos = 64 * MAXP;
    static constexpr int kMaxSeg = kPos / 21 + 2;  // segments longer than 20 elements that can coexist
    uint16_t pos_a[kPos];        // position of the k-th misplaced element of a segment's left side (at [wb + k])
    uint16_t pos_b[kPos];        // ... of its right side, counted from the right end
    uint8_t cut[kPos + 8];       // 1 = a finished boundary: a leaf / pivot / list starts here
    int16_t seg[kMaxSeg][4];     // {base, len, pred, limit | was_balanced << 8 | was_partitioned << 9}: lane i leads segment i
};  // (what a leader decides in a round stays in its registers; elements fetch it with ds_bpermute / v_readlane)

enum { kActDone = 0, kActNormal = 1, kActEqual = 2 };

#ifdef FCD_HIPEMU
#define FCD_LDS_AS
#else
#define FCD_LDS_AS __attribute__((address_space(3)))  // ds_* instructions instead of flat_* ones: a third of the latency
#endif

#if defined(FCD_COOP_PROF) && !defined(FCD_HIPEMU)
// developer instrument (tools/dev/time_coop.py): shader cycles per phase, summed over the calls of lane 0's wavefronts
__device__ unsigned long long g_coop_prof[16];
#define FCD_COOP_STAMP(slot)                                                             \
    do {                                                                                 \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();                   \
        prof_acc__[slot] += (unsigned)(now__ - prof_last__);                             \
        prof_last__ = now__;                                                             \
    } while (0)
#define FCD_COOP_STAMP_INIT                                         \
    unsigned prof_acc__[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    \
    unsigned long long prof_last__ = __builtin_amdgcn_s_memtime()
#define FCD_COOP_COUNT_ROUND ++prof_acc__[9]
#define FCD_COOP_COUNT_CALL                                                                   \
    do {                                                                                      \
        prof_acc__[10] = 1;                                                                   \
        for (int i__ = 0; i__ < 11; ++i__) atomicAdd(&g_coop_prof[i__], (unsigned long long)prof_acc__[i__]); \
    } while (0)
#else
#define FCD_COOP_COUNT_CALL ((void)0)
#define FCD_COOP_COUNT_ROUND ((void)0)
#define FCD_COOP_STAMP(slot) ((void)0)
#define FCD_COOP_STAMP_INIT ((void)0)
#endif

namespace coop_detail {

__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t bits_below(int n) {  // n in [0, 64]
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}

template <int MAXP>
struct Masks {
    uint64_t m[MAXP];
    // ones at positions < x
    __device__ __forceinline__ int prefix1(int x) const {
        int c = 0;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const int lo = x - 64 * j;
            if (lo > 0) c += __builtin_popcountll(m[j] & bits_below(lo));
        }
        return c;
    }
    // first position in [a, b) whose bit is 0, or b
    __device__ __forceinline__ int first_zero(int a, int b) const {
        int r = b;
#pragma unroll
        for (int j = MAXP - 1; j >= 0; --j) {
            const int lo = a - 64 * j, hi = b - 64 * j;
            if (hi <= 0 || lo >= 64) continue;
            uint64_t z = ~m[j];
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
        for (int j = 0; j < MAXP; ++j) {
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

}  // namespace coop_detail

// Sorts up to two lists v[start0 .. start0 + len0) and v[start1 .. start1 + len1) (len 0 = absent; the ranges must not
// overlap, start0 < start1, both inside [0, 64 * MAXP)) into the order sort_unstable_by leaves them in -- as far as the
// first `keep` positions of each list go: a segment that lies wholly behind them is dropped (segments never exchange
// elements, so the kept prefix cannot tell).  Called by all 64 lanes of a wavefront with the same arguments; `lane` =
// the caller's lane; v and s are LDS.
template <int MAXP>
__device__ __forceinline__ void coop_sort_inline(elem_t *v_generic, int start0, int len0, int start1, int len1, int keep,
                                                 CoopScratch<MAXP> *s_generic, int lane) {
    using namespace coop_detail;
    constexpr int kPos = 64 * MAXP;
    static_assert(sizeof(Scratch) * 2 <= sizeof(uint16_t) * 2 * kPos || kPos < kCoopMaxLen,
                  "the serial fall-back borrows the position tables as its scratch");
    typedef FCD_LDS_AS elem_t *vptr;
    const vptr v = (vptr)v_generic;
    FCD_LDS_AS CoopScratch<MAXP> *const s = (FCD_LDS_AS CoopScratch<MAXP> *)s_generic;

    FCD_COOP_STAMP_INIT;
    // ---- boundaries known from the start; lists that are leaves or too long for one-round partitions ----
    {
        FCD_LDS_AS uint64_t *c8 = reinterpret_cast<FCD_LDS_AS uint64_t *>(s->cut);  // (the struct is 16-byte aligned, kPos even)
        for (int p = lane; p < (kPos + 8) / 8; p += 64) c8[p] = 0ull;
    }
    sync();
    int nseg = 0;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int st = k ? start1 : start0, ln = k ? len1 : len0;
            if (ln <= 0) continue;
            s->cut[st] = 1;
            s->cut[st + ln] = 1;
            if (ln > 20 && ln <= kCoopMaxLen) {
                FCD_LDS_AS int16_t *f = s->seg[nseg++];
                int limit = 0;  // usize::BITS - len.leading_zeros()
                for (uint32_t m = (uint32_t)ln; m; m >>= 1) ++limit;
                f[0] = (int16_t)st;
                f[1] = (int16_t)ln;
                f[2] = -1;
                f[3] = (int16_t)(limit | 256 | 512);
            }
        }
    }
    nseg = __builtin_amdgcn_readfirstlane(nseg);
    if (kPos >= kCoopMaxLen && (len0 > kCoopMaxLen || len1 > kCoopMaxLen)) {
        Scratch *ser = reinterpret_cast<Scratch *>(s_generic->pos_a);
        if (lane == 0 && len0 > kCoopMaxLen) sort_desc(v_generic + start0, len0, ser);
        if (lane == 32 && len1 > kCoopMaxLen) sort_desc(v_generic + start1, len1, ser + 1);
    }
    sync();

    FCD_COOP_STAMP(0);
    while (nseg > 0) {
        FCD_COOP_STAMP(7);
        // ---- A: every segment's leader picks the pivot (and does what only ever touches a few elements) ----
        int base = 0, len = 0, pred = -1, limit = 0;
        bool wbal = true, wpar = true;
        int act = kActDone;
        uint32_t pk = 0;
        elem_t piv = 0;
        if (lane < nseg) {
            FCD_LDS_AS const int16_t *f = s->seg[lane];
            base = f[0];
            len = f[1];
            pred = f[2];
            const int fl = f[3];
            limit = fl & 255;
            wbal = (fl & 256) != 0;
            wpar = (fl & 512) != 0;
            const vptr w = v + base;
            if (limit == 0) {
                heapsort(w, len);
            } else {
                if (!wbal) {
                    break_patterns(w, len);
                    --limit;
                }
                // choose_pivot with its samples loaded up front: the adjacent triples around len/4, len/2, 3 len/4 (from
                // 50 elements: Tukey's ninther), the predecessor, the first element -- one LDS round trip
                const bool ninther = len >= 50;
                const int ia = len / 4, ib = len / 4 * 2, ic = len / 4 * 3;
                elem_t e[9];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int c = t == 0 ? ia : (t == 1 ? ib : ic);
                    e[3 * t + 1] = w[c];
                    e[3 * t] = ninther ? w[c - 1] : 0;
                    e[3 * t + 2] = ninther ? w[c + 1] : 0;
                }
                const elem_t first = w[0];
                const elem_t pe = pred >= 0 ? v[pred] : 0;
                int swaps = 0;
                int ix[3];
                elem_t ev[3];
                auto srt2 = [&](int &a, elem_t &ea, int &b, elem_t &eb) {  // sort2: the smaller (in sort order) index first
                    if (less(eb, ea)) {
                        const int ti = a;
                        a = b;
                        b = ti;
                        const elem_t te = ea;
                        ea = eb;
                        eb = te;
                        ++swaps;
                    }
                };
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int c = t == 0 ? ia : (t == 1 ? ib : ic);
                    int lo = c - 1, mi = c, hi = c + 1;
                    elem_t elo = e[3 * t], emi = e[3 * t + 1], ehi = e[3 * t + 2];
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
                int pivot = ix[1];
                elem_t pval = ev[1];
                bool likely_sorted = swaps == 0;
                bool moved = false;  // the samples above no longer describe the segment
                if (swaps >= 12) {
                    for (int i = 0; i < len / 2; ++i) swp(w, i, len - 1 - i);  // v.reverse()
                    pivot = len - 1 - pivot;
                    likely_sorted = true;
                    moved = true;
                }
                bool done = false;
                if (wbal && wpar && likely_sorted) {
                    done = partial_insertion_sort(w, len);
                    moved = moved || len >= 50;  // (a shorter segment is only inspected)
                }
                if (!done) {
                    if (moved) pval = w[pivot];
                    act = (pred >= 0 && !less(pe, pval)) ? kActEqual : kActNormal;
                    const elem_t f0 = moved ? w[0] : first;   // swap(0, pivot)
                    w[pivot] = f0;
                    w[0] = pval;
                    piv = pval;
                    pk = (uint32_t)(pval >> 32);
                }
            }
        }
        sync();
        FCD_COOP_STAMP(1);

        // ---- B: one lane per element: which side of its segment's pivot does it belong to? ----
        elem_t val[MAXP];
        int myseg[MAXP];
        bool bit[MAXP];
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            val[j] = v[64 * j + lane];
            myseg[j] = -1;
            bit[j] = false;
        }
        for (int sg = 0; sg < nseg; ++sg) {
            const int a = __builtin_amdgcn_readlane(act, sg);
            if (a == kActDone) continue;
            const int wb = __builtin_amdgcn_readlane(base, sg) + 1, we = wb - 1 + __builtin_amdgcn_readlane(len, sg);
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)pk, sg);
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                if (wb >= 64 * (j + 1) || we <= 64 * j) continue;  // (wave-uniform: the segment does not reach this plane)
                const int p = 64 * j + lane;
                if (p >= wb && p < we) {
                    myseg[j] = sg;
                    const uint32_t k = (uint32_t)(val[j] >> 32);
                    // NORMAL: less(e, pivot) -- the element belongs left.  EQUAL: less(pivot, e) -- it belongs right.
                    bit[j] = a == kActNormal ? k > k0 : k0 > k;
                }
            }
        }
        Masks<MAXP> mk;
        int ones_upto[MAXP];     // ones in the planes below plane j
        bool plane_busy[MAXP];   // some element of plane j belongs to a segment of this round
        {
            int run = 0;
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                mk.m[j] = __builtin_amdgcn_ballot_w64(bit[j]);
                plane_busy[j] = __builtin_amdgcn_ballot_w64(myseg[j] >= 0) != 0ull;
                ones_upto[j] = run;
                run += __builtin_popcountll(mk.m[j]);
            }
        }

        FCD_COOP_STAMP(2);
        // ---- C: leaders: the scans of `partition`, the block split of partition_in_blocks, the counts ----
        int a0 = 0, a1 = 0, a2 = 0, p0 = 0, p2 = 0, count = 0, cL = 0, cR = 0;
        auto ones_below = [&](int x) -> int {  // ones at positions < x: the plane's running count + one masked popcount
            int c = 0;
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int lo = x - 64 * j;
                if (lo > 0 && lo <= 64) c = ones_upto[j] + __builtin_popcountll(mk.m[j] & bits_below(lo));
            }
            return c;  // (x = 0: no plane qualifies, 0)
        };
        if (lane < nseg && act != kActDone) {
            const int wb = base + 1, we = base + len;
            if (act == kActNormal) {
                a0 = mk.first_zero(wb, we);                        // while l < r && is_less(v[l], pivot)
                const int last1 = mk.last_one(a0, we);
                a2 = last1 + 1 > a0 ? last1 + 1 : a0;              // while l < r && !is_less(v[r - 1], pivot)
                const int rem = a2 - a0;                           // <= 2 * BLOCK: one round, is_done at once
                a1 = a0 + rem / 2;                                 // block_l = rem / 2, block_r = rem - block_l
                p0 = ones_below(a0);
                const int p1 = ones_below(a1);
                p2 = ones_below(a2);
                cL = (a1 - a0) - (p1 - p0);                        // left block: elements that are NOT less than the pivot
                cR = p2 - p1;                                      // right block: elements that are
                count = cL < cR ? cL : cR;
            } else {
                p0 = ones_below(wb);
                p2 = ones_below(we);
                const int nE = (we - wb) - (p2 - p0);              // elements equal to the pivot: they end up on the left
                a0 = wb;
                a1 = wb + nE;
                a2 = we;
                count = ones_below(a1) - p0;                       // greater ones inside the left zone == equal ones outside it
            }
        }
        // what an element needs to know about its segment, two 16-bit fields to a word, fetched from the leader's lane
        const int w_a01 = a0 | (a1 << 16), w_a2b = a2 | ((base + 1) << 16), w_p02 = p0 | (p2 << 16), w_cnt = count | (act << 16);

        FCD_COOP_STAMP(3);
        // ---- D: every misplaced element's index among the misplaced ones of its side -> position tables ----
        int role[MAXP], kk[MAXP], s_cnt[MAXP], s_wb[MAXP];
        bool s_normal[MAXP];
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            role[j] = 0;
            kk[j] = 0;
            s_cnt[j] = 0;
            s_wb[j] = 0;
            s_normal[j] = false;
            if (!plane_busy[j]) continue;  // (wave-uniform)
            const int src = (myseg[j] < 0 ? 0 : myseg[j]) << 2;
            const int g_a01 = __builtin_amdgcn_ds_bpermute(src, w_a01), g_a2b = __builtin_amdgcn_ds_bpermute(src, w_a2b);
            const int g_p02 = __builtin_amdgcn_ds_bpermute(src, w_p02), g_cnt = __builtin_amdgcn_ds_bpermute(src, w_cnt);
            role[j] = 0;
            kk[j] = 0;
            s_cnt[j] = g_cnt & 0xFFFF;
            s_wb[j] = g_a2b >> 16;
            s_normal[j] = (g_cnt >> 16) == kActNormal;
            if (myseg[j] < 0) continue;
            const int p = 64 * j + lane;
            const int e0 = g_a01 & 0xFFFF, e1 = g_a01 >> 16, e2 = g_a2b & 0xFFFF, q0 = g_p02 & 0xFFFF, q2 = g_p02 >> 16;
            const int ones_before = ones_upto[j] + __builtin_popcountll(mk.m[j] & bits_below(lane));
            if (s_normal[j]) {
                if (p >= e0 && p < e1 && !bit[j]) {         // offsets_l, in tracing order (left to right)
                    role[j] = 1;
                    kk[j] = (p - e0) - (ones_before - q0);
                    s->pos_a[s_wb[j] + kk[j]] = (uint16_t)p;
                } else if (p >= e1 && p < e2 && bit[j]) {   // offsets_r, in tracing order (right to left)
                    role[j] = 2;
                    kk[j] = q2 - ones_before - 1;
                    s->pos_b[s_wb[j] + kk[j]] = (uint16_t)p;
                }
            } else {
                if (p < e1 && bit[j]) {                     // a greater element inside the left zone, from the left
                    role[j] = 1;
                    kk[j] = ones_before - q0;
                    s->pos_a[s_wb[j] + kk[j]] = (uint16_t)p;
                } else if (p >= e1 && !bit[j]) {            // an equal element outside it, from the right
                    role[j] = 2;
                    kk[j] = (e2 - p - 1) - (q2 - ones_before);
                    s->pos_b[s_wb[j] + kk[j]] = (uint16_t)p;
                }
            }
        }
        sync();

        FCD_COOP_STAMP(4);
        // ---- E: the moves.  NORMAL: the cyclic permutation L0 <- R0 <- L1 <- R1 ... <- R(count-1) <- L0 of the first
        // `count` misplaced pairs; EQUAL: the k-th greater element from the left swaps with the k-th equal one from
        // the right.  Every mover still holds its own value in a register. ----
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (role[j] == 0 || kk[j] >= s_cnt[j]) continue;
            int dest;
            if (s_normal[j])
                dest = role[j] == 1 ? s->pos_b[s_wb[j] + (kk[j] == 0 ? s_cnt[j] - 1 : kk[j] - 1)] : s->pos_a[s_wb[j] + kk[j]];
            else
                dest = role[j] == 1 ? s->pos_b[s_wb[j] + kk[j]] : s->pos_a[s_wb[j] + kk[j]];
            v[dest] = val[j];
        }
        sync();

        FCD_COOP_STAMP(5);
        // ---- F: leaders: park the left-over misplaced elements, put the pivot in place, queue the children ----
        int c_base[2] = {0, 0}, c_len[2] = {0, 0}, c_pred[2] = {-1, -1}, c_flag[2] = {0, 0};
        if (lane < nseg && act != kActDone) {
            const int wb = base + 1;
            if (act == kActNormal) {
                int bound = a1;
                if (cL > cR) {          // while start_l < end_l { end_l -= 1; swap(l + *end_l, r - 1); r -= 1 }
                    for (int j = cL - 1; j >= count; --j) {
                        --bound;
                        const int hole = s->pos_a[wb + j];
                        const elem_t t = v[hole], u = v[bound];
                        v[hole] = u;
                        v[bound] = t;
                    }
                } else if (cR > cL) {   // while start_r < end_r { end_r -= 1; swap(l, r - *end_r - 1); l += 1 }
                    for (int j = cR - 1; j >= count; --j) {
                        const int hole = s->pos_b[wb + j];
                        const elem_t t = v[hole], u = v[bound];
                        v[hole] = u;
                        v[bound] = t;
                        ++bound;
                    }
                }
                const int pm = bound - 1;  // where the pivot belongs: swap(0, mid)
                const elem_t t = v[pm];
                v[base] = t;
                v[pm] = piv;
                const int mid = pm - base;
                const int smaller = mid < len - mid ? mid : len - mid;
                const bool nb = smaller >= len / 8, np = a0 >= a2;
                const int nl = mid, nr = len - mid - 1;
                s->cut[pm] = 1;
                s->cut[pm + 1] = 1;
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
                s->cut[base + mid] = 1;
                c_base[1] = base + mid;
                c_len[1] = len - mid;
                c_pred[1] = pred;
                c_flag[1] = limit | (wbal ? 256 : 0) | (wpar ? 512 : 0);
            }
            // a segment wholly behind the kept prefix of its list is nobody's business
            const int horizon = ((len1 > 0 && base >= start1) ? start1 : start0) + keep;
            if (c_base[0] >= horizon) c_len[0] = 0;
            if (c_base[1] >= horizon) c_len[1] = 0;
        }
        const uint64_t q0 = __builtin_amdgcn_ballot_w64(c_len[0] > 20), q1 = __builtin_amdgcn_ballot_w64(c_len[1] > 20);
        const uint64_t below = bits_below(lane);
        int slot = __builtin_popcountll(q0 & below) + __builtin_popcountll(q1 & below);
        sync();  // (every leader has read its own record: the table can take the next round's)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c_len[c] > 20) {
                FCD_LDS_AS int16_t *f = s->seg[slot++];
                f[0] = (int16_t)c_base[c];
                f[1] = (int16_t)c_len[c];
                f[2] = (int16_t)c_pred[c];
                f[3] = (int16_t)c_flag[c];
            }
        }
        nseg = __builtin_popcountll(q0) + __builtin_popcountll(q1);
        sync();
        FCD_COOP_STAMP(6);
        if (lane == 0) FCD_COOP_COUNT_ROUND;
    }
    FCD_COOP_STAMP(7);

    // ---- leaves: what is left between two boundaries and holds 20 elements or fewer ends in pdqsort's insertion
    // sort -- a stable sort: every element ranks itself inside its leaf.  (Longer stretches are finished regions:
    // runs of equal elements, slices partial_insertion_sort or heapsort completed, serially sorted lists.) ----
    elem_t val[MAXP];
    int dest[MAXP];
    uint64_t cm[MAXP + 1];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) cm[j] = __builtin_amdgcn_ballot_w64(s->cut[64 * j + lane] != 0);
    cm[MAXP] = __builtin_amdgcn_ballot_w64(lane < 8 && s->cut[kPos + lane] != 0);
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int p = 64 * j + lane;
        dest[j] = -1;
        val[j] = 0;
        // (a leaf that starts behind the kept prefix of its list is nobody's business: 20 positions of slack cover
        // every leaf that reaches into it)
        const int reach = keep < kPos ? keep + 20 : kPos;
        const bool in1 = len1 > 0 && p >= start1 && p < start1 + len1, in0 = len0 > 0 && p >= start0 && p < start0 + len0;
        const int list_start = in1 ? start1 : start0;
        const bool in_list = in0 || in1;
        if (__builtin_amdgcn_ballot_w64(in_list && p < list_start + reach) == 0ull) continue;  // (wave-uniform: nothing of this plane matters)
        if (!in_list) continue;
        // nearest boundary at or below p, nearest one above it (every list starts and ends with one)
        int lo = -1, hi = -1;
#pragma unroll
        for (int jj = 0; jj <= MAXP; ++jj) {
            uint64_t up = cm[jj];
            if (jj < j) up = 0;
            if (jj == j) up = lane == 63 ? 0ull : (up & ~coop_detail::bits_below(lane + 1));
            if (hi < 0 && up) hi = 64 * jj + __builtin_ctzll(up);
        }
#pragma unroll
        for (int jj = MAXP - 1; jj >= 0; --jj) {
            uint64_t dn = cm[jj];
            if (jj > j) dn = 0;
            if (jj == j) dn &= coop_detail::bits_below(lane + 1);
            if (lo < 0 && dn) lo = 64 * jj + 63 - __builtin_clzll(dn);
        }
        const int n = hi - lo;
        if (lo < 0 || hi < 0 || n < 2 || n > 20 || lo >= list_start + keep) continue;  // (whole leaves: in or out)
        val[j] = v[p];
        const uint32_t key = (uint32_t)(val[j] >> 32);
        int rank = 0;
        for (int q = lo; q < hi; q += 4) {  // four keys per trip: their loads travel together
            uint32_t kq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kq[u] = (uint32_t)(v[q + u < kPos ? q + u : kPos - 1] >> 32);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                rank += (q + u < hi && (kq[u] > key || (kq[u] == key && q + u < p))) ? 1 : 0;
        }
        dest[j] = lo + rank;
    }
    sync();
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (dest[j] >= 0) v[dest[j]] = val[j];
    sync();
    FCD_COOP_STAMP(8);
    if (lane == 0) FCD_COOP_COUNT_CALL;
}

// The same as a call (the wide-beam kernel: its 128-VGPR budget cannot hold the routine next to its own state).
template <int MAXP>
__device__ __attribute__((noinline)) void coop_sort(elem_t *v, int start0, int len0, int start1, int len1, int keep,
                                                    CoopScratch<MAXP> *s, int lane) {
    coop_sort_inline<MAXP>(v, start0, len0, start1, len1, keep, s, lane);
}

}  // namespace pdq178
}  // namespace fcd
