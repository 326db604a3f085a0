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
struct CoopScratch {
    static constexpr int kPos = 64 * MAXP;
    static constexpr int kMaxSeg = kPos / 21 + 2;  // segments longer than 20 elements that can coexist
    uint16_t pos_a[kPos];        // position of the k-th misplaced element of a segment's left side (at [wb + k])
    uint16_t pos_b[kPos];        // ... of its right side, counted from the right end
    uint8_t cut[kPos + 8];       // 1 = a finished boundary: a leaf / pivot / list starts here
    int16_t seg[2][kMaxSeg][4];  // {base, len, pred, limit | was_balanced << 8 | was_partitioned << 9}, this round / next
    uint32_t key[kMaxSeg];       // this round: the pivot's key ...
    int16_t rt[kMaxSeg][14];     // ... and what the segment's leader decided (fields: enum below)
};  // (16-bit fields: the wide-beam kernel has 2.5 KB of LDS left at four wavefronts per SIMD)

enum { kActDone = 0, kActNormal = 1, kActEqual = 2 };
enum { R_ACT = 0, R_WB, R_WE, R_A0, R_A1, R_A2, R_P0, R_P1, R_P2, R_COUNT, R_CL, R_CR };

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
// overlap and must lie inside [0, 64 * MAXP)) into the order sort_unstable_by leaves them in.  Called by all 64 lanes
// of a wavefront with the same arguments; `lane` = the caller's lane; v and s are LDS.
template <int MAXP>
__device__ __attribute__((noinline)) void coop_sort(elem_t *v, int start0, int len0, int start1, int len1,
                                                    CoopScratch<MAXP> *s, int lane) {
    using namespace coop_detail;
    constexpr int kPos = 64 * MAXP;
    static_assert(sizeof(Scratch) * 2 <= sizeof(uint16_t) * 2 * kPos || kPos < kCoopMaxLen,
                  "the serial fall-back borrows the position tables as its scratch");

    // ---- boundaries known from the start; lists that are leaves or too long for one-round partitions ----
    for (int p = lane; p < kPos + 8; p += 64) s->cut[p] = 0;
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
                int16_t *f = s->seg[0][nseg++];
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
        Scratch *ser = reinterpret_cast<Scratch *>(s->pos_a);
        if (lane == 0 && len0 > kCoopMaxLen) sort_desc(v + start0, len0, ser);
        if (lane == 32 && len1 > kCoopMaxLen) sort_desc(v + start1, len1, ser + 1);
    }
    sync();

    int cur = 0;
    while (nseg > 0) {
        // ---- A: every segment's leader picks the pivot (and does what only ever touches a few elements) ----
        int base = 0, len = 0, pred = -1, limit = 0;
        bool wbal = true, wpar = true;
        int act = kActDone;
        if (lane < nseg) {
            const int16_t *f = s->seg[cur][lane];
            base = f[0];
            len = f[1];
            pred = f[2];
            limit = f[3] & 255;
            wbal = (f[3] & 256) != 0;
            wpar = (f[3] & 512) != 0;
            elem_t *w = v + base;
            if (limit == 0) {
                heapsort(w, len);
            } else {
                if (!wbal) {
                    break_patterns(w, len);
                    --limit;
                }
                bool likely_sorted = false;
                const int pivot = choose_pivot(w, len, likely_sorted);
                if (wbal && wpar && likely_sorted && partial_insertion_sort(w, len)) {
                    act = kActDone;
                } else {
                    act = (pred >= 0 && !less(v[pred], w[pivot])) ? kActEqual : kActNormal;
                    swp(w, 0, pivot);
                }
            }
            int16_t *r = s->rt[lane];
            r[R_ACT] = (int16_t)act;
            s->key[lane] = (uint32_t)(w[0] >> 32);
            r[R_WB] = (int16_t)(base + 1);
            r[R_WE] = (int16_t)(base + len);
        }
        sync();

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
            const int16_t *r = s->rt[sg];
            const int a = r[R_ACT], wb = r[R_WB], we = r[R_WE];
            const uint32_t pk = s->key[sg];
            if (a == kActDone) continue;
#pragma unroll
            for (int j = 0; j < MAXP; ++j) {
                const int p = 64 * j + lane;
                if (p >= wb && p < we) {
                    myseg[j] = sg;
                    const uint32_t k = (uint32_t)(val[j] >> 32);
                    // NORMAL: less(e, pivot) -- the element belongs left.  EQUAL: less(pivot, e) -- it belongs right.
                    bit[j] = a == kActNormal ? k > pk : pk > k;
                }
            }
        }
        Masks<MAXP> mk;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) mk.m[j] = __builtin_amdgcn_ballot_w64(bit[j]);

        // ---- C: leaders: the scans of `partition`, the block split of partition_in_blocks, the counts ----
        if (lane < nseg && act != kActDone) {
            int16_t *r = s->rt[lane];
            const int wb = base + 1, we = base + len;
            if (act == kActNormal) {
                const int l_abs = mk.first_zero(wb, we);            // while l < r && is_less(v[l], pivot)
                const int last1 = mk.last_one(l_abs, we);
                const int r_abs = last1 + 1 > l_abs ? last1 + 1 : l_abs;  // while l < r && !is_less(v[r - 1], pivot)
                const int rem = r_abs - l_abs;                      // <= 2 * BLOCK: one round, is_done at once
                const int s_abs = l_abs + rem / 2;                  // block_l = rem / 2, block_r = rem - block_l
                const int p0 = mk.prefix1(l_abs), p1 = mk.prefix1(s_abs), p2 = mk.prefix1(r_abs);
                const int cL = (s_abs - l_abs) - (p1 - p0);         // left block: elements that are NOT less than the pivot
                const int cR = p2 - p1;                             // right block: elements that are
                r[R_A0] = (int16_t)(l_abs);
                r[R_A1] = (int16_t)(s_abs);
                r[R_A2] = (int16_t)(r_abs);
                r[R_P0] = (int16_t)(p0);
                r[R_P1] = (int16_t)(p1);
                r[R_P2] = (int16_t)(p2);
                r[R_CL] = (int16_t)(cL);
                r[R_CR] = (int16_t)(cR);
                r[R_COUNT] = (int16_t)(cL < cR ? cL : cR);
            } else {
                const int p0 = mk.prefix1(wb), p2 = mk.prefix1(we);
                const int nE = (we - wb) - (p2 - p0);               // elements equal to the pivot: they end up on the left
                const int z_abs = wb + nE;
                const int p1 = mk.prefix1(z_abs);
                r[R_A0] = (int16_t)(wb);
                r[R_A1] = (int16_t)(z_abs);
                r[R_A2] = (int16_t)(we);
                r[R_P0] = (int16_t)(p0);
                r[R_P1] = (int16_t)(p1);
                r[R_P2] = (int16_t)(p2);
                r[R_COUNT] = (int16_t)(p1 - p0);                               // greater ones inside the left zone == equal ones outside it
            }
        }
        sync();

        // ---- D: every misplaced element's index among the misplaced ones of its side -> position tables ----
        int role[MAXP], kk[MAXP];
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            role[j] = 0;
            kk[j] = 0;
            if (myseg[j] < 0) continue;
            const int p = 64 * j + lane;
            const int16_t *r = s->rt[myseg[j]];
            const int a0 = r[R_A0], a1 = r[R_A1], a2 = r[R_A2], wb = r[R_WB];
            const int ones_before = mk.prefix1(p);
            if (r[R_ACT] == kActNormal) {
                if (p >= a0 && p < a1 && !bit[j]) {         // offsets_l, in tracing order (left to right)
                    role[j] = 1;
                    kk[j] = (p - a0) - (ones_before - r[R_P0]);
                    s->pos_a[wb + kk[j]] = (uint16_t)p;
                } else if (p >= a1 && p < a2 && bit[j]) {   // offsets_r, in tracing order (right to left)
                    role[j] = 2;
                    kk[j] = r[R_P2] - ones_before - 1;
                    s->pos_b[wb + kk[j]] = (uint16_t)p;
                }
            } else {
                if (p < a1 && bit[j]) {                     // a greater element inside the left zone, from the left
                    role[j] = 1;
                    kk[j] = ones_before - r[R_P0];
                    s->pos_a[wb + kk[j]] = (uint16_t)p;
                } else if (p >= a1 && !bit[j]) {            // an equal element outside it, from the right
                    role[j] = 2;
                    kk[j] = (a2 - p - 1) - (r[R_P2] - ones_before);
                    s->pos_b[wb + kk[j]] = (uint16_t)p;
                }
            }
        }
        sync();

        // ---- E: the moves.  NORMAL: the cyclic permutation L0 <- R0 <- L1 <- R1 ... <- R(count-1) <- L0 of the first
        // `count` misplaced pairs; EQUAL: the k-th greater element from the left swaps with the k-th equal one from
        // the right.  Every mover still holds its own value in a register. ----
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (role[j] == 0) continue;
            const int16_t *r = s->rt[myseg[j]];
            const int count = r[R_COUNT], wb = r[R_WB];
            if (kk[j] >= count) continue;
            int dest;
            if (r[R_ACT] == kActNormal)
                dest = role[j] == 1 ? s->pos_b[wb + (kk[j] == 0 ? count - 1 : kk[j] - 1)] : s->pos_a[wb + kk[j]];
            else
                dest = role[j] == 1 ? s->pos_b[wb + kk[j]] : s->pos_a[wb + kk[j]];
            v[dest] = val[j];
        }
        sync();

        // ---- F: leaders: park the left-over misplaced elements, put the pivot in place, queue the children ----
        int c_base[2] = {0, 0}, c_len[2] = {0, 0}, c_pred[2] = {-1, -1}, c_flag[2] = {0, 0};
        if (lane < nseg && act != kActDone) {
            const int16_t *r = s->rt[lane];
            const int wb = base + 1;
            if (act == kActNormal) {
                const int count = r[R_COUNT], cL = r[R_CL], cR = r[R_CR];
                int bound = r[R_A1];
                if (cL > cR) {          // while start_l < end_l { end_l -= 1; swap(l + *end_l, r - 1); r -= 1 }
                    for (int j = cL - 1; j >= count; --j) {
                        --bound;
                        const int hole = s->pos_a[wb + j];
                        const elem_t t = v[hole];
                        v[hole] = v[bound];
                        v[bound] = t;
                    }
                } else if (cR > cL) {   // while start_r < end_r { end_r -= 1; swap(l, r - *end_r - 1); l += 1 }
                    for (int j = cR - 1; j >= count; --j) {
                        const int hole = s->pos_b[wb + j];
                        const elem_t t = v[hole];
                        v[hole] = v[bound];
                        v[bound] = t;
                        ++bound;
                    }
                }
                const int pm = bound - 1;  // where the pivot belongs
                const elem_t t = v[base];
                v[base] = v[pm];
                v[pm] = t;
                const int mid = pm - base;
                const int smaller = mid < len - mid ? mid : len - mid;
                const bool nb = smaller >= len / 8, np = r[R_A0] >= r[R_A2];
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
                const int mid = (r[R_A1] - wb) + 1;  // the elements equal to the pivot (and the pivot) are done
                s->cut[base + mid] = 1;
                c_base[1] = base + mid;
                c_len[1] = len - mid;
                c_pred[1] = pred;
                c_flag[1] = limit | (wbal ? 256 : 0) | (wpar ? 512 : 0);
            }
        }
        const uint64_t q0 = __builtin_amdgcn_ballot_w64(c_len[0] > 20), q1 = __builtin_amdgcn_ballot_w64(c_len[1] > 20);
        const uint64_t below = bits_below(lane);
        int slot = __builtin_popcountll(q0 & below) + __builtin_popcountll(q1 & below);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c_len[c] > 20) {
                int16_t *f = s->seg[cur ^ 1][slot++];
                f[0] = (int16_t)c_base[c];
                f[1] = (int16_t)c_len[c];
                f[2] = (int16_t)c_pred[c];
                f[3] = (int16_t)c_flag[c];
            }
        }
        nseg = __builtin_popcountll(q0) + __builtin_popcountll(q1);
        cur ^= 1;
        sync();
    }

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
        const bool in_list = (len0 > 0 && p >= start0 && p < start0 + len0) || (len1 > 0 && p >= start1 && p < start1 + len1);
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
        if (lo < 0 || hi < 0 || n < 2 || n > 20) continue;
        val[j] = v[p];
        const uint32_t key = (uint32_t)(val[j] >> 32);
        int rank = 0;
        for (int q = lo; q < hi; ++q) {
            const uint32_t kq = (uint32_t)(v[q] >> 32);
            rank += (kq > key || (kq == key && q < p)) ? 1 : 0;
        }
        dest[j] = lo + rank;
    }
    sync();
#pragma unroll
    for (int j = 0; j < MAXP; ++j)
        if (dest[j] >= 0) v[dest[j]] = val[j];
    sync();
}

}  // namespace pdq178
}  // namespace fcd
