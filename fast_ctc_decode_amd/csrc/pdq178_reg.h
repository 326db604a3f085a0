// pdq178_reg.h -- pdq178.h's replay of Rust 1.78's sort_unstable_by on a list of at most 64 elements held ONE PER
// LANE in registers: nothing of a partition touches memory.
//
// Why: what a tie-flagged step costs a straggler's wavefront is the replay's instruction count and its LDS round
// trips (pdq178_wave.h).  From 64 elements down, a segment fits the lanes of the wavefront, and everything a partition
// needs is a cross-lane operation:
//   * choose_pivot's samples, the pivot, the first element and the predecessor are v_readlane (scalar results: the
//     sorting network with its swap count runs on the scalar unit);
//   * the classification is ONE vote; `partition`'s scans and partition_in_blocks' block split are bit scans and
//     population counts of that 64-bit mask; an element's index among the misplaced elements of its block is a
//     masked population count;
//   * "the position of the k-th misplaced element" is a ds_permute (element -> lane k), looked up with ds_bpermute; the
//     left-over swaps are the closed form of pdq178_wave.h (parking spots, chains of holes); the partition's whole
//     data movement -- cyclic permutation, parking, swap(0, mid) -- is ONE ds_permute of key and tag;
//   * leaves (<= 20 elements: an insertion sort, i.e. a stable sort) rank themselves with twenty ds_bpermutes.
// The segments of the list are taken one at a time off a stack kept in two registers (frame k in lane k); the rare
// heavy cases (heapsort at limit 0, partial_insertion_sort shifting on >= 50 elements) go through an LDS buffer to
// pdq178.h's serial routines on lane 0.  pdq178_wave.h hands every segment of 64 elements or fewer to this routine
// (with its recursion state: predecessor, limit, was_balanced / was_partitioned), so a 130-element list costs one
// partition through LDS and the rest here.  tests/test_pdq178.py compares the result with the oracle's restatement
// element for element.
#pragma once

#include "pdq178.h"

namespace fcd {
namespace pdq178 {

#ifndef FCD_LDS_AS
#ifdef FCD_HIPEMU
#define FCD_LDS_AS
#else
#define FCD_LDS_AS __attribute__((address_space(3)))
#endif
#endif

namespace reg_detail {

__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint64_t below(int n) {  // bits 0 .. n-1, n in [0, 64]
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}
// ones of `m` at positions below this lane (v_mbcnt_lo / v_mbcnt_hi; tests/hipemu has no such builtin)
__device__ __forceinline__ int ones_below_lane(uint64_t m, int lane) {
#ifdef FCD_HIPEMU
    return __builtin_popcountll(m & below(lane));
#else
    (void)lane;
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#endif
}
__device__ __forceinline__ uint32_t rdl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ int perm(int dst_lane, int v) { return __builtin_amdgcn_ds_permute(dst_lane << 2, v); }

}  // namespace reg_detail

// Sorts the list whose element i is (key, tag) of lane i, i < n <= 64, into the order sort_unstable_by leaves it in --
// as far as its first `keep` positions go -- as the continuation of a quicksort: `has_pred` / `pred_key` = the
// predecessor pivot of the segment (recurse's `pred`), `limit`, `wbal`, `wpar` = its limit and was_balanced /
// was_partitioned (a fresh list: no predecessor, the bit length of n, true, true).  Called by all 64 lanes with the
// same arguments; `spill` = n elements of LDS scratch for the serial fall-backs.
__device__ __forceinline__ void reg_sort(uint32_t &key, uint32_t &tag, const int n_in, const int keep_in, const bool has_pred,
                                         const uint32_t pred_key, const int limit_in, const bool wbal_in, const bool wpar_in,
                                         elem_t *spill_generic, const int lane) {
    using namespace reg_detail;
    typedef FCD_LDS_AS elem_t *vptr;
    const vptr spill = (vptr)spill_generic;
    const int n = uni(n_in), keep = uni(keep_in);
    if (n < 2) return;
    uint64_t cut = 1ull;  // finished boundaries below position 64 (the list's end, position n, is implied)
    int base = 0, len = n, pred = has_pred ? -2 : -1, limit = limit_in;  // pred: lane of the predecessor, -2 = the caller's, -1 = none
    bool wbal = wbal_in, wpar = wpar_in;
    int sp = 0;
    int fr_a = 0, fr_b = 0;  // base | len << 8, (pred + 2) | flags << 8
    bool busy = n > 20;
    auto to_lds = [&]() {
        if (lane < n) spill[lane] = ((elem_t)key << 32) | tag;
        sync();
    };
    auto from_lds = [&]() {
        sync();
        if (lane < n) {
            const elem_t e = spill[lane];
            key = (uint32_t)(e >> 32);
            tag = (uint32_t)e;
        }
    };
    while (busy) {
        bool finished = false;
        int c_base[2] = {0, 0}, c_len[2] = {0, 0}, c_pred[2] = {-1, -1}, c_flag[2] = {0, 0};
        if (limit == 0) {
            to_lds();
            if (lane == 0) heapsort(spill + base, len);
            from_lds();
            finished = true;
        } else {
            if (!wbal) {  // break_patterns: three swaps around the middle, positions from a xorshift seeded by the length
                uint64_t seed = (uint64_t)len;
                uint64_t modulus = 1;
                while (modulus < (uint64_t)len) modulus <<= 1;
                const int pos = len / 4 * 2;
                const bool two_draws = (FCD_PDQ178_FORM() & 1) != 0;  // (pdq178.h: std's generator until 2022)
#pragma unroll 1
                for (int i = 0; i < 3; ++i) {
                    seed ^= seed << 13;
                    seed ^= seed >> 7;
                    seed ^= seed << 17;
                    uint64_t other = (two_draws ? two_draws_number(len, i) : seed) & (modulus - 1);
                    if (other >= (uint64_t)len) other -= (uint64_t)len;
                    const int a = base + pos - 1 + i, b = base + (int)other;
                    const uint32_t ka = rdl(key, a), ta = rdl(tag, a), kb = rdl(key, b), tb = rdl(tag, b);
                    if (lane == a) {
                        key = kb;
                        tag = tb;
                    }
                    if (lane == b) {
                        key = ka;
                        tag = ta;
                    }
                }
                --limit;
            }
            // ---- choose_pivot ----
            const bool ninther = len >= 50;
            const int ia = len / 4, ib = ia * 2, ic = ia * 3;
            uint32_t e[9];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int c = base + (t == 0 ? ia : (t == 1 ? ib : ic));
                e[3 * t] = rdl(key, c - 1);
                e[3 * t + 1] = rdl(key, c);
                e[3 * t + 2] = rdl(key, c + 1);
            }
            int swaps = 0;
            auto srt2 = [&](int &a, uint32_t &ea, int &b, uint32_t &eb) {
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
                if (ninther) {
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
            bool likely_sorted = swaps == 0;
            const bool in_seg = lane >= base && lane < base + len;
            if (swaps >= 12) {  // v.reverse()
                const int src = in_seg ? base + (len - 1 - (lane - base)) : lane;
                key = (uint32_t)bperm(src, (int)key);
                tag = (uint32_t)bperm(src, (int)tag);
                pivot = len - 1 - pivot;
                likely_sorted = true;
            }
            if (wbal && wpar && likely_sorted) {
                // partial_insertion_sort: sorted already?  (every element against its predecessor)
                const uint32_t prevk = (uint32_t)bperm(lane > 0 ? lane - 1 : 0, (int)key);
                const uint64_t desc = __builtin_amdgcn_ballot_w64(lane > base && lane < base + len && key > prevk);
                if (desc == 0ull) {
                    finished = true;
                } else if (len >= 50) {  // it goes on to shift elements about: the serial routine (rare)
                    to_lds();
                    int done = 0;
                    if (lane == 0) done = partial_insertion_sort(spill + base, len) ? 1 : 0;
                    from_lds();
                    finished = __builtin_amdgcn_readlane(done, 0) != 0;
                }
            }
            if (!finished) {
                // ---- swap(0, pivot), in place; which partition ----
                const int ppos = base + pivot;
                const uint32_t kp = rdl(key, ppos), tp = rdl(tag, ppos), kf = rdl(key, base), tf = rdl(tag, base);
                const uint32_t pe = pred >= 0 ? rdl(key, pred) : pred_key;
                const bool equal = pred != -1 && !(pe > kp);  // !less(v[pred], v[pivot])
                // (selects, not branches: a divergent `if` costs four scalar instructions of mask bookkeeping)
                key = lane == ppos ? kf : key;
                tag = lane == ppos ? tf : tag;
                key = lane == base ? kp : key;
                tag = lane == base ? tp : tag;
                const int wb = base + 1, we = base + len;
                const bool in = lane >= wb && lane < we;
                // NORMAL: less(e, pivot) -- the element belongs left.  EQUAL: less(pivot, e) -- it belongs right.
                const bool bit = in && (equal ? kp > key : key > kp);
                const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
                const uint64_t seg = below(we) & ~below(wb);
                int a0, a1, a2, p0, p2, count, cL = 0, cR = 0;
                if (!equal) {
                    const uint64_t zeros = ~m & seg;
                    a0 = zeros ? __builtin_ctzll(zeros) : we;           // while l < r && is_less(v[l], pivot)
                    const uint64_t ones_after = m & ~below(a0);
                    const int last1 = ones_after ? 63 - __builtin_clzll(ones_after) : a0 - 1;
                    a2 = last1 + 1 > a0 ? last1 + 1 : a0;               // while l < r && !is_less(v[r - 1], pivot)
                    const int rem = a2 - a0;
                    a1 = a0 + rem / 2;                                  // block_l = rem / 2, block_r = rem - block_l
                    p0 = a0 - wb;                                       // (everything before a0 is a one)
                    const int p1 = __builtin_popcountll(m & below(a1));
                    p2 = __builtin_popcountll(m);                       // (nothing from a2 on is)
                    cL = (a1 - a0) - (p1 - p0);
                    cR = p2 - p1;
                    count = cL < cR ? cL : cR;
                } else {
                    p0 = 0;
                    p2 = __builtin_popcountll(m);
                    const int nE = (we - wb) - p2;                      // elements equal to the pivot: they end up on the left
                    a0 = wb;
                    a1 = wb + nE;
                    a2 = we;
                    count = __builtin_popcountll(m & below(a1));        // greater ones inside the left zone == equal ones outside it
                }
                const int ones_before = ones_below_lane(m, lane);
                bool is_l, is_r;
                int kk_l, kk_r;
                if (!equal) {
                    is_l = in && lane >= a0 && lane < a1 && !bit;   // offsets_l, in tracing order
                    is_r = in && lane >= a1 && lane < a2 && bit;    // offsets_r, in tracing order (from the right)
                    kk_l = (lane - a0) - (ones_before - p0);
                    kk_r = p2 - ones_before - 1;
                } else {
                    is_l = in && lane < a1 && bit;                  // a greater element inside the left zone, from the left
                    is_r = in && lane >= a1 && !bit;                // an equal element outside it, from the right
                    kk_l = ones_before;
                    kk_r = (a2 - lane - 1) - (p2 - ones_before);
                }
                const int role = is_l ? 1 : (is_r ? 2 : 0);
                const int kk = is_l ? kk_l : kk_r;
                // position of the k-th misplaced element of either side: lane k of two registers (at most 31 pairs: lane
                // 63 takes what has nothing to say)
                const int tab_l = perm(role == 1 ? kk : 63, lane), tab_r = perm(role == 2 ? kk : 63, lane);
                const int m_left = equal ? 0 : cL - count, m_right = equal ? 0 : cR - count;
                const int pm = a1 - m_left + m_right - 1;
                const int zlo = m_left > 0 ? a1 - m_left : a1, zhi = m_left > 0 ? a1 : a1 + m_right;
                const int org_l = a1 - cL, org_r = a1 + cR - 1;
                // after the cyclic permutation (NORMAL: L0 <- R0 <- L1 ... <- L0) / the pairwise swaps (EQUAL)
                const int idx_r = equal ? kk : (kk == 0 ? count - 1 : kk - 1);
                const int from_r = bperm((role == 1 && kk < count) ? idx_r : 63, tab_r);
                const int from_l = bperm((role == 2 && kk < count) ? kk : 63, tab_l);
                // a left-over misplaced element is parked against the block boundary
                const bool parked = role != 0 && kk >= count;
                const int z_pair = role == 1 ? from_r : from_l, z_park = role == 1 ? org_l + kk : org_r - kk;
                int z = role == 0 ? lane : (parked ? z_park : z_pair);
                // a well-placed element a parking swap met goes to that swap's hole -- and on, while the hole is a spot a
                // later swap visits (pdq178_wave.h)
                bool hop = in && !parked && z >= zlo && z < zhi;
                while (__builtin_amdgcn_ballot_w64(hop) != 0ull) {
                    const int j = m_left > 0 ? z - org_l : org_r - z;
                    const int h = bperm(hop ? j : 63, m_left > 0 ? tab_l : tab_r);
                    if (hop) z = h;
                    hop = hop && z >= zlo && z < zhi;
                }
                z = (in && !equal && z == pm) ? base : z;          // swap(0, mid): the last element of the left part ...
                z = lane == base ? (equal ? base : pm) : z;        // ... changes places with the pivot
                key = (uint32_t)perm(z, (int)key);
                tag = (uint32_t)perm(z, (int)tag);
                if (!equal) {
                    const int mid = pm - base;
                    const int smaller = mid < len - mid ? mid : len - mid;
                    const bool nb = smaller >= len / 8, np = a0 >= a2;
                    const int nl = mid, nr = len - mid - 1;
                    cut |= 1ull << pm;
                    if (pm + 1 < 64) cut |= 1ull << (pm + 1);
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
                    if (base + mid < 64) cut |= 1ull << (base + mid);
                    c_base[1] = base + mid;
                    c_len[1] = len - mid;
                    c_pred[1] = pred;
                    c_flag[1] = limit | (wbal ? 256 : 0) | (wpar ? 512 : 0);
                }
            }
        }
        const bool need0 = !finished && c_len[0] > 20 && c_base[0] < keep;
        const bool need1 = !finished && c_len[1] > 20 && c_base[1] < keep;
        if (need0 && need1) {
            if (lane == sp) {
                fr_a = c_base[1] | (c_len[1] << 8);
                fr_b = (c_pred[1] + 2) | (c_flag[1] << 8);
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
            base = fa & 255;
            len = fa >> 8;
            pred = (fb & 255) - 2;
            limit = (fb >> 8) & 255;
            wbal = ((fb >> 8) & 256) != 0;
            wpar = ((fb >> 8) & 512) != 0;
        } else {
            busy = false;
        }
    }
    // ---- leaves: a stretch of 20 elements or fewer between two boundaries ends in an insertion sort -- a stable sort:
    // every element ranks itself inside its leaf ----
    {
        const uint64_t below_w = cut << (63 - lane);  // bit 63 = position `lane` (position 0 is a boundary: never empty)
        const uint64_t above_w = (cut >> 1) >> lane;  // bit 0 = position lane + 1
        const int lo = lane - __builtin_clzll(below_w | 1ull);
        const int hi_c = above_w ? lane + 1 + __builtin_ctzll(above_w) : n;
        const int hi = hi_c < n ? hi_c : n;
        const int ln = hi - lo;
        const bool mine = lane < n && ln >= 2 && ln <= 20 && lo < keep;
        int rank = 0;
        if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
#pragma unroll
            for (int u = 0; u < 20; ++u) {
                const int q = lo + u;
                const uint32_t kq = (uint32_t)bperm(q < 63 ? q : 63, (int)key);
                rank += (mine && q < hi && (kq > key || (kq == key && q < lane))) ? 1 : 0;
            }
            const int dest = mine ? lo + rank : lane;
            key = (uint32_t)perm(dest, (int)key);
            tag = (uint32_t)perm(dest, (int)tag);
        }
    }
}

}  // namespace pdq178
}  // namespace fcd
