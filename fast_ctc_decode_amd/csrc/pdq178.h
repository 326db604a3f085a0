// pdq178.h -- the order Rust 1.78's `sort_unstable_by` leaves EQUAL keys in.
//
// The reference prunes every beam with
//     beam.sort_unstable_by(|a, b| b.probability().partial_cmp(&a.probability()) ...)
// (/root/reference/src/search.rs:122,262, src/duplex.rs:620,807) on a list that is in ascending node order
// (the stable sort_by_key + fold just before it, src/search.rs:245-260).  Up to 20 elements that is an insertion
// sort, which keeps equal probabilities in node order; above 20 it is core::slice::sort::quicksort of the pinned
// toolchain (Rust 1.78.0: .github/workflows/test.yml:16, build-wheels.sh:6) -- a pattern-defeating quicksort whose
// permutation of equal keys is a deterministic function of the whole list.  The kernels rank candidates exactly
// on (probability desc, node asc) without sorting; on the rare steps where a KEPT candidate ties with another one
// among more than 20 candidates, the quicksort is replayed on the node-ordered list -- by one lane with this file's
// routine (generic and duplex kernels), by the whole wavefront with pdq178_wave.h's (register kernels) -- and the
// wavefront adopts the ranks it produces (FCD_TIE_PDQ178, include/fcd.h).
//
// The routine follows library/core/src/slice/sort.rs as of 1.78 (recurse / choose_pivot / partial_insertion_sort /
// partition_equal / partition + partition_in_blocks with BLOCK = 128 / break_patterns / heapsort).  Neither the Rust
// source nor a toolchain exists in this image: it was restated from memory, twice and independently (here and in
// oracle/fcd_oracle.c, DEFINE_PDQSORT), and the two are compared element for element on adversarial lists
// (tests/test_pdq178.py, also on the GPU through fcd_debug_pdq178_sort_dev).  Recursion is an explicit stack:
// the quicksort recurses into the SHORTER side, so 24 frames cover any list below 2^24 elements.
//
// Attribution: the algorithm is core::slice::sort of the Rust standard library, version 1.78.0 (The Rust Project
// Developers; MIT OR Apache-2.0), itself after Orson Peters' pattern-defeating quicksort (zlib licence) with
// BlockQuicksort's partitioning (Edelkamp & Weiss).  No file was copied; what is restated is the sequence of
// comparisons and moves.  PINNING: tools/verify/pdq178_vectors.json holds 1731 lists with the permutation this
// restatement produces and tools/verify/pdq178_check.rs sorts them with rustc's own sort_unstable_by -- one command
// for whoever has the 1.78.0 toolchain; until somebody runs it the claim "follows Rust 1.78" rests on recollection.
//
// An element is a u64: the sort key is the upper word (the orderable bits of the probability, larger = earlier;
// is_less(a, b) = a.key > b.key, i.e. descending probability; keys never are NaNs -- a NaN among two or more
// candidates has already failed the read), the lower word rides along (which candidate this is).
#pragma once

#include <stdint.h>

namespace fcd {
namespace pdq178 {

typedef uint64_t elem_t;

constexpr int kBlock = 128;  // partition_in_blocks::BLOCK
constexpr int kFrames = 24;

struct Scratch {
    uint8_t offl[kBlock];
    uint8_t offr[kBlock];
    int32_t stack[kFrames * 4];
};

#define FCD_PDQ_FN static __device__ inline

// Which FORM of the two routines std changed in 2023 the replay follows -- one word per translation unit and device, 0
// unless somebody sets it (FCD_PDQ178_STD_FORM in the environment, read when the library is loaded and applied by
// fcd_create; fcd_debug_set_pdq178_std_form).  A rustc-1.65 build of std found compiled in the image this was written
// in agrees element for element with this restatement under form 3 (oracle/fcd_oracle.c fcdo_set_pdq_std_form,
// tools/verify/rust165_pdqsort.py); form 0 -- the later forms -- is Rust 1.78 as recalled, and
// tools/verify/pdq178_check.rs tells whoever has that toolchain which form its std carries.
//   bit 0: break_patterns draws a usize as two 32-bit xorshift numbers (13, 17, 5) -- std 1.20 .. 2022 -- instead of
//          one usize-wide xorshift (64-bit: 13, 7, 17; seed = len);
//   bit 1: partial_insertion_sort, having swapped the pair, calls shift_tail(&mut v[..i]) and shift_head(&mut v[i..])
//          -- std .. 2022 -- instead of insertion_sort_shift_left(&mut v[..i], i - 1) and
//          insertion_sort_shift_right(&mut v[..i], 1).
// Both routines run on rare paths of a rare step: the word is read there, nowhere else.  A translation unit that
// defines FCD_PDQ178_FORM0_ONLY before including this file replays form 0 whatever the word says (duplex.hip: its time
// loop fills the instruction cache and any change of the code around it costs 2-3 % -- measured; the duplex searches
// return a label string without a path, and none of 1024 config-5 pairs decodes differently under another form).
static __device__ int g_std_form = 0;
#ifdef FCD_PDQ178_FORM0_ONLY
#define FCD_PDQ178_FORM() 0
#else
#define FCD_PDQ178_FORM() g_std_form
#endif
#define FCD_PDQ178_DEFINE_STD_FORM_SETTER(NAME)                                                     \
    hipError_t NAME(int bits) {                                                                     \
        const int v = bits & 3;                                                                     \
        return hipMemcpyToSymbol(HIP_SYMBOL(pdq178::g_std_form), &v, sizeof(v));                    \
    }
// The list may sit behind a generic pointer (elem_t *) or an LDS one (pdq178_wave.h: ds_* instructions instead of
// flat_* ones -- a third of the latency): the helpers take either.

FCD_PDQ_FN bool less(elem_t a, elem_t b) { return (uint32_t)(a >> 32) > (uint32_t)(b >> 32); }

template <typename V>
FCD_PDQ_FN void swp(V v, int a, int b) {
    const elem_t t = v[a];
    v[a] = v[b];
    v[b] = t;
}

// insert_tail: v[n - 1] into the sorted v[0 .. n - 1)
template <typename V>
FCD_PDQ_FN void insert_tail(V v, int n) {
    const elem_t x = v[n - 1];
    int j = n - 1;
    while (j > 0 && less(x, v[j - 1])) {
        v[j] = v[j - 1];
        --j;
    }
    v[j] = x;
}

// insert_head: v[0] into the sorted v[1 .. n)
template <typename V>
FCD_PDQ_FN void insert_head(V v, int n) {
    if (n < 2 || !less(v[1], v[0])) return;
    const elem_t x = v[0];
    int j = 1;
    v[0] = v[1];
    while (j + 1 < n && less(v[j + 1], x)) {
        v[j] = v[j + 1];
        ++j;
    }
    v[j] = x;
}

template <typename V>
FCD_PDQ_FN void shift_left(V v, int len, int offset) {   // insertion_sort_shift_left
    for (int i = offset; i < len; ++i) insert_tail(v, i + 1);
}

template <typename V>
FCD_PDQ_FN void shift_right(V v, int len, int offset) {  // insertion_sort_shift_right
    for (int i = offset - 1; i >= 0; --i) insert_head(v + i, len - i);
}

// (the earlier forms sit out of line: inlined they grow every replay by a few hundred instructions nobody runs, and the
// duplex kernel -- whose time loop fills the instruction cache -- was 2.8 % slower for it)
template <typename V>
static __device__ __attribute__((noinline)) void shift_pair_earlier(V v, int i, int len) {
    insert_tail(v, i);            // shift_tail(&mut v[..i])
    insert_head(v + i, len - i);  // shift_head(&mut v[i..])
}

// break_patterns' i-th position under the generator of std 1.20 .. 2022: ((gen_u32() as u64) << 32) | (gen_u32() as u64)
static __device__ __attribute__((noinline)) uint64_t two_draws_number(int len, int i) {
    uint32_t r32 = (uint32_t)len;
    uint64_t x = 0;
    for (int k = 0; k <= i; ++k) {
        r32 ^= r32 << 13;
        r32 ^= r32 >> 17;
        r32 ^= r32 << 5;
        x = (uint64_t)r32 << 32;
        r32 ^= r32 << 13;
        r32 ^= r32 >> 17;
        r32 ^= r32 << 5;
        x |= (uint64_t)r32;
    }
    return x;
}

template <typename V>
FCD_PDQ_FN bool partial_insertion_sort(V v, int len) {
    const int kMaxSteps = 5, kShortestShifting = 50;
    int i = 1;
    for (int step = 0; step < kMaxSteps; ++step) {
        while (i < len && !less(v[i], v[i - 1])) ++i;
        if (i == len) return true;
        if (len < kShortestShifting) return false;
        swp(v, i - 1, i);
        if (FCD_PDQ178_FORM() & 2) {  // (std .. 2022)
            shift_pair_earlier(v, i, len);
        } else if (i >= 2) {
            shift_left(v, i, i - 1);
            shift_right(v, i, 1);  // (1.78 hands v[..i] to both)
        }
    }
    return false;
}

template <typename V>
FCD_PDQ_FN void sift_down(V v, int len, int node) {
    for (;;) {
        int child = 2 * node + 1;
        if (child >= len) break;
        if (child + 1 < len && less(v[child], v[child + 1])) ++child;
        if (!less(v[node], v[child])) break;
        swp(v, node, child);
        node = child;
    }
}

template <typename V>
FCD_PDQ_FN void heapsort(V v, int len) {
    for (int i = len / 2 - 1; i >= 0; --i) sift_down(v, len, i);
    for (int i = len - 1; i >= 1; --i) {
        swp(v, 0, i);
        sift_down(v, i, 0);
    }
}

template <typename V>
FCD_PDQ_FN void break_patterns(V v, int len) {
    if (len < 8) return;
    uint64_t seed = (uint64_t)len;  // usize is 64 bits on every platform the reference ships wheels for
    uint64_t modulus = 1;
    while (modulus < (uint64_t)len) modulus <<= 1;  // len.next_power_of_two()
    const int pos = len / 4 * 2;
    const bool two_draws = (FCD_PDQ178_FORM() & 1) != 0;  // (std 1.20 .. 2022)
    for (int i = 0; i < 3; ++i) {
        seed ^= seed << 13;
        seed ^= seed >> 7;
        seed ^= seed << 17;
        uint64_t other = (two_draws ? two_draws_number(len, i) : seed) & (modulus - 1);
        if (other >= (uint64_t)len) other -= (uint64_t)len;
        swp(v, pos - 1 + i, (int)other);
    }
}

template <typename V>
FCD_PDQ_FN void sort2(V v, int &a, int &b, int &swaps) {
    if (less(v[b], v[a])) {
        const int t = a;
        a = b;
        b = t;
        ++swaps;
    }
}

template <typename V>
FCD_PDQ_FN void sort3(V v, int &a, int &b, int &c, int &swaps) {
    sort2(v, a, b, swaps);
    sort2(v, b, c, swaps);
    sort2(v, a, b, swaps);
}

template <typename V>
FCD_PDQ_FN int choose_pivot(V v, int len, bool &likely_sorted) {
    const int kShortestMedianOfMedians = 50, kMaxSwaps = 4 * 3;
    int a = len / 4 * 1, b = len / 4 * 2, c = len / 4 * 3;
    int swaps = 0;
    if (len >= 8) {
        if (len >= kShortestMedianOfMedians) {
            int lo, hi;
            lo = a - 1, hi = a + 1;
            sort3(v, lo, a, hi, swaps);
            lo = b - 1, hi = b + 1;
            sort3(v, lo, b, hi, swaps);
            lo = c - 1, hi = c + 1;
            sort3(v, lo, c, hi, swaps);
        }
        sort3(v, a, b, c, swaps);
    }
    if (swaps < kMaxSwaps) {
        likely_sorted = swaps == 0;
        return b;
    }
    for (int i = 0; i < len / 2; ++i) swp(v, i, len - 1 - i);  // v.reverse()
    likely_sorted = true;
    return len - 1 - b;
}

FCD_PDQ_FN int partition_in_blocks(elem_t *v, int n, elem_t pivot, Scratch *s) {
    int l = 0, block_l = kBlock, r = n, block_r = kBlock;
    int sl = 0, el = 0, sr = 0, er = 0;
    uint8_t *offl = s->offl, *offr = s->offr;
    for (;;) {
        const bool is_done = (r - l) <= 2 * kBlock;
        if (is_done) {
            int rem = r - l;
            if (sl < el || sr < er) rem -= kBlock;
            if (sl < el) {
                block_r = rem;
            } else if (sr < er) {
                block_l = rem;
            } else {
                block_l = rem / 2;
                block_r = rem - block_l;
            }
        }
        if (sl == el) {  // trace block_l elements from the left: the ones that belong on the right
            sl = el = 0;
            for (int i = 0; i < block_l; ++i) {
                offl[el] = (uint8_t)i;
                el += less(v[l + i], pivot) ? 0 : 1;
            }
        }
        if (sr == er) {  // trace block_r elements from the right: the ones that belong on the left
            sr = er = 0;
            for (int i = 0; i < block_r; ++i) {
                offr[er] = (uint8_t)i;
                er += less(v[r - 1 - i], pivot) ? 1 : 0;
            }
        }
        const int count = (el - sl) < (er - sr) ? (el - sl) : (er - sr);
        if (count > 0) {  // one cyclic permutation instead of `count` swaps
            const elem_t tmp = v[l + offl[sl]];
            v[l + offl[sl]] = v[r - offr[sr] - 1];
            for (int k = 1; k < count; ++k) {
                ++sl;
                v[r - offr[sr] - 1] = v[l + offl[sl]];
                ++sr;
                v[l + offl[sl]] = v[r - offr[sr] - 1];
            }
            v[r - offr[sr] - 1] = tmp;
            ++sl;
            ++sr;
        }
        if (sl == el) l += block_l;
        if (sr == er) r -= block_r;
        if (is_done) break;
    }
    if (sl < el) {
        while (sl < el) {
            --el;
            swp(v, l + offl[el], r - 1);
            --r;
        }
        return r;
    }
    if (sr < er) {
        while (sr < er) {
            --er;
            swp(v, l, r - offr[er] - 1);
            ++l;
        }
        return l;
    }
    return l;
}

FCD_PDQ_FN int partition(elem_t *v, int len, int pivot_idx, bool &was_partitioned, Scratch *s) {
    swp(v, 0, pivot_idx);
    const elem_t pivot = v[0];
    elem_t *w = v + 1;
    int l = 0, r = len - 1;
    while (l < r && less(w[l], pivot)) ++l;
    while (l < r && !less(w[r - 1], pivot)) --r;
    const int mid = l + partition_in_blocks(w + l, r - l, pivot, s);
    was_partitioned = l >= r;
    swp(v, 0, mid);
    return mid;
}

template <typename V>
FCD_PDQ_FN int partition_equal(V v, int len, int pivot_idx) {
    swp(v, 0, pivot_idx);
    const elem_t pivot = v[0];
    V w = v + 1;
    const int wn = len - 1;
    if (wn == 0) return 0;
    int l = 0, r = wn;
    for (;;) {
        while (l < r && !less(pivot, w[l])) ++l;
        for (;;) {
            --r;
            if (l >= r || !less(pivot, w[r])) break;
        }
        if (l >= r) break;
        swp(w, l, r);
        ++l;
    }
    return l + 1;
}

// a[0 .. n) into the order sort_unstable_by leaves it in.  One caller (lane) per list; `s` is that caller's own.
static __device__ __attribute__((noinline)) void sort_desc(elem_t *a, int n, Scratch *s) {
    if (n < 2) return;
    int limit = 0;  // usize::BITS - len.leading_zeros()
    for (uint32_t m = (uint32_t)n; m; m >>= 1) ++limit;
    int sp = 0;
    int base = 0, len = n, pred = -1;
    bool was_balanced = true, was_partitioned = true;
    for (;;) {
        elem_t *v = a + base;
        bool finished = false;
        if (len <= 20) {  // MAX_INSERTION
            if (len >= 2) shift_left(v, len, 1);
            finished = true;
        } else if (limit == 0) {
            heapsort(v, len);
            finished = true;
        } else {
            if (!was_balanced) {
                break_patterns(v, len);
                --limit;
            }
            bool likely_sorted = false;
            const int pivot = choose_pivot(v, len, likely_sorted);
            if (was_balanced && was_partitioned && likely_sorted && partial_insertion_sort(v, len)) {
                finished = true;
            } else if (pred >= 0 && !less(a[pred], v[pivot])) {
                // the pivot equals the predecessor: everything equal to it goes left and is done
                const int mid = partition_equal(v, len, pivot);
                base += mid;
                len -= mid;
                continue;
            } else {
                bool was_p = false;
                const int mid = partition(v, len, pivot, was_p, s);
                const int smaller = mid < len - mid ? mid : len - mid;
                was_balanced = smaller >= len / 8;
                was_partitioned = was_p;
                const int nl = mid, nr = len - mid - 1;
                // recurse into the shorter side, carry on with the longer one: the frame keeps the longer side
                int32_t *f = s->stack + 4 * sp++;
                f[3] = limit | (was_balanced ? 256 : 0) | (was_partitioned ? 512 : 0);
                if (nl < nr) {
                    f[0] = base + mid + 1;
                    f[1] = nr;
                    f[2] = base + mid;
                    len = nl;
                } else {
                    f[0] = base;
                    f[1] = nl;
                    f[2] = pred;
                    pred = base + mid;
                    base = base + mid + 1;
                    len = nr;
                }
                was_balanced = was_partitioned = true;
                continue;
            }
        }
        if (finished) {
            if (sp == 0) return;
            const int32_t *f = s->stack + 4 * --sp;
            base = f[0];
            len = f[1];
            pred = f[2];
            limit = f[3] & 255;
            was_balanced = (f[3] & 256) != 0;
            was_partitioned = (f[3] & 512) != 0;
        }
    }
}

#undef FCD_PDQ_FN

}  // namespace pdq178
}  // namespace fcd
