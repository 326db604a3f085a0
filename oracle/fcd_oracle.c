/*
 * fcd_oracle.c -- CPU restatement of nanoporetech/fast-ctc-decode (v0.3.7) search algorithms.
 *
 * TEST INFRASTRUCTURE ONLY (see fcd_oracle.h).  Plain C99, single-threaded per read.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile) -- f32 arithmetic must not
 * be contracted into FMAs: the reference multiplies and adds in separate roundings.
 *
 * Citations are path:line relative to /root/reference.
 */
#include "fcd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define FCDO_PANIC 100 /* the reference would panic (= abort, Cargo.toml:40) on this input */

/* ------------------------------------------------------------------------------------------
 * Suffix tree: src/tree.rs:4-194 + src/vec2d.rs.  Append-only arena; node index == creation
 * order (tree.rs:129), children row of n_labels i32 per node initialised to -1 (tree.rs:143),
 * root children kept apart (tree.rs:40-43).  `data` is the usize payload of the 1D searches
 * (creation time); the duplex searches keep their payload in a parallel array.
 * ---------------------------------------------------------------------------------------- */
struct fcdo_tree {
    int64_t n_labels;
    int64_t len, cap;
    int32_t *parent;
    int32_t *label;
    int64_t *data;
    int32_t *children; /* len * n_labels */
    int32_t *root_children;
};

static fcdo_tree *tree_new_cap(int64_t n_labels, int64_t cap);

fcdo_tree *fcdo_tree_new(int64_t n_labels) { return tree_new_cap(n_labels, 1024); }

static void tree_reset(fcdo_tree *t) {
    t->len = 0;
    for (int64_t i = 0; i < t->n_labels; ++i) t->root_children[i] = -1;
}

static fcdo_tree *tree_new_cap(int64_t n_labels, int64_t cap) {
    fcdo_tree *t = (fcdo_tree *)calloc(1, sizeof(*t));
    t->n_labels = n_labels;
    t->cap = cap > 16 ? cap : 16;
    t->parent = (int32_t *)malloc(sizeof(int32_t) * t->cap);
    t->label = (int32_t *)malloc(sizeof(int32_t) * t->cap);
    t->data = (int64_t *)malloc(sizeof(int64_t) * t->cap);
    t->children = (int32_t *)malloc(sizeof(int32_t) * t->cap * (n_labels > 0 ? n_labels : 1));
    t->root_children = (int32_t *)malloc(sizeof(int32_t) * (n_labels > 0 ? n_labels : 1));
    for (int64_t i = 0; i < n_labels; ++i) t->root_children[i] = -1;
    return t;
}

void fcdo_tree_free(fcdo_tree *t) {
    if (!t) return;
    free(t->parent);
    free(t->label);
    free(t->data);
    free(t->children);
    free(t->root_children);
    free(t);
}

/* tree.rs:125-145 */
int32_t fcdo_tree_add_node(fcdo_tree *t, int32_t parent, int64_t label, int64_t data) {
    if (t->len == t->cap) {
        t->cap *= 2;
        t->parent = (int32_t *)realloc(t->parent, sizeof(int32_t) * t->cap);
        t->label = (int32_t *)realloc(t->label, sizeof(int32_t) * t->cap);
        t->data = (int64_t *)realloc(t->data, sizeof(int64_t) * t->cap);
        t->children = (int32_t *)realloc(t->children, sizeof(int32_t) * t->cap * t->n_labels);
    }
    int32_t idx = (int32_t)t->len;
    if (parent == -1)
        t->root_children[label] = idx;
    else
        t->children[(int64_t)parent * t->n_labels + label] = idx;
    t->parent[idx] = parent;
    t->label[idx] = (int32_t)label;
    t->data[idx] = data;
    for (int64_t i = 0; i < t->n_labels; ++i) t->children[(int64_t)idx * t->n_labels + i] = -1;
    t->len++;
    return idx;
}

/* tree.rs:147-161 */
int32_t fcdo_tree_get_child(const fcdo_tree *t, int32_t node, int64_t label) {
    int32_t idx = (node == -1) ? t->root_children[label]
                               : t->children[(int64_t)node * t->n_labels + label];
    return idx >= 0 ? idx : -1;
}

/* tree.rs:104-110 */
int64_t fcdo_tree_label(const fcdo_tree *t, int32_t node) {
    return node >= 0 ? t->label[node] : -1;
}
int32_t fcdo_tree_parent(const fcdo_tree *t, int32_t node) { return t->parent[node]; }
int64_t fcdo_tree_data(const fcdo_tree *t, int32_t node) { return t->data[node]; }
int64_t fcdo_tree_len(const fcdo_tree *t) { return t->len; }

/* ------------------------------------------------------------------------------------------
 * phred: src/search.rs:31-36
 * ---------------------------------------------------------------------------------------- */
static uint32_t sat_u32(float q) { /* Rust `as u32`: saturating, NaN -> 0 */
    if (!(q == q)) return 0;
    if (q <= 0.0f) return 0;
    if (q >= 4294967296.0f) return 4294967295u;
    return (uint32_t)q;
}

static uint32_t phred_code(float prob, float qscale, float qbias) {
    const float max = 1e-4f;
    float om = 1.0f - prob;
    float p = (om < max) ? max : om;
    float q = -10.0f * log10f(p) * qscale + qbias;
    return sat_u32(roundf(q)) + 33u; /* f32::round = half away from zero = roundf */
}

char fcdo_phred(float prob, float qscale, float qbias) {
    return (char)phred_code(prob, qscale, qbias);
}

/* ------------------------------------------------------------------------------------------
 * viterbi_search: src/search.rs:303-383
 * ---------------------------------------------------------------------------------------- */
int fcdo_viterbi_search(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                        int collapse_repeats, float qscale, float qbias,
                        int32_t *labels, int64_t *path, uint32_t *quals, int64_t *n_out) {
    if (T <= 0 || N <= 0) return FCDO_PANIC; /* assert!(!network_output.is_empty()) :329 */
    int64_t n = 0, nq = 0;
    int64_t last_label = -1;
    int64_t count = 0;
    float total = 0.0f;
    for (int64_t t = 0; t < T; ++t) {
        const float *pr = x + t * rs;
        /* find_max :303-318: strict '>' so the first maximum wins; a NaN in column 0 sticks */
        int64_t label = 0;
        float prob = pr[0];
        for (int64_t j = 1; j < N; ++j) {
            float v = pr[j * cs];
            if (v > prob) {
                label = j;
                prob = v;
            }
        }
        if (label != 0 && (!collapse_repeats || last_label != label)) { /* :347 */
            if (count > 0) {
                if (quals) quals[nq] = phred_code(total / (float)count, qscale, qbias);
                nq++;
                total = 0.0f;
                count = 0;
            }
            labels[n] = (int32_t)label;
            path[n] = t;
            n++;
        }
        if (label != 0) { /* :362-365 */
            total += prob;
            count++;
        }
        last_label = label;
    }
    if (count > 0) { /* :370-376 */
        if (quals) quals[nq] = phred_code(total / (float)count, qscale, qbias);
        nq++;
    }
    *n_out = n;
    return FCDO_OK;
}

/* ------------------------------------------------------------------------------------------
 * 1D beam search: src/search.rs:7-28 (SearchPoint), :159-301 (beam_search), :38-157 (crf)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t node;
    int64_t state;
    float label_prob;
    float gap_prob;
} sp1;

static inline float sp1_prob(const sp1 *p) { return p->label_prob + p->gap_prob; } /* :25-27 */

typedef struct {
    sp1 *v;
    int64_t len, cap;
} sp1vec;

static void sp1_push(sp1vec *b, sp1 p) {
    if (b->len == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        b->v = (sp1 *)realloc(b->v, sizeof(sp1) * b->cap);
    }
    b->v[b->len++] = p;
}

/*
 * Stable sorts.  Rust's sort_by_key (:245) is stable; sort_unstable_by (:262) on <= 20 elements
 * is an insertion sort (behaves stably); above 20 it is pdqsort whose tie order is
 * implementation-defined -- PARITY UNPINNED, this restatement stays stable.
 */
#define DEFINE_STABLE_SORT(NAME, TYPE, LESS)                                              \
    static void NAME##_insertion(TYPE *v, int64_t n) {                                    \
        for (int64_t i = 1; i < n; ++i) {                                                 \
            TYPE x = v[i];                                                                \
            int64_t j = i;                                                                \
            while (j > 0 && LESS(&x, &v[j - 1])) {                                        \
                v[j] = v[j - 1];                                                          \
                --j;                                                                      \
            }                                                                             \
            v[j] = x;                                                                     \
        }                                                                                 \
    }                                                                                     \
    static void NAME##_merge(TYPE *v, TYPE *tmp, int64_t n) {                             \
        if (n <= 20) {                                                                    \
            NAME##_insertion(v, n);                                                       \
            return;                                                                       \
        }                                                                                 \
        int64_t h = n / 2;                                                                \
        NAME##_merge(v, tmp, h);                                                          \
        NAME##_merge(v + h, tmp, n - h);                                                  \
        memcpy(tmp, v, sizeof(TYPE) * h);                                                 \
        int64_t i = 0, j = h, k = 0;                                                      \
        while (i < h && j < n) {                                                          \
            if (LESS(&v[j], &tmp[i]))                                                     \
                v[k++] = v[j++];                                                          \
            else                                                                          \
                v[k++] = tmp[i++];                                                        \
        }                                                                                 \
        while (i < h) v[k++] = tmp[i++];                                                  \
    }                                                                                     \
    static void NAME(TYPE *v, int64_t n, TYPE **tmp, int64_t *tmpcap) {                   \
        if (n <= 20) {                                                                    \
            NAME##_insertion(v, n);                                                       \
            return;                                                                       \
        }                                                                                 \
        if (*tmpcap < n) {                                                                \
            *tmpcap = n * 2;                                                              \
            *tmp = (TYPE *)realloc(*tmp, sizeof(TYPE) * (*tmpcap));                       \
        }                                                                                 \
        NAME##_merge(v, *tmp, n);                                                         \
    }

#define SP1_NODE_LESS(a, b) ((a)->node < (b)->node)
#define SP1_PROB_GREATER(a, b) (sp1_prob(a) > sp1_prob(b)) /* descending :263-265 */
DEFINE_STABLE_SORT(sp1_sort_node, sp1, SP1_NODE_LESS)
DEFINE_STABLE_SORT(sp1_sort_prob, sp1, SP1_PROB_GREATER)


/*
 * Rust 1.78 core::slice::sort::quicksort -- what `sort_unstable_by` (src/search.rs:122,262, src/duplex.rs:620,807)
 * runs above 20 elements -- restated FROM MEMORY of library/core/src/slice/sort.rs as of 1.78 (pattern-defeating
 * quicksort: insertion sort up to 20, choose_pivot = median of three / Tukey ninther from 50 with swap counting
 * and slice reversal at 12 swaps, partial_insertion_sort on likely-sorted slices, partition_equal against the
 * predecessor pivot, BlockQuicksort partition_in_blocks with BLOCK = 128 and its cyclic block swaps,
 * break_patterns with the xorshift generator seeded by the length, heapsort once the imbalance budget
 * floor(log2 n) + 1 is spent).  Neither the Rust source nor a Rust toolchain exists in this environment; what does
 * exist (found in round 5) is a rustc-1.65 build of std, compiled into libcst's native module with its symbols:
 * tools/verify/rust165_pdqsort.py calls its core::slice::sort::recurse on (key, node) records.  With the EARLIER
 * forms of the two routines std changed in 2023 (fcdo_set_pdq_std_form below) this restatement equals that binary
 * element for element on every committed vector and on 1.7 million random lists of 2 .. 5000 elements (half of them
 * reaching break_patterns, a fifth shifting in partial_insertion_sort, heapsort included): pivot choice, both
 * partitions, the recursion with its limit and flags, the insertion sorts, heapsort and the reversal are PINNED to a
 * compiled std.  The default forms of those two routines are Rust 1.78 as recalled -- the part that is still only
 * recollection, and what tools/verify/pdq178_check.rs asks a 1.78 toolchain.  Every sort is a correct descending
 * sort either way; only the order of EQUAL keys is at stake (tools/pdqsort_ties.py).  Since
 * round 4 it is the oracle's default and the kernels follow the same order (FCD_TIE_PDQ178, csrc/pdq178.h: a second
 * restatement, written separately, compared with this one element for element in tests/test_pdq178.py); "ties keep
 * ascending node order" (FCD_TIE_STABLE) remains selectable on both sides.
 */
static int g_unstable_sort_mode = 1; /* 1 = the pdqsort restatement above (default since round 4: the product's
                                      * default is FCD_TIE_PDQ178, which csrc/pdq178.h restates separately),
                                      * 0 = the stable rule (FCD_TIE_STABLE) */
void fcdo_set_unstable_sort(int mode) { g_unstable_sort_mode = mode ? 1 : 0; }
int fcdo_get_unstable_sort(void) { return g_unstable_sort_mode; }
/* The two routines of std's pdqsort that CHANGED between the toolchain a binary in this image was built with (rustc
 * 1.65: libcst's native module, whose compiled core::slice::sort::recurse tools/verify/rust165_pdqsort.py calls) and
 * the reference's 1.78, as far as the authors recall -- selectable so that the REST of the restatement can be checked
 * against that binary (with both bits set it agrees with it on every committed vector, element for element):
 *   bit 0  break_patterns' random numbers: 0 = `seed = len`, ONE usize-wide xorshift per number (64-bit: 13, 7, 17),
 *          std 2023 .. 1.80; 1 = two 32-bit xorshift draws (13, 17, 5) glued into a usize, std 1.20 .. 2022;
 *   bit 1  partial_insertion_sort after swapping the out-of-order pair: 0 = insertion_sort_shift_left(&mut v[..i], i - 1)
 *          then insertion_sort_shift_right(&mut v[..i], 1) under `if i >= 2` (the 2023 refactor of the insertion sorts:
 *          BOTH on v[..i]); 1 = shift_tail(&mut v[..i]) then shift_head(&mut v[i..]), std .. 2022.
 * Default 0 = Rust 1.78 as recalled; the kernels restate form 0 only (csrc/pdq178.h says where the two spots are).
 * Test hook. */
static int g_pdq_std_form = 0;
static _Thread_local int64_t g_pdq_break_calls = 0, g_pdq_shift_calls = 0;
void fcdo_set_pdq_std_form(int bits) { g_pdq_std_form = bits & 3; }
int fcdo_get_pdq_std_form(void) { return g_pdq_std_form; }
/* how often this thread's quicksorts reached break_patterns / shifted elements in partial_insertion_sort */
int64_t fcdo_pdq_break_patterns_calls(int reset) {
    const int64_t n = g_pdq_break_calls;
    if (reset) g_pdq_break_calls = 0;
    return n;
}
int64_t fcdo_pdq_partial_shift_calls(int reset) {
    const int64_t n = g_pdq_shift_calls;
    if (reset) g_pdq_shift_calls = 0;
    return n;
}

/* Test hook (tools/verify/rust165_pdqsort.py, tests/test_rust165_pdqsort.py): an EXTERNAL routine with the signature of
 * core::slice::sort::recurse<T, F> over 24-byte records ordered by their first u64 -- the one a rustc-1.65 build of std
 * carries inside libcst's native module -- sorts in place of the restatement: the searches of this file then run on
 * Rust's own quicksort.  The records' keys are the elements' ranks under the comparator (equal elements, equal keys), the
 * second word says which element it is. */
typedef void (*fcdo_external_recurse)(void *v, size_t len, void *is_less, const void *pred, uint32_t limit);
static fcdo_external_recurse g_external_recurse = NULL;
void fcdo_set_external_recurse(void *fn) { g_external_recurse = (fcdo_external_recurse)fn; }

#define DEFINE_PDQSORT(NAME, TYPE, LESS)                                                                      \
    static void NAME##_swap(TYPE *a, TYPE *b) {                                                               \
        TYPE t = *a;                                                                                          \
        *a = *b;                                                                                              \
        *b = t;                                                                                               \
    }                                                                                                         \
    /* insert_tail: v[n-1] into the sorted v[..n-1] */                                                        \
    static void NAME##_insert_tail(TYPE *v, int64_t n) {                                                      \
        if (n >= 2 && LESS(&v[n - 1], &v[n - 2])) {                                                           \
            TYPE tmp = v[n - 1];                                                                              \
            v[n - 1] = v[n - 2];                                                                              \
            int64_t hole = n - 2;                                                                             \
            for (int64_t j = n - 3; j >= 0; --j) {                                                            \
                if (!LESS(&tmp, &v[j])) break;                                                                \
                v[j + 1] = v[j];                                                                              \
                hole = j;                                                                                     \
            }                                                                                                 \
            v[hole] = tmp;                                                                                    \
        }                                                                                                     \
    }                                                                                                         \
    /* insert_head: v[0] into the sorted v[1..] */                                                            \
    static void NAME##_insert_head(TYPE *v, int64_t n) {                                                      \
        if (n >= 2 && LESS(&v[1], &v[0])) {                                                                   \
            TYPE tmp = v[0];                                                                                  \
            v[0] = v[1];                                                                                      \
            int64_t hole = 1;                                                                                 \
            for (int64_t i = 2; i < n; ++i) {                                                                 \
                if (!LESS(&v[i], &tmp)) break;                                                                \
                v[i - 1] = v[i];                                                                              \
                hole = i;                                                                                     \
            }                                                                                                 \
            v[hole] = tmp;                                                                                    \
        }                                                                                                     \
    }                                                                                                         \
    static void NAME##_shift_left(TYPE *v, int64_t len, int64_t offset) {                                     \
        for (int64_t i = offset; i < len; ++i) NAME##_insert_tail(v, i + 1);                                  \
    }                                                                                                         \
    static void NAME##_shift_right(TYPE *v, int64_t len, int64_t offset) {                                    \
        for (int64_t i = offset - 1; i >= 0; --i) NAME##_insert_head(v + i, len - i);                         \
    }                                                                                                         \
    static int NAME##_partial_insertion_sort(TYPE *v, int64_t len) {                                          \
        const int64_t MAX_STEPS = 5, SHORTEST_SHIFTING = 50;                                                  \
        int64_t i = 1;                                                                                        \
        for (int64_t step = 0; step < MAX_STEPS; ++step) {                                                    \
            while (i < len && !LESS(&v[i], &v[i - 1])) ++i;                                                   \
            if (i == len) return 1;                                                                           \
            if (len < SHORTEST_SHIFTING) return 0;                                                            \
            NAME##_swap(&v[i - 1], &v[i]);                                                                    \
            ++g_pdq_shift_calls;                                                                              \
            if (g_pdq_std_form & 2) { /* std until the insertion-sort refactor of 2023 (bit 1, above) */      \
                NAME##_insert_tail(v, i);            /* shift_tail(&mut v[..i]) */                            \
                NAME##_insert_head(v + i, len - i);  /* shift_head(&mut v[i..]) */                            \
            } else if (i >= 2) {                                                                              \
                NAME##_shift_left(v, i, i - 1);  /* the smaller element to the left */                        \
                NAME##_shift_right(v, i, 1);     /* (1.78 passes v[..i] here too) */                          \
            }                                                                                                 \
        }                                                                                                     \
        return 0;                                                                                             \
    }                                                                                                         \
    static void NAME##_heapsort_sift(TYPE *v, int64_t len, int64_t node) {                                    \
        for (;;) {                                                                                            \
            int64_t child = 2 * node + 1;                                                                     \
            if (child >= len) break;                                                                          \
            if (child + 1 < len) child += LESS(&v[child], &v[child + 1]) ? 1 : 0;                             \
            if (!LESS(&v[node], &v[child])) break;                                                            \
            NAME##_swap(&v[node], &v[child]);                                                                 \
            node = child;                                                                                     \
        }                                                                                                     \
    }                                                                                                         \
    static void NAME##_heapsort(TYPE *v, int64_t len) {                                                       \
        for (int64_t i = len / 2 - 1; i >= 0; --i) NAME##_heapsort_sift(v, len, i);                           \
        for (int64_t i = len - 1; i >= 1; --i) {                                                              \
            NAME##_swap(&v[0], &v[i]);                                                                        \
            NAME##_heapsort_sift(v, i, 0);                                                                    \
        }                                                                                                     \
    }                                                                                                         \
    static void NAME##_break_patterns(TYPE *v, int64_t len) {                                                 \
        ++g_pdq_break_calls;                                                                                  \
        if (len < 8) return;                                                                                  \
        uint64_t seed = (uint64_t)len;                                                                        \
        uint32_t r32 = (uint32_t)len;                                                                         \
        uint64_t modulus = 1;                                                                                 \
        while (modulus < (uint64_t)len) modulus <<= 1; /* next_power_of_two */                                \
        int64_t pos = len / 4 * 2;                                                                            \
        for (int64_t i = 0; i < 3; ++i) {                                                                     \
            if (!(g_pdq_std_form & 1)) {                                                                      \
                seed ^= seed << 13;                                                                           \
                seed ^= seed >> 7;                                                                            \
                seed ^= seed << 17;                                                                           \
            } else { /* ((gen_u32() as u64) << 32) | (gen_u32() as u64) */                                    \
                r32 ^= r32 << 13;                                                                             \
                r32 ^= r32 >> 17;                                                                             \
                r32 ^= r32 << 5;                                                                              \
                seed = (uint64_t)r32 << 32;                                                                   \
                r32 ^= r32 << 13;                                                                             \
                r32 ^= r32 >> 17;                                                                             \
                r32 ^= r32 << 5;                                                                              \
                seed |= (uint64_t)r32;                                                                        \
            }                                                                                                 \
            uint64_t other = seed & (modulus - 1);                                                            \
            if (other >= (uint64_t)len) other -= (uint64_t)len;                                               \
            NAME##_swap(&v[pos - 1 + i], &v[other]);                                                          \
        }                                                                                                     \
    }                                                                                                         \
    static void NAME##_sort2(const TYPE *v, int64_t *a, int64_t *b, int *swaps) {                             \
        if (LESS(&v[*b], &v[*a])) {                                                                           \
            int64_t t = *a;                                                                                   \
            *a = *b;                                                                                          \
            *b = t;                                                                                           \
            ++*swaps;                                                                                         \
        }                                                                                                     \
    }                                                                                                         \
    static void NAME##_sort3(const TYPE *v, int64_t *a, int64_t *b, int64_t *c, int *swaps) {                 \
        NAME##_sort2(v, a, b, swaps);                                                                         \
        NAME##_sort2(v, b, c, swaps);                                                                         \
        NAME##_sort2(v, a, b, swaps);                                                                         \
    }                                                                                                         \
    static int64_t NAME##_choose_pivot(TYPE *v, int64_t len, int *likely_sorted) {                            \
        const int64_t SHORTEST_MEDIAN_OF_MEDIANS = 50;                                                        \
        const int MAX_SWAPS = 4 * 3;                                                                          \
        int64_t a = len / 4 * 1, b = len / 4 * 2, c = len / 4 * 3;                                            \
        int swaps = 0;                                                                                        \
        if (len >= 8) {                                                                                       \
            if (len >= SHORTEST_MEDIAN_OF_MEDIANS) {                                                          \
                int64_t lo, hi;                                                                               \
                lo = a - 1, hi = a + 1;                                                                       \
                NAME##_sort3(v, &lo, &a, &hi, &swaps);                                                        \
                lo = b - 1, hi = b + 1;                                                                       \
                NAME##_sort3(v, &lo, &b, &hi, &swaps);                                                        \
                lo = c - 1, hi = c + 1;                                                                       \
                NAME##_sort3(v, &lo, &c, &hi, &swaps);                                                        \
            }                                                                                                 \
            NAME##_sort3(v, &a, &b, &c, &swaps);                                                              \
        }                                                                                                     \
        if (swaps < MAX_SWAPS) {                                                                              \
            *likely_sorted = swaps == 0;                                                                      \
            return b;                                                                                         \
        }                                                                                                     \
        for (int64_t i = 0; i < len / 2; ++i) NAME##_swap(&v[i], &v[len - 1 - i]); /* v.reverse() */          \
        *likely_sorted = 1;                                                                                   \
        return len - 1 - b;                                                                                   \
    }                                                                                                         \
    static int64_t NAME##_partition_in_blocks(TYPE *v, int64_t n, const TYPE *pivot) {                        \
        enum { BLOCK = 128 };                                                                                 \
        int64_t l = 0, block_l = BLOCK, r = n, block_r = BLOCK;                                               \
        int sl = 0, el = 0, sr = 0, er = 0;                                                                   \
        uint8_t offl[BLOCK], offr[BLOCK];                                                                     \
        for (;;) {                                                                                            \
            const int is_done = (r - l) <= 2 * BLOCK;                                                         \
            if (is_done) {                                                                                    \
                int64_t rem = r - l;                                                                          \
                if (sl < el || sr < er) rem -= BLOCK;                                                         \
                if (sl < el) {                                                                                \
                    block_r = rem;                                                                            \
                } else if (sr < er) {                                                                         \
                    block_l = rem;                                                                            \
                } else {                                                                                      \
                    block_l = rem / 2;                                                                        \
                    block_r = rem - block_l;                                                                  \
                }                                                                                             \
            }                                                                                                 \
            if (sl == el) { /* trace block_l elements from the left */                                        \
                sl = el = 0;                                                                                  \
                for (int64_t i = 0; i < block_l; ++i) {                                                       \
                    offl[el] = (uint8_t)i;                                                                    \
                    el += LESS(&v[l + i], pivot) ? 0 : 1;                                                     \
                }                                                                                             \
            }                                                                                                 \
            if (sr == er) { /* trace block_r elements from the right */                                       \
                sr = er = 0;                                                                                  \
                for (int64_t i = 0; i < block_r; ++i) {                                                       \
                    offr[er] = (uint8_t)i;                                                                    \
                    er += LESS(&v[r - 1 - i], pivot) ? 1 : 0;                                                 \
                }                                                                                             \
            }                                                                                                 \
            const int count = (el - sl) < (er - sr) ? (el - sl) : (er - sr);                                  \
            if (count > 0) { /* one cyclic permutation instead of `count` swaps */                            \
                TYPE tmp = v[l + offl[sl]];                                                                   \
                v[l + offl[sl]] = v[r - offr[sr] - 1];                                                        \
                for (int k = 1; k < count; ++k) {                                                             \
                    ++sl;                                                                                     \
                    v[r - offr[sr] - 1] = v[l + offl[sl]];                                                    \
                    ++sr;                                                                                     \
                    v[l + offl[sl]] = v[r - offr[sr] - 1];                                                    \
                }                                                                                             \
                v[r - offr[sr] - 1] = tmp;                                                                    \
                ++sl;                                                                                         \
                ++sr;                                                                                         \
            }                                                                                                 \
            if (sl == el) l += block_l;                                                                       \
            if (sr == er) r -= block_r;                                                                       \
            if (is_done) break;                                                                               \
        }                                                                                                     \
        if (sl < el) {                                                                                        \
            while (sl < el) {                                                                                 \
                --el;                                                                                         \
                NAME##_swap(&v[l + offl[el]], &v[r - 1]);                                                     \
                --r;                                                                                          \
            }                                                                                                 \
            return r;                                                                                         \
        } else if (sr < er) {                                                                                 \
            while (sr < er) {                                                                                 \
                --er;                                                                                         \
                NAME##_swap(&v[l], &v[r - offr[er] - 1]);                                                     \
                ++l;                                                                                          \
            }                                                                                                 \
            return l;                                                                                         \
        }                                                                                                     \
        return l;                                                                                             \
    }                                                                                                         \
    static int64_t NAME##_partition(TYPE *v, int64_t len, int64_t pivot_idx, int *was_partitioned) {          \
        NAME##_swap(&v[0], &v[pivot_idx]);                                                                    \
        const TYPE pivot = v[0];                                                                              \
        TYPE *w = v + 1;                                                                                      \
        int64_t l = 0, r = len - 1;                                                                           \
        while (l < r && LESS(&w[l], &pivot)) ++l;                                                             \
        while (l < r && !LESS(&w[r - 1], &pivot)) --r;                                                        \
        const int64_t mid = l + NAME##_partition_in_blocks(w + l, r - l, &pivot);                             \
        *was_partitioned = l >= r;                                                                            \
        NAME##_swap(&v[0], &v[mid]);                                                                          \
        return mid;                                                                                           \
    }                                                                                                         \
    static int64_t NAME##_partition_equal(TYPE *v, int64_t len, int64_t pivot_idx) {                          \
        NAME##_swap(&v[0], &v[pivot_idx]);                                                                    \
        const TYPE pivot = v[0];                                                                              \
        TYPE *w = v + 1;                                                                                      \
        const int64_t wn = len - 1;                                                                           \
        if (wn == 0) return 0;                                                                                \
        int64_t l = 0, r = wn;                                                                                \
        for (;;) {                                                                                            \
            while (l < r && !LESS(&pivot, &w[l])) ++l;                                                        \
            for (;;) {                                                                                        \
                --r;                                                                                          \
                if (l >= r || !LESS(&pivot, &w[r])) break;                                                    \
            }                                                                                                 \
            if (l >= r) break;                                                                                \
            NAME##_swap(&w[l], &w[r]);                                                                        \
            ++l;                                                                                              \
        }                                                                                                     \
        return l + 1;                                                                                         \
    }                                                                                                         \
    static void NAME##_recurse(TYPE *v, int64_t len, const TYPE *pred, uint32_t limit) {                      \
        const int64_t MAX_INSERTION = 20;                                                                     \
        int was_balanced = 1, was_partitioned = 1;                                                            \
        for (;;) {                                                                                            \
            if (len <= MAX_INSERTION) {                                                                       \
                if (len >= 2) NAME##_shift_left(v, len, 1);                                                   \
                return;                                                                                       \
            }                                                                                                 \
            if (limit == 0) {                                                                                 \
                NAME##_heapsort(v, len);                                                                      \
                return;                                                                                       \
            }                                                                                                 \
            if (!was_balanced) {                                                                              \
                NAME##_break_patterns(v, len);                                                                \
                --limit;                                                                                      \
            }                                                                                                 \
            int likely_sorted = 0;                                                                            \
            const int64_t pivot = NAME##_choose_pivot(v, len, &likely_sorted);                                \
            if (was_balanced && was_partitioned && likely_sorted) {                                           \
                if (NAME##_partial_insertion_sort(v, len)) return;                                            \
            }                                                                                                 \
            if (pred && !LESS(pred, &v[pivot])) {                                                             \
                const int64_t mid = NAME##_partition_equal(v, len, pivot);                                    \
                v += mid;                                                                                     \
                len -= mid;                                                                                   \
                continue;                                                                                     \
            }                                                                                                 \
            int was_p = 0;                                                                                    \
            const int64_t mid = NAME##_partition(v, len, pivot, &was_p);                                      \
            const int64_t smaller = mid < len - mid ? mid : len - mid;                                        \
            was_balanced = smaller >= len / 8;                                                                \
            was_partitioned = was_p;                                                                          \
            TYPE *left = v, *pv = v + mid, *right = v + mid + 1;                                              \
            const int64_t nl = mid, nr = len - mid - 1;                                                       \
            if (nl < nr) {                                                                                    \
                NAME##_recurse(left, nl, pred, limit);                                                        \
                v = right;                                                                                    \
                len = nr;                                                                                     \
                pred = pv;                                                                                    \
            } else {                                                                                          \
                NAME##_recurse(right, nr, pv, limit);                                                         \
                v = left;                                                                                     \
                len = nl;                                                                                     \
            }                                                                                                 \
        }                                                                                                     \
    }                                                                                                         \
    static void NAME(TYPE *v, int64_t len) {                                                                  \
        uint32_t limit = 0; /* usize::BITS - len.leading_zeros() = floor(log2 len) + 1 */                     \
        for (uint64_t n = (uint64_t)len; n; n >>= 1) ++limit;                                                 \
        if (g_external_recurse && len >= 2) {                                                                 \
            uint64_t *rec = (uint64_t *)malloc(sizeof(uint64_t) * 3 * (size_t)len);                           \
            TYPE *copy = (TYPE *)malloc(sizeof(TYPE) * (size_t)len);                                          \
            uint64_t dummy[8] = {0};                                                                          \
            for (int64_t i = 0; i < len; ++i) {                                                               \
                uint64_t key = 0; /* how many elements come strictly before this one */                       \
                for (int64_t j = 0; j < len; ++j) key += LESS(&v[j], &v[i]) ? 1 : 0;                          \
                rec[3 * i] = key;                                                                             \
                rec[3 * i + 1] = (uint64_t)i;                                                                 \
                rec[3 * i + 2] = 0;                                                                           \
                copy[i] = v[i];                                                                               \
            }                                                                                                 \
            g_external_recurse(rec, (size_t)len, dummy, NULL, limit);                                         \
            for (int64_t i = 0; i < len; ++i) v[i] = copy[rec[3 * i + 1]];                                    \
            free(copy);                                                                                       \
            free(rec);                                                                                        \
            return;                                                                                           \
        }                                                                                                     \
        NAME##_recurse(v, len, NULL, limit);                                                                  \
    }

DEFINE_PDQSORT(sp1_pdq_prob, sp1, SP1_PROB_GREATER)

/* Test hook: sorts n (probability, node) pairs by descending probability with the pdqsort restatement;
 * `node` rides along so that the order of equal probabilities can be inspected. */
void fcdo_test_pdqsort(float *prob, int32_t *node, int64_t n) {
    sp1 *v = (sp1 *)malloc(sizeof(sp1) * (n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        sp1 e = {node[i], 0, prob[i], 0.0f};
        v[i] = e;
    }
    sp1_pdq_prob(v, n);
    for (int64_t i = 0; i < n; ++i) {
        prob[i] = v[i].label_prob;
        node[i] = v[i].node;
    }
    free(v);
}

/*
 * Shared tail of every 1D step: :245-282 (== :105-142).  Returns FCDO_* status.
 */
/*
 * Exhaustive tie enumeration (fcdo_beam_search_all_tie_orders).  The output of a 1D beam search is a
 * function of the KEPT SET of every step and of which entry is beam[0] after the last step: the order of
 * the kept entries only feeds node numbering, and node numbers only feed tie-breaking; the merge sums have
 * at most two non-zero addends (order-free); normalisation divides by the maximum, which tied entries share;
 * a node's creation time is the first step at which it is a candidate.  So the only places where ANY tie
 * rule -- Rust's pdqsort above 20 candidates included -- can change the result are (a) a group of equal
 * probabilities straddling the truncation boundary (which m of the k tied candidates are kept) and (b) equal
 * probabilities at the top after the last step (which of them is walked).  A tie_ctx replays the search
 * under one resolution of every such event; the driver below enumerates them all like an odometer.
 */
typedef struct {
    int64_t *choice, *nopt;
    int64_t cap, depth, pos;
    int overflow;
} tie_ctx;

static int64_t tie_next_choice(tie_ctx *tc, int64_t n_options) {
    if (tc->pos >= tc->cap) {
        tc->overflow = 1;
        return 0;
    }
    int64_t i = tc->pos++;
    if (i >= tc->depth) {
        tc->choice[i] = 0;
        tc->depth = i + 1;
    }
    tc->nopt[i] = n_options;
    return tc->choice[i] < n_options ? tc->choice[i] : 0;
}

static int64_t n_choose_k(int64_t n, int64_t k, int64_t cap) {
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    int64_t r = 1;
    for (int64_t i = 1; i <= k; ++i) {
        r = r * (n - k + i) / i;
        if (r > cap) return cap + 1;
    }
    return r;
}

/*
 * n_amb (nullable) counts the steps whose outcome the restatement cannot pin to the reference: the
 * merged list holds more than 20 candidates (so Rust 1.78's sort_unstable_by is a true pdqsort whose
 * tie order is implementation-defined) AND a candidate that survives the truncation has
 * exactly the probability of another candidate (kept or dropped).  Ties among dropped candidates never
 * matter.  With no such step in a read, the kept list -- set AND order -- of every > 20-candidate step is
 * the same under any tie order, every <= 20-candidate step is Rust's stable insertion sort, which the
 * stable sort here reproduces, so node numbering, beam and output follow the reference step for step.
 */
static int sp1_merge_prune(sp1vec *beam, int64_t beam_size, sp1 **tmp, int64_t *tmpcap, int64_t *n_amb,
                           tie_ctx *tc, int last_step) {
    sp1_sort_node(beam->v, beam->len, tmp, tmpcap); /* :245 stable */
    /* :246-260 fold equal nodes into the first occurrence, then retain */
    int64_t w = 0;
    for (int64_t i = 0; i < beam->len; ++i) {
        if (w > 0 && beam->v[w - 1].node == beam->v[i].node) {
            beam->v[w - 1].label_prob += beam->v[i].label_prob;
            beam->v[w - 1].gap_prob += beam->v[i].gap_prob;
        } else {
            beam->v[w++] = beam->v[i];
        }
    }
    beam->len = w;
    /* :261-272 every element of a >= 2 element slice takes part in >= 1 partial_cmp, so any
     * NaN probability sets has_nans */
    if (beam->len >= 2) {
        for (int64_t i = 0; i < beam->len; ++i) {
            float p = sp1_prob(&beam->v[i]);
            if (p != p) return FCDO_INCOMPARABLE;
        }
    }
    sp1 *stable_copy = NULL;
    if (g_unstable_sort_mode == 1 && beam->len > 20) {
        /* the counters below are defined on the stably sorted list: take them there, then impose the order of
         * the pdqsort restatement (same multiset of probabilities, possibly another order of equal ones) */
        stable_copy = (sp1 *)malloc(sizeof(sp1) * beam->len);
        memcpy(stable_copy, beam->v, sizeof(sp1) * beam->len);
        sp1_pdq_prob(stable_copy, beam->len);
    }
    sp1_sort_prob(beam->v, beam->len, tmp, tmpcap);
    if (n_amb) {
        /* sorted: equal probabilities are adjacent.  n_amb[0]: > 20 candidates and a kept candidate ties
         * with another one; n_amb[1]: (any candidate count) a tie across the truncation boundary or
         * between ranks 0 and 1 -- the ties that can change the kept set or the best entry */
        if (beam->len > 20) {
            int tie = 0;
            for (int64_t i = 0; i < beam_size && i + 1 < beam->len; ++i)
                tie |= sp1_prob(&beam->v[i]) == sp1_prob(&beam->v[i + 1]);
            if (tie) ++n_amb[0];
        }
        int crit = beam->len >= 2 && sp1_prob(&beam->v[0]) == sp1_prob(&beam->v[1]);
        crit |= beam->len > beam_size && sp1_prob(&beam->v[beam_size - 1]) == sp1_prob(&beam->v[beam_size]);
        if (crit) ++n_amb[1];
    }
    if (stable_copy) {
        memcpy(beam->v, stable_copy, sizeof(sp1) * beam->len);
        free(stable_copy);
    }
    if (tc && beam->len > beam_size &&
        sp1_prob(&beam->v[beam_size - 1]) == sp1_prob(&beam->v[beam_size])) {
        /* (a) k equal candidates [g0, g1) straddle the boundary, m of them fit: take the c-th m-subset
         * (lexicographic; c = 0 is the stable rule) by moving the chosen ones to the front of the group */
        float pb = sp1_prob(&beam->v[beam_size]);
        int64_t g0 = beam_size - 1, g1 = beam_size + 1;
        while (g0 > 0 && sp1_prob(&beam->v[g0 - 1]) == pb) --g0;
        while (g1 < beam->len && sp1_prob(&beam->v[g1]) == pb) ++g1;
        int64_t k = g1 - g0, m = beam_size - g0;
        int64_t n_sub = n_choose_k(k, m, 1 << 20);
        if (n_sub > (1 << 20)) tc->overflow = 1;
        int64_t c = tie_next_choice(tc, n_sub);
        if (c > 0 && k <= 64) {
            sp1 grp[64], pick[64], rest[64];
            memcpy(grp, beam->v + g0, sizeof(sp1) * k);
            int64_t np = 0, nr = 0, need = m;
            for (int64_t j = 0; j < k; ++j) { /* unrank: is element j in the c-th subset? */
                int64_t with_j = need > 0 ? n_choose_k(k - j - 1, need - 1, 1 << 20) : 0;
                if (need > 0 && c < with_j) {
                    pick[np++] = grp[j];
                    --need;
                } else {
                    rest[nr++] = grp[j];
                    c -= with_j;
                }
            }
            memcpy(beam->v + g0, pick, sizeof(sp1) * np);
            memcpy(beam->v + g0 + np, rest, sizeof(sp1) * nr);
        } else if (c > 0) {
            tc->overflow = 1;
        }
    }
    if (beam->len > beam_size) beam->len = beam_size; /* :273 */
    if (tc && last_step && beam->len >= 2 && sp1_prob(&beam->v[0]) == sp1_prob(&beam->v[1])) {
        /* (b) the walk starts at beam[0]: any of the kept entries tied for the top may be it */
        int64_t h = 2;
        while (h < beam->len && sp1_prob(&beam->v[h]) == sp1_prob(&beam->v[0])) ++h;
        int64_t c = tie_next_choice(tc, h);
        if (c > 0) {
            sp1 t0 = beam->v[0];
            beam->v[0] = beam->v[c];
            beam->v[c] = t0;
        }
    }
    if (beam->len == 0) return FCDO_RAN_OUT_OF_BEAM;  /* :274-277 */
    float top = sp1_prob(&beam->v[0]);                /* :278-282 */
    for (int64_t i = 0; i < beam->len; ++i) {
        beam->v[i].label_prob /= top;
        beam->v[i].gap_prob /= top;
    }
    return FCDO_OK;
}

/* :285-300: walk leaf -> root, then reverse */
static int64_t tree_walk_1d(const fcdo_tree *tree, int32_t node, int32_t *labels, int64_t *path) {
    int64_t n = 0;
    for (int32_t cur = node; cur >= 0; cur = tree->parent[cur]) {
        labels[n] = tree->label[cur] + 1;
        if (path) path[n] = tree->data[cur];
        n++;
    }
    for (int64_t i = 0; i < n / 2; ++i) {
        int32_t tl = labels[i];
        labels[i] = labels[n - 1 - i];
        labels[n - 1 - i] = tl;
        if (path) {
            int64_t tp = path[i];
            path[i] = path[n - 1 - i];
            path[n - 1 - i] = tp;
        }
    }
    return n;
}

/* Reusable per-thread scratch of the batch driver (avoids allocator traffic in the timed loop). */
typedef struct {
    fcdo_tree *tree;
    sp1vec beam, next;
    sp1 *tmp;
    int64_t tmpcap;
} beam_ws;

static int beam_search_ws(beam_ws *ws, const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                          int64_t beam_size, float thr, int collapse_repeats, int32_t *labels,
                          int64_t *path, int64_t *n_out, int64_t *n_nodes_out, int64_t *n_amb, tie_ctx *tc);

int fcdo_beam_search(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                     int64_t beam_size, float thr, int collapse_repeats,
                     int32_t *labels, int64_t *path, int64_t *n_out, int64_t *n_nodes_out) {
    return fcdo_beam_search_ex(x, T, N, rs, cs, beam_size, thr, collapse_repeats, labels, path, n_out,
                               n_nodes_out, NULL);
}

int fcdo_beam_search_ex(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                        int64_t beam_size, float thr, int collapse_repeats,
                        int32_t *labels, int64_t *path, int64_t *n_out, int64_t *n_nodes_out,
                        int64_t *n_ambiguous_out) {
    beam_ws ws;
    memset(&ws, 0, sizeof(ws));
    ws.tree = fcdo_tree_new(N - 1);
    if (n_ambiguous_out) n_ambiguous_out[0] = n_ambiguous_out[1] = 0;
    int st = beam_search_ws(&ws, x, T, N, rs, cs, beam_size, thr, collapse_repeats, labels, path,
                            n_out, n_nodes_out, n_ambiguous_out, NULL);
    free(ws.beam.v);
    free(ws.next.v);
    free(ws.tmp);
    fcdo_tree_free(ws.tree);
    return st;
}

/* Runs the search under EVERY resolution of the ties that can change the result (see tie_ctx).  labels /
 * path / n_out receive the stable-rule result (the first branch).  Returns its status; *n_branches = the
 * number of resolutions explored, *n_distinct = 1 when they all produced the same (status, labels, path),
 * 2 otherwise; *complete = 0 if max_branches (or the subset bound) stopped the enumeration early. */
int fcdo_beam_search_all_tie_orders(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                                    int64_t beam_size, float thr, int collapse_repeats,
                                    int32_t *labels, int64_t *path, int64_t *n_out,
                                    int64_t max_branches, int64_t *n_branches, int64_t *n_distinct,
                                    int *complete) {
    beam_ws ws;
    memset(&ws, 0, sizeof(ws));
    ws.tree = fcdo_tree_new(N - 1);
    tie_ctx tc;
    memset(&tc, 0, sizeof(tc));
    tc.cap = 4096;
    tc.choice = (int64_t *)calloc(tc.cap, sizeof(int64_t));
    tc.nopt = (int64_t *)calloc(tc.cap, sizeof(int64_t));
    int64_t cap_out = T > 0 ? T : 1;
    int32_t *l2 = (int32_t *)malloc(sizeof(int32_t) * cap_out);
    int64_t *p2 = (int64_t *)malloc(sizeof(int64_t) * cap_out);
    int64_t branches = 0, distinct = 1, n0 = 0;
    int st0 = FCDO_OK, done = 0;
    *complete = 1;
    while (!done) {
        tc.pos = 0;
        int64_t n = 0;
        int st = beam_search_ws(&ws, x, T, N, rs, cs, beam_size, thr, collapse_repeats,
                                branches == 0 ? labels : l2, branches == 0 ? path : p2, &n, NULL, NULL, &tc);
        if (branches == 0) {
            st0 = st;
            n0 = (st == FCDO_OK) ? n : 0;
        } else if (st != st0 || (st == FCDO_OK && (n != n0 || memcmp(l2, labels, sizeof(int32_t) * n) ||
                                                   memcmp(p2, path, sizeof(int64_t) * n)))) {
            distinct = 2;
        }
        ++branches;
        /* odometer: bump the last event that still has an untried option, forget everything after it */
        int64_t i = tc.pos - 1;
        while (i >= 0 && tc.choice[i] + 1 >= tc.nopt[i]) --i;
        if (i < 0) {
            done = 1;
        } else {
            tc.choice[i]++;
            tc.depth = i + 1;
        }
        if (tc.overflow || (!done && branches >= max_branches)) {
            *complete = 0;
            done = 1;
        }
    }
    *n_out = n0;
    *n_branches = branches;
    *n_distinct = distinct;
    free(tc.choice);
    free(tc.nopt);
    free(l2);
    free(p2);
    free(ws.beam.v);
    free(ws.next.v);
    free(ws.tmp);
    fcdo_tree_free(ws.tree);
    return st0;
}

static int beam_search_ws(beam_ws *ws, const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                          int64_t beam_size, float thr, int collapse_repeats, int32_t *labels,
                          int64_t *path, int64_t *n_out, int64_t *n_nodes_out, int64_t *n_amb, tie_ctx *tc) {
    int64_t alphabet_size = N - 1; /* :167 */
    fcdo_tree *tree = ws->tree;
    tree_reset(tree);
    sp1vec beam = ws->beam, next = ws->next;
    beam.len = 0;
    next.len = 0;
    sp1 *tmp = ws->tmp;
    int64_t tmpcap = ws->tmpcap;
    int status = FCDO_OK;
    sp1 root = {-1, 0, 0.0f, 1.0f}; /* :170-175 */
    sp1_push(&beam, root);

    for (int64_t idx = 0; idx < T; ++idx) { /* :178 */
        const float *pr = x + idx * rs;
        next.len = 0;
        float pr0 = pr[0];
        for (int64_t bi = 0; bi < beam.len; ++bi) {
            sp1 b = beam.v[bi];
            int64_t tip_label = fcdo_tree_label(tree, b.node); /* :187 */
            if (pr0 > thr) {                                   /* :191-198 */
                sp1 c = {b.node, b.state, 0.0f, (b.label_prob + b.gap_prob) * pr0};
                sp1_push(&next, c);
            }
            for (int64_t label = 0; label < alphabet_size; ++label) { /* :200 */
                float pr_b = pr[(label + 1) * cs];
                if (pr_b < thr) continue; /* :201 */
                if (collapse_repeats && label == tip_label) { /* :205 */
                    sp1 stay = {b.node, b.state, b.label_prob * pr_b, 0.0f};
                    sp1_push(&next, stay);
                    int32_t child = fcdo_tree_get_child(tree, b.node, label);
                    if (child < 0 && b.gap_prob > 0.0f) /* :212-218 */
                        child = fcdo_tree_add_node(tree, b.node, label, idx);
                    if (child >= 0) { /* :220-227 */
                        sp1 c = {child, b.state, b.gap_prob * pr_b, 0.0f};
                        sp1_push(&next, c);
                    }
                } else { /* :228-239 */
                    int32_t child = fcdo_tree_get_child(tree, b.node, label);
                    if (child < 0) child = fcdo_tree_add_node(tree, b.node, label, idx);
                    sp1 c = {child, b.state, (b.label_prob + b.gap_prob) * pr_b, 0.0f};
                    sp1_push(&next, c);
                }
            }
        }
        sp1vec t = beam; /* :242 swap */
        beam = next;
        next = t;
        status = sp1_merge_prune(&beam, beam_size, &tmp, &tmpcap, n_amb, tc, idx == T - 1);
        if (status != FCDO_OK) break;
    }

    if (status == FCDO_OK) *n_out = tree_walk_1d(tree, beam.v[0].node, labels, path);
    if (n_nodes_out) *n_nodes_out = tree->len;
    ws->beam = beam;
    ws->next = next;
    ws->tmp = tmp;
    ws->tmpcap = tmpcap;
    return status;
}

/* ndarray-stats QuantileExt::argmax / max (Cargo.toml:12): first maximum wins, NaN -> Err ->
 * unwrap() -> panic (src/search.rs:56,58) */
static int argmax_strided(const float *v, int64_t n, int64_t s, int64_t *arg, float *mx) {
    if (n <= 0) return FCDO_PANIC;
    int64_t a = 0;
    float m = v[0];
    if (m != m) return FCDO_PANIC;
    for (int64_t i = 1; i < n; ++i) {
        float e = v[i * s];
        if (e != e) return FCDO_PANIC;
        if (e > m) {
            m = e;
            a = i;
        }
    }
    *arg = a;
    *mx = m;
    return FCDO_OK;
}

int fcdo_crf_beam_search(const float *x, int64_t T, int64_t S, int64_t N,
                         int64_t s0, int64_t s1, int64_t s2,
                         const float *init, int64_t n_init, int64_t is0,
                         int64_t beam_size, float thr,
                         int32_t *labels, int64_t *path, int64_t *n_out) {
    return fcdo_crf_beam_search_ex(x, T, S, N, s0, s1, s2, init, n_init, is0, beam_size, thr, labels, path,
                                   n_out, NULL);
}

int fcdo_crf_beam_search_ex(const float *x, int64_t T, int64_t S, int64_t N,
                            int64_t s0, int64_t s1, int64_t s2,
                            const float *init, int64_t n_init, int64_t is0,
                            int64_t beam_size, float thr,
                            int32_t *labels, int64_t *path, int64_t *n_out, int64_t *n_ambiguous_out) {
    if (n_ambiguous_out) n_ambiguous_out[0] = n_ambiguous_out[1] = 0;
    if (T <= 0 || S <= 0 || N <= 0) return FCDO_PANIC; /* :46 */
    int64_t n_state = S, n_base = N - 1;               /* :50-51 */
    int64_t st0;
    float mx;
    if (argmax_strided(init, n_init, is0, &st0, &mx) != FCDO_OK) return FCDO_PANIC;
    fcdo_tree *tree = fcdo_tree_new(n_base);
    sp1vec beam = {0}, next = {0};
    sp1 *tmp = NULL;
    int64_t tmpcap = 0;
    int status = FCDO_OK;
    sp1 root = {-1, st0, mx, init[0]}; /* :54-59 */
    sp1_push(&beam, root);

    for (int64_t idx = 0; idx < T && status == FCDO_OK; ++idx) { /* :62 */
        next.len = 0;
        for (int64_t bi = 0; bi < beam.len; ++bi) {
            sp1 b = beam.v[bi];
            if (b.state < 0 || b.state >= n_state) { /* ndarray OOB panic :72 */
                status = FCDO_PANIC;
                break;
            }
            const float *pr = x + idx * s0 + b.state * s1;
            if (pr[0] > thr) { /* :75-82 */
                sp1 c = {b.node, b.state, 0.0f, (b.label_prob + b.gap_prob) * pr[0]};
                sp1_push(&next, c);
            }
            for (int64_t label = 0; label < n_base; ++label) { /* :84 */
                float pr_b = pr[(label + 1) * s2];
                if (pr_b < thr) continue;
                int32_t child = fcdo_tree_get_child(tree, b.node, label);
                if (child < 0) child = fcdo_tree_add_node(tree, b.node, label, idx);
                sp1 c = {child, (b.state * n_base) % n_state + label, /* :97 */
                         (b.label_prob + b.gap_prob) * pr_b, 0.0f};
                sp1_push(&next, c);
            }
        }
        if (status != FCDO_OK) break;
        sp1vec t = beam;
        beam = next;
        next = t;
        status = sp1_merge_prune(&beam, beam_size, &tmp, &tmpcap, n_ambiguous_out, NULL, 0); /* :104-142 */
    }
    if (status == FCDO_OK) *n_out = tree_walk_1d(tree, beam.v[0].node, labels, path);
    free(beam.v);
    free(next.v);
    free(tmp);
    fcdo_tree_free(tree);
    return status;
}

/* src/search.rs:385-423 */
int fcdo_crf_greedy_search(const float *x, int64_t T, int64_t S, int64_t N,
                           int64_t s0, int64_t s1, int64_t s2,
                           const float *init, int64_t n_init, int64_t is0,
                           float qscale, float qbias,
                           int32_t *labels, int64_t *path, uint32_t *quals, int64_t *n_out) {
    if (T <= 0 || S <= 0 || N <= 0) return FCDO_PANIC;
    int64_t n_state = S, n_base = N - 1;
    int64_t state;
    float mx;
    if (argmax_strided(init, n_init, is0, &state, &mx) != FCDO_OK) return FCDO_PANIC;
    int64_t n = 0;
    for (int64_t idx = 0; idx < T; ++idx) {
        if (state < 0 || state >= n_state) return FCDO_PANIC;
        const float *pr = x + idx * s0 + state * s1;
        int64_t label;
        float prob;
        if (argmax_strided(pr, N, s2, &label, &prob) != FCDO_OK) return FCDO_PANIC;
        if (label > 0) {
            path[n] = idx;
            labels[n] = (int32_t)label;
            if (quals) quals[n] = phred_code(prob, qscale, qbias);
            n++;
            state = (state * n_base) % n_state + (label - 1); /* :415 */
        }
    }
    *n_out = n;
    return FCDO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Duplex (2D) search: src/duplex.rs
 * ---------------------------------------------------------------------------------------- */
static const float NEG_INF = -INFINITY;

/* LogSpace::add, src/duplex.rs:42-63.  mode MAX reproduces the `fastexp` default feature,
 * whose exp() returns 0.0 for every input on little-endian targets (src/fastexp.rs:33-58:
 * the f32 view of the i64 union reads the low 32 bits of `i << 52`), so
 * big + ln_1p(0.0) == big. */
/*
 * libm flavour.  Rust's f32::ln / exp / ln_1p call the platform libm (logf / expf / log1pf), whose
 * last-bit behaviour differs between glibc versions (2.35's logf is < 0.82 ULP, log1pf < 1 ULP;
 * glibc >= 2.41 ships correctly rounded CORE-MATH versions).  FCDO_MATH_CR selects correctly
 * rounded f32 results (evaluated in x87 long double, double-rounding risk ~2^-40): a
 * platform-independent definition the HIP kernels reproduce (they evaluate in f64).  Without the
 * flag the host libm is used, exactly as the reference would on this machine.
 */
static inline float ln_m(float x, int mode) {
    return (mode & FCDO_MATH_CR) ? (float)logl((long double)x) : logf(x);
}
static inline float exp_m(float x, int mode) {
    return (mode & FCDO_MATH_CR) ? (float)expl((long double)x) : expf(x);
}
static inline float log1p_m(float x, int mode) {
    return (mode & FCDO_MATH_CR) ? (float)log1pl((long double)x) : log1pf(x);
}

/* instrumentation for the duplex roofline (tools/bench_configs.py): LogSpace::add calls of this thread */
static __thread int64_t g_logadd_calls = 0;
static int64_t g_duplex_ties[4];
/* the per-search twin of fcd_result.ambiguous (include/fcd.h) for the duplex searches, of the calling thread's
 * most recent search: [0] pruning steps with more than 20 candidates in which a kept candidate has exactly the
 * probability of another candidate, [1] steps with equal probabilities at ranks 0 / 1 or across the truncation
 * boundary */
static __thread int64_t t_duplex_amb[2];
void fcdo_duplex_last_ambiguous(int64_t out[2]) {
    out[0] = t_duplex_amb[0];
    out[1] = t_duplex_amb[1];
}
/* test/analysis instrument (not thread-safe): {pruning steps, steps with > 20 candidates in which a kept candidate
 * ties with another, steps with a tie across the truncation boundary, reads whose final top two tie} summed
 * over the duplex searches run since the last reset */
void fcdo_duplex_tie_steps(int64_t out[4], int reset) {
    for (int i = 0; i < 4; ++i) {
        if (out) out[i] = g_duplex_ties[i];
        if (reset) g_duplex_ties[i] = 0;
    }
}

int64_t fcdo_logadd_calls(int reset) {
    int64_t n = g_logadd_calls;
    if (reset) g_logadd_calls = 0;
    return n;
}

float fcdo_logspace_add(float a, float b, int mode) {
    ++g_logadd_calls;
    float big, small;
    if (a <= b) {
        big = b;
        small = a;
    } else {
        big = a;
        small = b;
    }
    if (small == NEG_INF) return big;
    if ((mode & 3) == FCDO_LOGADD_MAX) return big + 0.0f; /* big + ln_1p(+-0.0) */
    return big + log1p_m(exp_m(small - big, mode), mode);
}
/* (test hook) out[i] = f(x[i]) with the C library's binary32 routine: 0 expf, 1 logf, 2 log1pf -- what the reference's
 * f32::exp / ln / ln_1p call (src/duplex.rs:17,25,50) on the machine this runs on */
void fcdo_libm_apply(int which, const float *x, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = which == 0 ? expf(x[i]) : (which == 1 ? logf(x[i]) : log1pf(x[i]));
}

void fcdo_logspace_add_batch(const float *a, const float *b, float *out, int64_t n, int mode) {
    for (int64_t i = 0; i < n; ++i) out[i] = fcdo_logspace_add(a[i], b[i], mode);
}

#define LADD(a, b) fcdo_logspace_add((a), (b), mode)
static inline float lmax(float self, float other) { return (self < other) ? other : self; } /* :33-39 */

typedef struct {
    float label, gap;
} ppair; /* ProbPair :82-126 */

static const ppair PP_ZERO = {-INFINITY, -INFINITY};

typedef struct { /* SecondaryProbs :152-210 */
    int64_t offset;
    ppair *probs;
    int64_t len, cap;
    float max_prob;
} secprobs;

static void sec_push(secprobs *s, ppair p) {
    if (s->len == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 16;
        s->probs = (ppair *)realloc(s->probs, sizeof(ppair) * s->cap);
    }
    s->probs[s->len++] = p;
}

static ppair sec_get(const secprobs *s, int64_t at) { /* :167-179 */
    int64_t index = at - s->offset;
    if (index < 0 || index >= s->len) return PP_ZERO;
    return s->probs[index];
}

static void sec_discard_until(secprobs *s, int64_t keep_from) { /* :181-191 */
    if (keep_from > s->offset) {
        int64_t first = keep_from - s->offset;
        if (first < s->len) {
            memmove(s->probs, s->probs + first, sizeof(ppair) * (s->len - first));
            s->len -= first;
        } else {
            s->len = 0;
        }
        s->offset = keep_from;
    }
}

static int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static float pairs_update_max(const ppair *probs, int64_t len, int64_t offset, int64_t lower,
                              int64_t upper, int mode) { /* :193-204 */
    /* (lower - offset) can overflow for the isize::MIN/MAX KAT; do it in saturating form */
    int64_t begin, end;
    if (lower <= offset) begin = 0;
    else begin = clampi(lower - offset, 0, len);
    if (upper <= offset) end = begin;
    else {
        /* upper - offset may overflow when offset < 0 and upper == INT64_MAX */
        int64_t d = (offset < 0 && upper > INT64_MAX + offset) ? INT64_MAX : upper - offset;
        end = clampi(d, begin, len);
    }
    float m = NEG_INF;
    for (int64_t i = begin; i < end; ++i) m = lmax(m, LADD(probs[i].label, probs[i].gap));
    return m;
}

/* white-box hooks for KAT K18 */
void fcdo_secondary_get(const float *pairs, int64_t len, int64_t offset, int64_t at,
                        float *label_out, float *gap_out) {
    secprobs s = {offset, (ppair *)pairs, len, len, NEG_INF};
    ppair p = sec_get(&s, at);
    *label_out = p.label;
    *gap_out = p.gap;
}
float fcdo_secondary_update_max(const float *pairs, int64_t len, int64_t offset, int64_t lower,
                                int64_t upper, int mode) {
    return pairs_update_max((const ppair *)pairs, len, offset, lower, upper, mode);
}

typedef struct {
    const float *x; /* log-space copy, C-contiguous */
    int64_t T, S, N;
} lognet;

static inline const float *lognet_row(const lognet *n, int64_t t, int64_t state) {
    return n->x + (t * n->S + state) * n->N;
}

/* One recurrence step shared by build/extend (:233-246, :373-385 and crf :275-287,:320-334). */
static inline ppair sec_step(const float *row, ppair last, ppair prev_parent, int64_t label,
                             int is_repeat, int mode) {
    ppair r;
    r.gap = LADD(last.label, last.gap) + row[0];
    if (is_repeat)
        r.label = row[label + 1] + LADD(last.label, prev_parent.gap);
    else
        r.label = row[label + 1] + LADD(last.label, LADD(prev_parent.label, prev_parent.gap));
    return r;
}

/* build_secondary_probs :212-249 / crf_build_secondary_probs :251-291 (is_repeat = 0) */
static secprobs sec_build(const lognet *n2, const secprobs *parent, int64_t label, int is_repeat,
                          int64_t tstate, int64_t lower, int64_t upper, int mode) {
    secprobs s = {lower, NULL, 0, 0, NEG_INF};
    s.cap = upper - lower;
    s.probs = (ppair *)malloc(sizeof(ppair) * (s.cap > 0 ? s.cap : 1));
    ppair last = PP_ZERO;
    for (int64_t idx = lower; idx < upper; ++idx) {
        last = sec_step(lognet_row(n2, idx, tstate), last, sec_get(parent, idx - 1), label,
                        is_repeat, mode);
        sec_push(&s, last);
        s.max_prob = lmax(s.max_prob, LADD(last.label, last.gap));
    }
    return s;
}

/* extend_secondary_probs :338-387 / crf_extend_secondary_probs :293-336 */
/* Returns 0, or -1 where the reference panics: `assert!(current_end < upper_bound)` (:363-366, :311-314) fails
 * for a beam entry whose window already reaches the new bound -- possible only after the envelope's upper bound
 * moved BACK (last_upper_bound is the previous row's bound, not the largest seen, :524) and forward again by less.
 * (Found by tests/test_naive_crosscheck.py: rounds 1 and 2 of this restatement skipped the extension silently.) */
static int sec_extend(secprobs *s, const lognet *n2, const secprobs *parent, int64_t label,
                      int is_repeat, int64_t tstate, int64_t lower, int64_t upper, int mode) {
    if (lower > s->offset) { /* :351-359 */
        sec_discard_until(s, lower - 1);
        if (s->len == 0) s->offset = lower;
        s->max_prob = pairs_update_max(s->probs, s->len, s->offset, lower, upper, mode);
    }
    int64_t current_end = s->offset + s->len;
    if (current_end >= upper) return -1;
    ppair last = s->len > 0 ? s->probs[s->len - 1] : PP_ZERO;
    for (int64_t idx = current_end; idx < upper; ++idx) {
        last = sec_step(lognet_row(n2, idx, tstate), last, sec_get(parent, idx - 1), label,
                        is_repeat, mode);
        sec_push(s, last);
        s->max_prob = lmax(s->max_prob, LADD(last.label, last.gap));
    }
    return 0;
}

typedef struct { /* duplex SearchPoint :128-150 */
    int32_t node;
    int64_t state;
    ppair prob_1;
    float prob_2_max;
} sp2;

typedef struct {
    sp2 *v;
    int64_t len, cap;
} sp2vec;

static void sp2_push(sp2vec *b, sp2 p) {
    if (b->len == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        b->v = (sp2 *)realloc(b->v, sizeof(sp2) * b->cap);
    }
    b->v[b->len++] = p;
}

typedef struct {
    sp2 p;
    float prob; /* cached probability() for the final sort */
} sp2k;

#define SP2_NODE_LESS(a, b) ((a)->node < (b)->node)
#define SP2K_PROB_GREATER(a, b) ((a)->prob > (b)->prob)
DEFINE_STABLE_SORT(sp2_sort_node, sp2, SP2_NODE_LESS)
DEFINE_STABLE_SORT(sp2k_sort_prob, sp2k, SP2K_PROB_GREATER)
DEFINE_PDQSORT(sp2k_pdq_prob, sp2k, SP2K_PROB_GREATER)

typedef struct {
    secprobs *v;
    int64_t len, cap;
} secvec;

static void secvec_push(secvec *d, secprobs s) {
    if (d->len == d->cap) {
        d->cap = d->cap ? d->cap * 2 : 256;
        d->v = (secprobs *)realloc(d->v, sizeof(secprobs) * d->cap);
    }
    d->v[d->len++] = s;
}

static float *to_logspace(const float *x, int64_t T, int64_t S, int64_t N, int64_t s0, int64_t s1,
                          int64_t s2, int mode) { /* LogSpace::new = ln :24-26, :452-453 */
    float *o = (float *)malloc(sizeof(float) * (size_t)(T * S * N > 0 ? T * S * N : 1));
    for (int64_t t = 0; t < T; ++t)
        for (int64_t s = 0; s < S; ++s)
            for (int64_t j = 0; j < N; ++j)
                o[(t * S + s) * N + j] = ln_m(x[t * s0 + s * s1 + j * s2], mode);
    return o;
}

/*
 * Common driver for duplex::beam_search (:443-650, crf = 0) and duplex::crf_beam_search
 * (:652-834, crf = 1).
 */
static int duplex_core(const lognet *n1, const lognet *n2, int crf, int64_t init_state_1,
                       int64_t init_state_2, const uint64_t *envelope, int64_t e0, int64_t e1,
                       int64_t beam_size, float thr_real, int collapse_repeats, int mode,
                       int32_t *labels, int64_t *n_out) {
    const int64_t N = n1->N, n_base = N - 1, n_state = n1->S;
    const float thr = ln_m(thr_real, mode); /* :454 */
    int status = FCDO_OK;
    t_duplex_amb[0] = t_duplex_amb[1] = 0;
    if (n1->T <= 0) return FCDO_PANIC; /* envelope[(0,1)] out of bounds :477 */

    fcdo_tree *tree = fcdo_tree_new(n_base);
    secvec data = {0}; /* node payloads, index == node index */
    sp2vec beam = {0}, next = {0};
    sp2 *tmp = NULL;
    int64_t tmpcap = 0;
    sp2k *keyed = NULL, *ktmp = NULL;
    int64_t keyedcap = 0, ktmpcap = 0;

    sp2 root = {-1, init_state_1, {NEG_INF, 0.0f}, 0.0f}; /* :464-473 */
    sp2_push(&beam, root);

    /* root_probs :389-409 / crf_root_probs :411-441 */
    secprobs rootp = {-1, NULL, 0, 0, 0.0f};
    {
        uint64_t ub = envelope[0 * e0 + 1 * e1];
        if (ub > (uint64_t)n2->T) { /* slice(s![..upper_bound]) panics */
            status = FCDO_PANIC;
            goto done;
        }
        float cur = 0.0f;
        ppair p = {NEG_INF, cur};
        sec_push(&rootp, p);
        int64_t state = init_state_2;
        for (uint64_t t = 0; t < ub; ++t) {
            if (state < 0 || state >= n2->S) {
                status = FCDO_PANIC;
                goto done;
            }
            cur = cur + lognet_row(n2, (int64_t)t, state)[0];
            ppair q = {NEG_INF, cur};
            sec_push(&rootp, q);
            if (crf) state = (state * n_base) % n_state; /* :437 */
        }
    }

    int64_t last_upper = 0; /* :480 */
    for (int64_t t1 = 0; t1 < n1->T; ++t1) {
        next.len = 0;
        uint64_t lo_u = envelope[t1 * e0], hi_u = envelope[t1 * e0 + e1];
        int64_t upper_t = hi_u > (uint64_t)n2->T ? n2->T : (int64_t)hi_u; /* :485 */
        if (lo_u >= (uint64_t)upper_t || lo_u > (uint64_t)last_upper) {   /* :486-488 */
            status = FCDO_INVALID_ENVELOPE;
            break;
        }
        int64_t lower_t = (int64_t)lo_u;

        if (upper_t > last_upper) { /* :490-522 */
            sp2_sort_node(beam.v, beam.len, &tmp, &tmpcap); /* parents before children :493 */
            for (int64_t bi = 0; bi < beam.len; ++bi) {
                int32_t node = beam.v[bi].node;
                if (node < 0) continue;
                int32_t par = tree->parent[node];
                int64_t lab = tree->label[node];
                const secprobs *pp = par >= 0 ? &data.v[par] : &rootp;
                int is_repeat = 0;
                int64_t tstate = 0;
                if (crf) {
                    tstate = beam.v[bi].state; /* :708,:725-728 */
                    if (tstate < 0 || tstate >= n2->S) {
                        status = FCDO_PANIC;
                        break;
                    }
                } else {
                    is_repeat = (par >= 0 && tree->label[par] == lab); /* :512 */
                }
                if (sec_extend(&data.v[node], n2, pp, lab, is_repeat, tstate, lower_t, upper_t, mode) != 0) {
                    status = FCDO_PANIC;
                    break;
                }
            }
            if (status != FCDO_OK) break;
        }
        last_upper = upper_t; /* :524 */

        for (int64_t bi = 0; bi < beam.len && status == FCDO_OK; ++bi) { /* :526 */
            sp2 tip = beam.v[bi];
            int64_t tip_label = fcdo_tree_label(tree, tip.node);
            const float *row;
            if (crf) {
                if (tip.state < 0 || tip.state >= n1->S) {
                    status = FCDO_PANIC;
                    break;
                }
                row = lognet_row(n1, t1, tip.state); /* :749 */
            } else {
                row = lognet_row(n1, t1, 0);
            }
            float tip_total = LADD(tip.prob_1.label, tip.prob_1.gap);
            if (row[0] > thr) { /* :529-534 */
                sp2 c = tip;
                c.prob_1.label = NEG_INF;
                c.prob_1.gap = tip_total + row[0];
                sp2_push(&next, c);
            }
            for (int64_t label = 0; label < n_base; ++label) { /* :536 */
                float prob = row[label + 1];
                if (prob < thr) continue;
                const secprobs *pp = tip.node >= 0 ? &data.v[tip.node] : &rootp;
                if (!crf && collapse_repeats && label == tip_label) { /* :540-570 */
                    sp2 stay = tip;
                    stay.prob_1.label = tip.prob_1.label + prob;
                    stay.prob_1.gap = NEG_INF;
                    sp2_push(&next, stay);
                    int32_t child = fcdo_tree_get_child(tree, tip.node, label);
                    if (child < 0 && tip.prob_1.gap > NEG_INF) { /* :546 */
                        secprobs s = sec_build(n2, pp, label, 1, 0, lower_t, upper_t, mode);
                        child = fcdo_tree_add_node(tree, tip.node, label, 0);
                        secvec_push(&data, s);
                    }
                    if (child >= 0) {
                        sp2 c = tip;
                        c.node = child;
                        c.prob_1.label = tip.prob_1.gap + prob;
                        c.prob_1.gap = NEG_INF;
                        sp2_push(&next, c);
                    }
                } else { /* :571-592 / crf :764-786 */
                    int32_t child = fcdo_tree_get_child(tree, tip.node, label);
                    if (child < 0) {
                        int64_t tstate = 0;
                        if (crf) {
                            tstate = tip.state; /* :772 */
                            if (tstate >= n2->S) {
                                status = FCDO_PANIC;
                                break;
                            }
                        }
                        secprobs s = sec_build(n2, pp, label, 0, tstate, lower_t, upper_t, mode);
                        child = fcdo_tree_add_node(tree, tip.node, label, 0);
                        secvec_push(&data, s);
                        pp = tip.node >= 0 ? &data.v[tip.node] : &rootp; /* data.v may move */
                    }
                    sp2 c = tip;
                    c.node = child;
                    if (crf) c.state = (tip.state * n_base) % n_state + label; /* :782 */
                    c.prob_1.label = tip_total + prob;
                    c.prob_1.gap = NEG_INF;
                    sp2_push(&next, c);
                }
            }
        }
        if (status != FCDO_OK) break;

        sp2vec sw = beam; /* :595 */
        beam = next;
        next = sw;

        sp2_sort_node(beam.v, beam.len, &tmp, &tmpcap); /* :598 */
        int64_t w = 0;
        for (int64_t i = 0; i < beam.len; ++i) { /* :599-612 */
            if (w > 0 && beam.v[w - 1].node == beam.v[i].node) {
                beam.v[w - 1].prob_1.label = LADD(beam.v[w - 1].prob_1.label, beam.v[i].prob_1.label);
                beam.v[w - 1].prob_1.gap = LADD(beam.v[w - 1].prob_1.gap, beam.v[i].prob_1.gap);
            } else {
                beam.v[w++] = beam.v[i];
            }
        }
        beam.len = w;
        if (keyedcap < beam.len) {
            keyedcap = beam.len * 2;
            keyed = (sp2k *)realloc(keyed, sizeof(sp2k) * keyedcap);
        }
        int has_nan = 0;
        for (int64_t i = 0; i < beam.len; ++i) { /* :613-618 */
            if (beam.v[i].node >= 0) beam.v[i].prob_2_max = data.v[beam.v[i].node].max_prob;
            keyed[i].p = beam.v[i];
            keyed[i].prob = LADD(beam.v[i].prob_1.label, beam.v[i].prob_1.gap) + beam.v[i].prob_2_max;
            if (keyed[i].prob != keyed[i].prob) has_nan = 1;
        }
        if (beam.len >= 2 && has_nan) { /* :619-631 */
            status = FCDO_INCOMPARABLE;
            break;
        }
        sp2k *pdq_copy = NULL;
        if (g_unstable_sort_mode == 1 && beam.len > 20) { /* see sp1_merge_prune */
            pdq_copy = (sp2k *)malloc(sizeof(sp2k) * beam.len);
            memcpy(pdq_copy, keyed, sizeof(sp2k) * beam.len);
            sp2k_pdq_prob(pdq_copy, beam.len);
        }
        sp2k_sort_prob(keyed, beam.len, &ktmp, &ktmpcap);
        { /* tie statistics (fcdo_duplex_tie_steps): :620 / :807 is sort_unstable_by, i.e. the order of EQUAL
           * probabilities is Rust's pdqsort's above 20 candidates; this restatement keeps node order */
            const int64_t n = beam.len, kept = n < beam_size ? n : beam_size;
            int tie = 0, boundary = 0;
            for (int64_t i = 0; i + 1 < n; ++i) {
                if (keyed[i].prob != keyed[i + 1].prob) continue;
                if (i < kept) tie = 1;
                if (i + 1 == kept) boundary = 1;
            }
            if (n > 20 && tie) t_duplex_amb[0] += 1;
            if (boundary || (n >= 2 && keyed[0].prob == keyed[1].prob)) t_duplex_amb[1] += 1;
            g_duplex_ties[0] += 1;                                  /* pruning steps */
            if (n > 20 && tie) g_duplex_ties[1] += 1;               /* > 20 candidates and a kept one tied */
            if (boundary) g_duplex_ties[2] += 1;                    /* a tie across the truncation boundary */
            if (t1 + 1 == n1->T && n >= 2 && keyed[0].prob == keyed[1].prob) g_duplex_ties[3] += 1; /* a tie for the answer */
        }
        if (pdq_copy) {
            memcpy(keyed, pdq_copy, sizeof(sp2k) * beam.len);
            free(pdq_copy);
        }
        if (beam.len > beam_size) beam.len = beam_size; /* :632 */
        if (beam.len == 0) {                            /* :633-636 */
            status = FCDO_RAN_OUT_OF_BEAM;
            break;
        }
        for (int64_t i = 0; i < beam.len; ++i) beam.v[i] = keyed[i].p;
    }

    if (status == FCDO_OK) *n_out = tree_walk_1d(tree, beam.v[0].node, labels, NULL); /* :638-649 */

done:
    for (int64_t i = 0; i < data.len; ++i) free(data.v[i].probs);
    free(data.v);
    free(rootp.probs);
    free(beam.v);
    free(next.v);
    free(tmp);
    free(keyed);
    free(ktmp);
    fcdo_tree_free(tree);
    return status;
}

int fcdo_beam_search_duplex(const float *x1, int64_t T1, int64_t rs1, int64_t cs1,
                            const float *x2, int64_t T2, int64_t rs2, int64_t cs2,
                            int64_t N, const uint64_t *envelope, int64_t e0, int64_t e1,
                            int64_t beam_size, float thr, int collapse_repeats,
                            int logadd_mode, int32_t *labels, int64_t *n_out) {
    float *l1 = to_logspace(x1, T1, 1, N, rs1, 0, cs1, logadd_mode);
    float *l2 = to_logspace(x2, T2, 1, N, rs2, 0, cs2, logadd_mode);
    lognet n1 = {l1, T1, 1, N}, n2 = {l2, T2, 1, N};
    int st = duplex_core(&n1, &n2, 0, 0, 0, envelope, e0, e1, beam_size, thr, collapse_repeats,
                         logadd_mode, labels, n_out);
    free(l1);
    free(l2);
    return st;
}

int fcdo_crf_beam_search_duplex(const float *x1, int64_t T1, const int64_t *st1,
                                const float *init1, int64_t n_init1, int64_t i1s,
                                const float *x2, int64_t T2, const int64_t *st2,
                                const float *init2, int64_t n_init2, int64_t i2s,
                                int64_t S, int64_t N,
                                const uint64_t *envelope, int64_t e0, int64_t e1,
                                int64_t beam_size, float thr,
                                int logadd_mode, int32_t *labels, int64_t *n_out) {
    int64_t a1, a2;
    float m;
    if (argmax_strided(init1, n_init1, i1s, &a1, &m) != FCDO_OK) return FCDO_PANIC; /* :679 */
    if (argmax_strided(init2, n_init2, i2s, &a2, &m) != FCDO_OK) return FCDO_PANIC; /* :691 */
    float *l1 = to_logspace(x1, T1, S, N, st1[0], st1[1], st1[2], logadd_mode);
    float *l2 = to_logspace(x2, T2, S, N, st2[0], st2[1], st2[2], logadd_mode);
    lognet n1 = {l1, T1, S, N}, n2 = {l2, T2, S, N};
    int st = duplex_core(&n1, &n2, 1, a1, a2, envelope, e0, e1, beam_size, thr, 0, logadd_mode,
                         labels, n_out);
    free(l1);
    free(l2);
    return st;
}

/* ------------------------------------------------------------------------------------------
 * Batch drivers (CPU baseline + differential tests).  One read per task, pthreads.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float *x;
    int64_t n_tasks; /* n_reads * passes: task k decodes read k % n_reads; only pass 0 is stored */
    int64_t n_reads, T, N, beam_size;
    float thr;
    int collapse, kind; /* kind 0 = beam, 1 = viterbi */
    int32_t *labels;
    int64_t *path, *lens;
    int32_t *status;
    int64_t *ambiguous; /* nullable: per-read count of unpinned tie steps (sp1_merge_prune) */
    volatile int64_t *next;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    beam_ws ws;
    memset(&ws, 0, sizeof(ws));
    /* worst-case tree up front: T * beam * labels nodes (search.rs:200-239) */
    ws.tree = tree_new_cap(j->N - 1, j->kind == 0 ? j->T * j->beam_size * (j->N - 1) + 16 : 16);
    int32_t *sl = (int32_t *)malloc(sizeof(int32_t) * (j->T > 0 ? j->T : 1));
    int64_t *sp = (int64_t *)malloc(sizeof(int64_t) * (j->T > 0 ? j->T : 1));
    for (;;) {
        int64_t k = __sync_fetch_and_add(j->next, 1);
        if (k >= j->n_tasks) break;
        const int64_t r = k % j->n_reads;
        const int store = k < j->n_reads;
        const float *x = j->x + r * j->T * j->N;
        int32_t *ol = store ? j->labels + r * j->T : sl;
        int64_t *op = store ? j->path + r * j->T : sp;
        int64_t n = 0, amb[2] = {0, 0};
        int st;
        if (j->kind == 0)
            st = beam_search_ws(&ws, x, j->T, j->N, j->N, 1, j->beam_size, j->thr, j->collapse, ol,
                                op, &n, NULL, (store && j->ambiguous) ? amb : NULL, NULL);
        else
            st = fcdo_viterbi_search(x, j->T, j->N, j->N, 1, j->collapse, 1.0f, 0.0f, ol, op, NULL, &n);
        if (store) {
            j->lens[r] = (st == FCDO_OK) ? n : 0;
            if (j->status) j->status[r] = st;
            if (j->ambiguous) {
                j->ambiguous[2 * r] = amb[0];
                j->ambiguous[2 * r + 1] = amb[1];
            }
        }
    }
    free(sl);
    free(sp);
    free(ws.beam.v);
    free(ws.next.v);
    free(ws.tmp);
    fcdo_tree_free(ws.tree);
    return NULL;
}

static int run_batch(batch_job *job, int n_threads) {
    volatile int64_t next = 0;
    job->next = &next;
    if (n_threads <= 1) {
        batch_worker(job);
        return 0;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, batch_worker, job);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
    return 0;
}

int fcdo_beam_search_batch(const float *x, int64_t n_reads, int64_t T, int64_t N,
                           int64_t beam_size, float thr, int collapse,
                           int32_t *labels, int64_t *path, int64_t *lens, int32_t *status,
                           int n_threads, int64_t n_passes) {
    return fcdo_beam_search_batch_ex(x, n_reads, T, N, beam_size, thr, collapse, labels, path, lens, status,
                                     NULL, n_threads, n_passes);
}

int fcdo_beam_search_batch_ex(const float *x, int64_t n_reads, int64_t T, int64_t N,
                              int64_t beam_size, float thr, int collapse,
                              int32_t *labels, int64_t *path, int64_t *lens, int32_t *status,
                              int64_t *ambiguous, int n_threads, int64_t n_passes) {
    if (n_passes < 1) n_passes = 1;
    batch_job job = {x, n_reads * n_passes, n_reads, T, N, beam_size, thr, collapse, 0,
                     labels, path, lens, status, ambiguous, NULL};
    return run_batch(&job, n_threads);
}

int fcdo_viterbi_batch(const float *x, int64_t n_reads, int64_t T, int64_t N, int collapse,
                       int32_t *labels, int64_t *path, int64_t *lens, int n_threads) {
    batch_job job = {x, n_reads, n_reads, T, N, 0, 0.0f, collapse, 1, labels, path, lens, NULL, NULL, NULL};
    return run_batch(&job, n_threads);
}
