/*
 * fcd_oracle.h -- CPU restatement of nanoporetech/fast-ctc-decode's search algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fast_ctc_decode_amd/, include/)
 * may include, link, dlopen or execute this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it -- as the checker / the reported CPU baseline.
 *
 * The reference is Rust (PyO3 cdylib) and cannot be built in this environment (no rustc /
 * cargo / maturin, no network), so this is a from-scratch plain-C restatement written from the
 * reference's behaviour.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).  The restatement is pinned by the reference's own
 * known-answer tests (tests/test_oracle_kat.py runs every KAT listed in SURVEY.md section 4).
 *
 * Parity status:
 *   - viterbi / beam_search / crf_beam_search / crf_greedy: pinned by KATs K1-K13, K16.
 *   - tie order of Rust's sort_unstable_by for > 20 candidates (pdqsort): restated (fcd_oracle.c,
 *     DEFINE_PDQSORT; the default since round 4) and, since round 5, pinned against a rustc-1.65 build
 *     of std found compiled in this image (tools/verify/rust165_pdqsort.py) -- all of it except the
 *     later (2023) forms of break_patterns' generator and partial_insertion_sort's shifting, which are
 *     Rust 1.78 AS RECALLED (tools/verify/pdq178_check.rs asks a 1.78 toolchain).  Up to 20 candidates
 *     the sort is an insertion sort: ties keep ascending node order (the list is first stably sorted
 *     by node); that rule for every length stays selectable (fcdo_set_unstable_sort(0)).
 *   - duplex: pinned by K14, K15, K16, K18; libm bit-level results (logf/expf/log1pf)
 *     are PARITY UNPINNED beyond those KATs.
 *
 * All matrices are f32 with explicit element strides (in elements, not bytes), so any
 * numpy view can be passed without a copy, like the reference's zero-copy ndarray views
 * (src/lib.rs:198,352).
 */
#ifndef FCD_ORACLE_H
#define FCD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors SearchError (src/lib.rs:36-41); 0 = Ok. */
enum {
    FCDO_OK = 0,
    FCDO_RAN_OUT_OF_BEAM = 1,    /* "Ran out of search space (beam_cut_threshold too high)" */
    FCDO_INCOMPARABLE = 2,       /* "Failed to compare values (NaNs in input?)" */
    FCDO_INVALID_ENVELOPE = 3    /* "Invalid envelope values" */
};

/* duplex log-add modes (SURVEY.md section 0 finding 3) */
enum {
    FCDO_LOGADD_LOGSUMEXP = 0,   /* cargo --no-default-features: libm expf/log1pf */
    FCDO_LOGADD_MAX = 1,         /* default-feature wheels: fastexp() == 0.0 => max() */
    FCDO_MATH_CR = 4             /* OR-able: correctly rounded ln/exp/ln_1p instead of the host libm */
};

/* src/search.rs:31-36 */
char fcdo_phred(float prob, float qscale, float qbias);

/*
 * src/search.rs:320-383.  labels[] receives alphabet indices (1..N-1), path[] the row index
 * of each emission, quals[] (nullable) one phred code point per emitted label (Rust char::from_u32).
 * Output arrays need room for T entries.  Returns FCDO_OK.
 */
int fcdo_viterbi_search(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                        int collapse_repeats, float qscale, float qbias,
                        int32_t *labels, int64_t *path, uint32_t *quals, int64_t *n_out);

/*
 * src/search.rs:159-301.  labels[] receives alphabet indices (label+1), i.e. index into the
 * caller's alphabet; path[] the node-creation times.  Room for T entries each.
 * n_nodes_out (nullable) receives the size of the suffix tree (instrumentation only).
 */
int fcdo_beam_search(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                     int64_t beam_size, float beam_cut_threshold, int collapse_repeats,
                     int32_t *labels, int64_t *path, int64_t *n_out, int64_t *n_nodes_out);

/*
 * The same search plus the tie instrument of SURVEY.md 8a A4.  Rust 1.78's sort_unstable_by
 * (src/search.rs:122,262) is a stable insertion sort up to 20 elements and pdqsort -- implementation-defined
 * tie order -- above; this restatement breaks ties by ascending node index (the stable behaviour).
 * n_ambiguous_out (nullable, TWO entries) receives
 *   [0] the number of steps with MORE than 20 merged candidates in which a candidate that survives the
 *       truncation has exactly the probability of another candidate.  0 => the beam, set and order, follows
 *       the reference step for step (every other step is pinned by the stable rule);
 *   [1] the number of steps (any candidate count) with equal probabilities at ranks 0 / 1 or across the
 *       truncation boundary.  0 => kept sets and best entry do not depend on ANY tie rule.
 * A read with either entry 0 is pinned to the reference; the rest can be settled exhaustively with
 * fcdo_beam_search_all_tie_orders below.
 */
int fcdo_beam_search_ex(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                        int64_t beam_size, float beam_cut_threshold, int collapse_repeats,
                        int32_t *labels, int64_t *path, int64_t *n_out, int64_t *n_nodes_out,
                        int64_t *n_ambiguous_out);

/*
 * src/search.rs:38-157.  x is (T,S,N) with strides (s0,s1,s2); init is (>=S,) stride is0,
 * n_init its length.
 */
int fcdo_crf_beam_search(const float *x, int64_t T, int64_t S, int64_t N,
                         int64_t s0, int64_t s1, int64_t s2,
                         const float *init, int64_t n_init, int64_t is0,
                         int64_t beam_size, float beam_cut_threshold,
                         int32_t *labels, int64_t *path, int64_t *n_out);

/*
 * Exhaustive tie enumeration: the search is replayed under every possible resolution of the ties that can
 * change its result -- which m of k equal candidates straddling the truncation boundary are kept (at any
 * step), and which of the entries tied for the top is walked after the last step.  If all replays give the
 * same (status, labels, path), the result is what the reference returns under ANY tie order, pdqsort's
 * included (argument in fcd_oracle.c above tie_ctx).  labels / path / n_out: the stable-rule result.
 */
int fcdo_beam_search_all_tie_orders(const float *x, int64_t T, int64_t N, int64_t rs, int64_t cs,
                                    int64_t beam_size, float beam_cut_threshold, int collapse_repeats,
                                    int32_t *labels, int64_t *path, int64_t *n_out,
                                    int64_t max_branches, int64_t *n_branches, int64_t *n_distinct,
                                    int *complete);

int fcdo_crf_beam_search_ex(const float *x, int64_t T, int64_t S, int64_t N,
                            int64_t s0, int64_t s1, int64_t s2,
                            const float *init, int64_t n_init, int64_t is0,
                            int64_t beam_size, float beam_cut_threshold,
                            int32_t *labels, int64_t *path, int64_t *n_out, int64_t *n_ambiguous_out);

/* src/search.rs:385-423.  quals[] (nullable) one phred char per label. */
int fcdo_crf_greedy_search(const float *x, int64_t T, int64_t S, int64_t N,
                           int64_t s0, int64_t s1, int64_t s2,
                           const float *init, int64_t n_init, int64_t is0,
                           float qscale, float qbias,
                           int32_t *labels, int64_t *path, uint32_t *quals, int64_t *n_out);

/*
 * src/duplex.rs:443-650.  envelope is (T1,2) uint64 with strides (e0,e1) in elements.
 * labels[] needs room for T1 entries.
 */
int fcdo_beam_search_duplex(const float *x1, int64_t T1, int64_t rs1, int64_t cs1,
                            const float *x2, int64_t T2, int64_t rs2, int64_t cs2,
                            int64_t N, const uint64_t *envelope, int64_t e0, int64_t e1,
                            int64_t beam_size, float beam_cut_threshold, int collapse_repeats,
                            int logadd_mode, int32_t *labels, int64_t *n_out);

/* src/duplex.rs:652-834 */
int fcdo_crf_beam_search_duplex(const float *x1, int64_t T1, const int64_t *st1 /*3 strides*/,
                                const float *init1, int64_t n_init1, int64_t i1s,
                                const float *x2, int64_t T2, const int64_t *st2,
                                const float *init2, int64_t n_init2, int64_t i2s,
                                int64_t S, int64_t N,
                                const uint64_t *envelope, int64_t e0, int64_t e1,
                                int64_t beam_size, float beam_cut_threshold,
                                int logadd_mode, int32_t *labels, int64_t *n_out);

/*
 * Batch driver used by bench.py's cpu_baseline leg and the differential tests:
 * decodes n_reads C-contiguous (T,N) reads with beam_search, n_threads pthreads
 * (one read per task).  labels/path are (n_reads, T) row-major, lens/status (n_reads,).
 * n_passes > 1 decodes the batch that many times (timing only; the first pass is stored).
 */
int fcdo_beam_search_batch(const float *x, int64_t n_reads, int64_t T, int64_t N,
                           int64_t beam_size, float thr, int collapse,
                           int32_t *labels, int64_t *path, int64_t *lens, int32_t *status,
                           int n_threads, int64_t n_passes);

/* as above; ambiguous (nullable, n_reads entries) receives fcdo_beam_search_ex's tie count per read */
int fcdo_beam_search_batch_ex(const float *x, int64_t n_reads, int64_t T, int64_t N,
                              int64_t beam_size, float thr, int collapse,
                              int32_t *labels, int64_t *path, int64_t *lens, int32_t *status,
                              int64_t *ambiguous, int n_threads, int64_t n_passes);

int fcdo_viterbi_batch(const float *x, int64_t n_reads, int64_t T, int64_t N, int collapse,
                       int32_t *labels, int64_t *path, int64_t *lens, int n_threads);

/* ---- white-box hooks for KAT K17 (src/tree.rs:200-269) and K18 (src/duplex.rs:841-993) ---- */
typedef struct fcdo_tree fcdo_tree;
fcdo_tree *fcdo_tree_new(int64_t n_labels);
void fcdo_tree_free(fcdo_tree *t);
int32_t fcdo_tree_add_node(fcdo_tree *t, int32_t parent, int64_t label, int64_t data);
int32_t fcdo_tree_get_child(const fcdo_tree *t, int32_t node, int64_t label); /* -1 = None */
int64_t fcdo_tree_label(const fcdo_tree *t, int32_t node);                    /* -1 = None */
int32_t fcdo_tree_parent(const fcdo_tree *t, int32_t node);
int64_t fcdo_tree_data(const fcdo_tree *t, int32_t node);
int64_t fcdo_tree_len(const fcdo_tree *t);

/* SecondaryProbs window arithmetic on an explicit array of (label,gap) log-probs. */
void fcdo_secondary_get(const float *pairs, int64_t len, int64_t offset, int64_t at,
                        float *label_out, float *gap_out);
float fcdo_secondary_update_max(const float *pairs, int64_t len, int64_t offset,
                                int64_t lower, int64_t upper, int logadd_mode);
float fcdo_logspace_add(float a, float b, int logadd_mode);
/* number of LogSpace::add evaluations made by the calling thread since the last reset (instrumentation) */
int64_t fcdo_logadd_calls(int reset);
/* tie statistics of the duplex searches' prune (src/duplex.rs:620,807): see fcd_oracle.c */
void fcdo_duplex_tie_steps(int64_t out[4], int reset);
/* tie counters of the calling thread's most recent duplex search, defined like fcd_result.ambiguous
 * (include/fcd.h): [0] steps with > 20 candidates and a kept candidate tied, [1] steps with a tie at ranks 0 / 1
 * or across the truncation boundary */
void fcdo_duplex_last_ambiguous(int64_t out[2]);
/* out[i] = fcdo_logspace_add(a[i], b[i], logadd_mode) -- lets tests compare millions of operands */
void fcdo_logspace_add_batch(const float *a, const float *b, float *out, int64_t n, int logadd_mode);
void fcdo_libm_apply(int which, const float *x, float *out, int64_t n); /* 0 expf, 1 logf, 2 log1pf of the host's libm */

/* Order of EQUAL probabilities in the prune of the beam searches (src/search.rs:122,262, src/duplex.rs:620,807:
 * sort_unstable_by).  0 (default): the stable rule -- what Rust's insertion sort does up to 20 candidates, and
 * what FCD_TIE_STABLE selects in the kernels.  1: above 20 candidates, the order left by a restatement of Rust 1.78's
 * pdqsort (written from memory; pinned in round 5 against a compiled rustc-1.65 std except for two routines std
 * changed in 2023: see fcd_oracle.c) -- the default since round 4, like the kernels' FCD_TIE_PDQ178.  Process-wide
 * switch; the tie counters are taken on the stably sorted list in both modes. */
void fcdo_set_unstable_sort(int mode);
int fcdo_get_unstable_sort(void);
/* the two routines of std's pdqsort that changed between rustc 1.65 and 1.78 (fcd_oracle.c): bit 0 = break_patterns'
 * generator as until 2022, bit 1 = partial_insertion_sort's shifting as until 2022; 0 = Rust 1.78 as recalled (default).
 * And how often this thread's quicksorts reached either.  Test hooks: tools/verify/rust165_pdqsort.py */
void fcdo_set_pdq_std_form(int bits);
int fcdo_get_pdq_std_form(void);
int64_t fcdo_pdq_break_patterns_calls(int reset);
int64_t fcdo_pdq_partial_shift_calls(int reset);
/* test hook: an external core::slice::sort::recurse over 24-byte (key, index, -) records replaces the restatement (NULL: off) */
void fcdo_set_external_recurse(void *fn);
void fcdo_test_pdqsort(float *prob, int32_t *node, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* FCD_ORACLE_H */
