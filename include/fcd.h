/*
 * fcd.h -- C ABI of the MI355X-native batched CTC decoder (libfcd_hip.so).
 *
 * This is the drop-in boundary for fast-ctc-decode's hot path.  The reference exports no C ABI
 * of its own (only PyInit_fast_ctc_decode, /root/reference/src/lib.rs:617-628); its FFI seam is
 * the set of Rust search functions the PyO3 wrappers call with the GIL released.  Each entry
 * point below replaces one of those calls, batched over reads, and cites it:
 *
 *   fcd_viterbi_search_*      <- search::viterbi_search      src/search.rs:320-327 (call site src/lib.rs:199-208)
 *   fcd_beam_search_*         <- search::beam_search         src/search.rs:159-165 (call site src/lib.rs:353-361)
 *   fcd_crf_beam_search_*     <- search::crf_beam_search     src/search.rs:38-44   (call site src/lib.rs:274-282)
 *   fcd_crf_greedy_search_*   <- search::crf_greedy_search   src/search.rs:385-392 (call site src/lib.rs:237-246)
 *   fcd_beam_search_duplex_*  <- duplex::beam_search         src/duplex.rs:443-451 (call site src/lib.rs:474-484)
 *   fcd_crf_beam_search_duplex_* <- duplex::crf_beam_search  src/duplex.rs:652-661 (call site src/lib.rs:563-575)
 *   per-read status codes     <- enum SearchError            src/lib.rs:36-41
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy / HIP types in any signature
 *     (a hipStream_t is passed as void*).
 *   - `*_dev` functions take DEVICE pointers (inputs already resident in HBM, outputs written to
 *     HBM) and only enqueue work on the handle's stream; `*_host` functions take HOST pointers,
 *     stage through the handle's workspace and return after synchronising.
 *   - strides are in ELEMENTS (the reference accepts arbitrarily strided ndarray views,
 *     src/lib.rs:198,352).
 *   - the device writes label INDICES into the caller's alphabet (1..N-1, 0 is the blank and is
 *     never emitted); strings are joined on the host (src/search.rs:293,358).
 *   - `path` values are the u32 row index at which the emitted label's tree node was created
 *     (src/search.rs:214,231,359); the reference's usize is narrowed, T must be < 2^28.
 *   - return value: FCD_OK or a negative FCD_E_* for API misuse / runtime failure;
 *     per-read search outcomes go to status[read] (FCD_ST_*).
 *   - one handle may be used by one host thread at a time (calls on a handle are serialised by
 *     an internal mutex); use one handle per thread / per stream for concurrency.
 */
#ifndef FCD_H
#define FCD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCD_VERSION_MAJOR 0
#define FCD_VERSION_MINOR 1

/* return codes */
enum {
    FCD_OK = 0,
    FCD_E_INVALID = -1,     /* bad argument (null pointer, negative size, unsupported shape) */
    FCD_E_HIP = -2,         /* a HIP runtime call failed; see fcd_last_error() */
    FCD_E_NOMEM = -3,       /* workspace allocation failed */
    FCD_E_UNSUPPORTED = -4, /* shape outside what the kernels implement */
    FCD_E_NODEVICE = -5     /* no usable gfx950 device */
};

/* per-read status, mirrors SearchError (src/lib.rs:36-41) */
enum {
    FCD_ST_OK = 0,
    FCD_ST_RAN_OUT_OF_BEAM = 1,   /* "Ran out of search space (beam_cut_threshold too high)" */
    FCD_ST_INCOMPARABLE = 2,      /* "Failed to compare values (NaNs in input?)" */
    FCD_ST_INVALID_ENVELOPE = 3,  /* "Invalid envelope values" */
    FCD_ST_BAD_STATE = 4,         /* the reference panics (aborts the process) here: a CRF state index outside [0,S),
                                     NaN / out-of-range init_state, or -- duplex -- an envelope whose upper bound moves
                                     back and then forward by less than a beam entry's window already covers
                                     (assert!(current_end < upper_bound), src/duplex.rs:363-366) */
    FCD_ST_INTERNAL = 5           /* tree arena exhausted -- a bug in workspace sizing */
};

/* duplex log-add flavour (SURVEY.md section 0 finding 3) */
enum {
    FCD_LOGADD_LOGSUMEXP = 0, /* reference built with --no-default-features */
    FCD_LOGADD_MAX = 1,       /* reference's default `fastexp` feature (exp() == 0.0) */
    /* FCD_LOGADD_LOGSUMEXP computes ln / exp / ln_1p correctly rounded -- a platform-independent definition.  The
     * reference computes them with whatever expf / logf / log1pf its process links (src/duplex.rs:17,25,50); this
     * flavour reproduces ONE such library bit for bit: glibc 2.35 on x86-64 with FMA (csrc/glibc235_math.h; verified
     * against that libm on every binary32 argument).  Strings then equal the reference's on such a host, not only on
     * >= 90 % of the pairs.  Slower: no fast paths. */
    FCD_LOGADD_LOGSUMEXP_GLIBC235 = 2
};

/* kernel selection for fcd_beam_search_* (0 = pick the fastest that supports the shape) */
enum {
    FCD_KERNEL_AUTO = 0,
    FCD_KERNEL_GENERIC = 1,  /* LDS-resident beam, any beam_size / alphabet */
    FCD_KERNEL_WAVE = 2,     /* register-resident beam: beam_size <= 8 with N <= 7, or beam_size <= 12
                                with N <= 5; packs two reads per wavefront when beam_size <= 5, N <= 5 */
    FCD_KERNEL_WAVE1 = 3,    /* the same kernel, always one read per wavefront */
    FCD_KERNEL_LANE = 4      /* one beam entry per lane: beam_size <= 64, N <= 8, plain (non-CRF) search */
};

/* Order of EQUAL probabilities in the prune of the beam searches (src/search.rs:122,262, src/duplex.rs:620,807:
 * sort_unstable_by on a list that is in ascending node order).  Up to 20 candidates Rust's sort is an insertion
 * sort and ties keep node order; above 20 it is the pattern-defeating quicksort of the pinned toolchain (Rust
 * 1.78.0, .github/workflows/test.yml:16), whose permutation of equal keys is a deterministic function of the list.
 *   FCD_TIE_PDQ178 (default)  on a step with more than 20 candidates in which a candidate that survives the
 *                  truncation ties with another one, that quicksort is replayed on the node-ordered list -- by the
 *                  whole wavefront in the register kernels (csrc/pdq178_wave.h, pdq178_reg.h), by one lane in the
 *                  LDS-resident ones (csrc/pdq178.h) -- and the search adopts its order; every other step is ranked
 *                  exactly on (probability desc, node asc), which is the same thing.  The quicksort was restated
 *                  from memory of library/core/src/slice/sort.rs (no Rust source or toolchain in the build image)
 *                  and then pinned, element for element, against a rustc-1.65 build of std found compiled in that
 *                  image (tools/verify/rust165_pdqsort.py) -- all of it but the two routines std changed in 2023,
 *                  whose LATER forms (Rust 1.78 as recalled) are the default.  The environment variable
 *                  FCD_PDQ178_STD_FORM (0 .. 3, read at load time; bit 0: break_patterns' generator as until 2022,
 *                  bit 1: partial_insertion_sort's shifting as until 2022) selects the earlier ones process-wide
 *                  for the 1-D searches (the duplex searches replay the default form only: csrc/pdq178.h says why;
 *                  none of 1024 BASELINE config-5 pairs decodes differently under another form);
 *                  tools/verify/pdq178_check.rs tells a holder of rustc 1.78.0 in one command which form it carries.
 *   FCD_TIE_STABLE ties always keep ascending node order (what rounds 1-3 shipped): one of the admissible answers
 *                  of an unstable sort, but not the one Rust 1.78 gives on about 0.05 % of BASELINE config-2 reads.
 * A handle follows the process default until fcd_set_tie_order names an order for it (FCD_TIE_DEFAULT: follow
 * again); the process default is FCD_TIE_PDQ178, or what the environment variable FCD_TIE_ORDER (pdq178 | stable)
 * says at load time, or what fcd_set_default_tie_order set last.  The lanes of a host job follow the handle the job
 * was begun on; a coalescer's own handles follow the process default. */
enum { FCD_TIE_DEFAULT = -1, FCD_TIE_PDQ178 = 0, FCD_TIE_STABLE = 1 };

typedef struct fcd_handle fcd_handle;

/* Shape/stride description of a batch of posterior matrices.
 *   1D searches: element (read r, row t, column j)        at base[r*stride_read + t*stride_t + j*stride_n]
 *   CRF searches: element (read r, row t, state s, col j) at base[r*stride_read + t*stride_t + s*stride_s + j*stride_n]
 * lengths (nullable): per-read row count T_r <= T (ragged batches); device pointer for *_dev,
 * host pointer for *_host.
 * Any non-negative strides are accepted.  Two layouts have kernels of their own in the HBM-bound searches
 * (viterbi_search, crf_greedy_search; 16-byte aligned base): READ-MAJOR, every read a contiguous (T, N) / (T, S, N) block
 * (stride_n = 1, stride_s = N, stride_t = S*N), and TIME-MAJOR, the (T, B, N) / (T, B, S, N) tensor a basecaller network
 * emits seen as a batch (stride_read = S*N, stride_t = B*S*N: no transposition needed).  The beam searches fetch one row
 * per step and read and run at the same speed on either. */
/* Element type of the posteriors.  Basecaller networks emit half precision; the reference only takes float32
 * (src/lib.rs:182,325: &PyArray<f32>), which costs its callers a host-side upcast.  f16 and bf16 convert to float32
 * EXACTLY, so a search on half-precision input IS the reference's search on the upcast matrix; the kernels convert
 * in registers while loading (no separate pass; the HBM-bound viterbi search streams half the bytes). */
enum { FCD_DTYPE_F32 = 0, FCD_DTYPE_F16 = 1, FCD_DTYPE_BF16 = 2 };

typedef struct fcd_batch {
    const void *post;   /* elements of type `dtype`: float (FCD_DTYPE_F32) or 16-bit words (F16 / BF16) */
    int64_t n_reads;
    int64_t T;          /* rows allocated per read */
    int64_t S;          /* CRF states; 1 for the plain searches */
    int64_t N;          /* alphabet size including the blank */
    int64_t stride_read;
    int64_t stride_t;
    int64_t stride_s;
    int64_t stride_n;
    const int64_t *lengths;
    int32_t dtype;      /* FCD_DTYPE_*; strides are in elements of that type (0 = float32: zero-initialised structs
                           keep their meaning) */
} fcd_batch;

/* Output of the 1D searches; every array has n_reads rows.
 *   labels : [n_reads * out_stride] u8   alphabet index of each emitted label, in sequence order
 *   path   : [n_reads * out_stride] u32  (nullable)
 *   qual   : [n_reads * out_stride] f32  (nullable; viterbi/crf_greedy only) the probability the
 *            reference feeds to phred() for each emitted label (src/search.rs:348-356,370-376)
 *   out_len: [n_reads] u32   number of emitted labels
 *   status : [n_reads] i32   FCD_ST_*  (nullable for viterbi)
 *   ambiguous: [n_reads][2] u32 (nullable; the beam searches -- fcd_beam_search_*, fcd_crf_beam_search_* and the
 *            two duplex searches, whose prune is the same sort_unstable_by, src/duplex.rs:620,807; device pointer
 *            for *_dev, host pointer for *_host) -- a TIE INSTRUMENT, not a reference output.  The reference
 *            orders candidates with sort_unstable_by (src/search.rs:122,262): a stable insertion sort up to
 *            20 elements, pdqsort -- implementation-defined tie order -- above.  Under FCD_TIE_STABLE the kernels
 *            break exact probability ties by ascending node index, which is what the stable path does; under
 *            FCD_TIE_PDQ178 (default) the steps counted in [r][0] are exactly the ones re-ranked by the restated
 *            quicksort.  The counters do not depend on the tie order within a step.
 *              [r][0] steps with MORE than 20 candidates in which a candidate that survives the truncation has
 *                     exactly the probability of another candidate.  0 for a read => its beam, set and order,
 *                     follows the reference step for step;
 *              [r][1] steps (any candidate count) with equal probabilities at ranks 0 / 1 or across the
 *                     truncation boundary.  0 for a read => no tie rule can change a kept set or the best entry.
 *            A read with either counter at 0 is pinned to the reference; the others can be settled with the
 *            oracle's exhaustive replay (oracle/fcd_oracle.h, fcdo_beam_search_all_tie_orders).
 *            Passing the array selects instrumented kernel instantiations (slower by a few percent).
 * out_stride must be >= the longest possible output (T is always enough). */
typedef struct fcd_result {
    uint8_t *labels;
    uint32_t *path;
    float *qual;
    uint32_t *out_len;
    int32_t *status;
    int64_t out_stride;
    uint32_t *ambiguous;
} fcd_result;

/* ---- library / handle ---- */
int fcd_version(void);                                   /* major*1000 + minor */
int fcd_device_count(void);                              /* number of visible HIP devices, 0 if none */
int fcd_create(int device, fcd_handle **out);            /* binds to a device, creates its own stream */
int fcd_destroy(fcd_handle *h);                         /* FCD_E_INVALID while a host job (fcd_*_host_begin) is open */
int fcd_set_stream(fcd_handle *h, void *hip_stream);     /* launch on this hipStream_t; NULL = the HIP null (legacy default) stream */
int fcd_reset_stream(fcd_handle *h);                     /* back to the handle's own non-blocking stream */
int fcd_synchronize(fcd_handle *h);                      /* waits for the handle's stream (and for overlapping calls, below) */
/* Overlapping calls (no counterpart in the reference, whose searches are synchronous: src/lib.rs:199).
 * A batch is as slow as its slowest read, and a batch of 4096 reads fills half of the chip's wavefront slots: under
 * FCD_TIE_PDQ178 a wide-beam read whose every step ties runs 2.3x as long as the rest of its batch, on
 * one wavefront, while the chip idles.  fcd_set_overlap(h, n), n in 2 .. 8: the *_dev beam searches (1-D and duplex)
 * are enqueued round-robin on n internal streams -- each behind the handle's stream AS IT STOOD WHEN THE CALL WAS
 * MADE, not behind one another, so the stragglers of a call run under the next calls.  Wide-beam jobs share ONE tree
 * arena whose slabs are handed out on the device, as many as the chip holds wavefronts (csrc/slab_pool.h); the other
 * kernels' calls get a region of the workspace per internal stream.  Results are complete once fcd_overlap_join(h) has
 * made the handle's stream wait for them (fcd_overlap_join_stream: any other hipStream_t), or after fcd_synchronize.
 * Any *_dev search on this handle whose output arrays overlap those of a call still in flight is ordered behind it;
 * READERS of such results (fcd_pack_results_dev, the caller's own kernels and copies) need the join, and the inputs of a
 * call must stay untouched until it is joined.  n = 0 (default): every call is in stream order; changing n joins what is
 * in flight.  The internal streams are created in the high priority class, whose hardware queues the runtime hands out
 * separately from those of the process's normal streams. */
int fcd_set_overlap(fcd_handle *h, int streams);
int fcd_overlap_join(fcd_handle *h);
int fcd_overlap_join_stream(fcd_handle *h, void *hip_stream);
/* One call at a time: fcd_overlap_last_slot = the internal stream (0 .. n-1) the latest overlapping call went to, -1 if
 * none; fcd_overlap_join_slot makes `hip_stream` wait for what THAT internal stream has been given so far -- the call
 * itself as long as fewer than n further calls have been made since.  (A consumer that lags n - 1 calls behind -- a
 * gather of results on a communication stream -- waits for exactly the call it consumes.) */
int fcd_overlap_last_slot(fcd_handle *h);
int fcd_overlap_join_slot(fcd_handle *h, int slot, void *hip_stream);
const char *fcd_last_error(const fcd_handle *h);         /* text of the last failure on this handle */
const char *fcd_status_string(int status);               /* exact SearchError Display text, src/lib.rs:46-53 */
/* cap (bytes) on the per-call tree-arena workspace; batches needing more are decoded in chunks */
int fcd_set_workspace_limit(fcd_handle *h, int64_t bytes);
/* The handle's device workspace (tree arena, staging) only grows and is kept between calls; this waits for the
 * handle's stream and gives it all back (the next call allocates afresh).  For processes that share the GPU
 * with another allocator (PyTorch's caching allocator) after an unusually large job. */
int fcd_release_workspace(fcd_handle *h);
/* tie order of the beam searches' prune (FCD_TIE_*, above) */
int fcd_set_tie_order(fcd_handle *h, int order);         /* FCD_TIE_DEFAULT: follow the process default again */
int fcd_get_tie_order(const fcd_handle *h);              /* the order searches on this handle use right now */
int fcd_set_default_tie_order(int order);                /* FCD_TIE_PDQ178 or FCD_TIE_STABLE; process-wide */
/* (test hooks and developer instruments -- fcd_debug_*, the probes and sweeps -- are declared in fcd_debug.h: the
 * library exports them, but they are no part of the surface a binding generator should consume) */
/* duration (ms) of the decode kernel(s) of the last call on this handle, measured with HIP
 * events on the stream the kernels were launched on; <0 if unavailable */
double fcd_last_kernel_ms(fcd_handle *h);
/* Every search call brackets its kernel launches with a HIP event pair on the launch stream
 * (ring of 256).  fcd_timing_reset forgets them; fcd_timing_mean_ms synchronises on and averages
 * the pairs recorded since the reset (n_calls, nullable, receives how many). */
int fcd_timing_reset(fcd_handle *h);
double fcd_timing_mean_ms(fcd_handle *h, int64_t *n_calls);

/* ---- search::viterbi_search (src/search.rs:320-383) ---- */
int fcd_viterbi_search_dev(fcd_handle *h, const fcd_batch *in, int collapse_repeats,
                           const fcd_result *out);
int fcd_viterbi_search_host(fcd_handle *h, const fcd_batch *in, int collapse_repeats,
                            const fcd_result *out);

/* ---- search::beam_search (src/search.rs:159-301) ---- */
int fcd_beam_search_dev(fcd_handle *h, const fcd_batch *in, int64_t beam_size,
                        float beam_cut_threshold, int collapse_repeats, int kernel,
                        const fcd_result *out);
int fcd_beam_search_host(fcd_handle *h, const fcd_batch *in, int64_t beam_size,
                         float beam_cut_threshold, int collapse_repeats, int kernel,
                         const fcd_result *out);

/* ---- search::crf_beam_search (src/search.rs:38-157) ----
 * init: [n_reads * init_stride] f32, n_init entries used per read (src/search.rs:54-59). */
int fcd_crf_beam_search_dev(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                            int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                            const fcd_result *out);
/* same as fcd_crf_beam_search_dev with an explicit FCD_KERNEL_* choice (tests / benchmarks) */
int fcd_crf_beam_search_dev_k(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                              int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                              int kernel, const fcd_result *out);
int fcd_crf_beam_search_host(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                             int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                             const fcd_result *out);
int fcd_crf_beam_search_host_k(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                               int64_t init_stride, int64_t beam_size, float beam_cut_threshold,
                               int kernel, const fcd_result *out);

/* ---- search::crf_greedy_search (src/search.rs:385-423) ---- */
int fcd_crf_greedy_search_dev(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                              int64_t init_stride, const fcd_result *out);
int fcd_crf_greedy_search_host(fcd_handle *h, const fcd_batch *in, const float *init,
                               int64_t n_init, int64_t init_stride, const fcd_result *out);

/* ---- duplex::beam_search (src/duplex.rs:443-650) ----
 * (Two kernels serve the duplex searches since round 6, with identical results: the slot-resident one --
 * csrc/duplex_slots.hip -- wherever beam_size * N <= 64 and the live nodes' windows fit the LDS, the any-shape one --
 * csrc/duplex.hip -- otherwise.  Nothing at this boundary depends on which runs; FCD_DUPLEX_KERNEL=legacy|slots and
 * fcd_debug_set_duplex_kernel (fcd_debug.h) force one for A/B runs and tests.)
 * in1/in2 describe the two reads of each pair (same n_reads and N); envelope is
 * [n_reads * env_stride] pairs of u64 (lo,hi), row t of pair r at envelope[(r*env_stride + t)*2].
 * Only labels/out_len/status (and the tie counters `ambiguous`, when given) of `out` are written (the reference
 * returns the string only). */
int fcd_beam_search_duplex_dev(fcd_handle *h, const fcd_batch *in1, const fcd_batch *in2,
                               const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                               float beam_cut_threshold, int collapse_repeats, int logadd_mode,
                               const fcd_result *out);
int fcd_beam_search_duplex_host(fcd_handle *h, const fcd_batch *in1, const fcd_batch *in2,
                                const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                                float beam_cut_threshold, int collapse_repeats, int logadd_mode,
                                const fcd_result *out);

/* ---- duplex::crf_beam_search (src/duplex.rs:652-834) ----
 * in1/in2 are (T,S,N) batches with the same S and N; init1/init2: [n_reads * initX_stride] f32 with
 * n_initX entries used per pair (the start state is their argmax, src/duplex.rs:679,691). */
int fcd_crf_beam_search_duplex_dev(fcd_handle *h, const fcd_batch *in1, const float *init1,
                                   int64_t n_init1, int64_t init1_stride, const fcd_batch *in2,
                                   const float *init2, int64_t n_init2, int64_t init2_stride,
                                   const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                                   float beam_cut_threshold, int logadd_mode, const fcd_result *out);
int fcd_crf_beam_search_duplex_host(fcd_handle *h, const fcd_batch *in1, const float *init1,
                                    int64_t n_init1, int64_t init1_stride, const fcd_batch *in2,
                                    const float *init2, int64_t n_init2, int64_t init2_stride,
                                    const uint64_t *envelope, int64_t env_stride, int64_t beam_size,
                                    float beam_cut_threshold, int logadd_mode, const fcd_result *out);

/* ---- alignment-band estimator for the duplex searches ----
 * NOT a reference function: duplex::beam_search takes the envelope as an argument, the PyO3 wrapper
 * defaults to the full matrix and its docstring only anticipates a better default
 * (/root/reference/src/lib.rs:376-378,459-468).  Input: per pair, the label sequences of the two
 * reads with their emission times (e.g. the labels / path / out_len arrays of fcd_viterbi_search_*).
 * The label sequences are aligned globally (unit-cost edit distance); matched labels anchor read-1
 * time to read-2 time; envelope row i = [centre(i) - band, centre(i) + band + 1) clipped to
 * [0, T2], with lo(0) = 0, hi(T1 - 1) = T2 and consecutive rows touching (src/duplex.rs:485-488).
 * T1 / T2: nullable per-pair row counts (else T1cap / T2cap).  Limits: at most 65535 labels per pair and
 * ~13000 labels in read 2 (the DP rows live in LDS as u16; FCD_E_UNSUPPORTED beyond).  Executable specification: tests/envelope_model.py. */
int fcd_duplex_envelope_dev(fcd_handle *h, int64_t n_pairs,
                            const uint8_t *labels1, const uint32_t *path1, const uint32_t *len1,
                            int64_t stride1, const int64_t *T1, int64_t T1cap,
                            const uint8_t *labels2, const uint32_t *path2, const uint32_t *len2,
                            int64_t stride2, const int64_t *T2, int64_t T2cap,
                            int64_t band, uint64_t *envelope, int64_t env_stride);
int fcd_duplex_envelope_host(fcd_handle *h, int64_t n_pairs,
                             const uint8_t *labels1, const uint32_t *path1, const uint32_t *len1,
                             int64_t stride1, const int64_t *T1, int64_t T1cap,
                             const uint8_t *labels2, const uint32_t *path2, const uint32_t *len2,
                             int64_t stride2, const int64_t *T2, int64_t T2cap,
                             int64_t band, uint64_t *envelope, int64_t env_stride);


/* ---- compact wire format of a shard's results, for the ONE gather of the multi-GPU path ----
 * The searches write fixed-stride rows; only out_len[r] entries of row r are meaningful (~48 % at BASELINE
 * config 2).  These three calls turn a result into one contiguous buffer holding just the used prefixes
 * (layout in csrc/pack.hip) and back.  All pointers are device memory; work is enqueued on the handle's stream.
 *   1. fcd_result_offsets_dev: offsets[i] = sum of out_len[0..i) (clamped to out_stride), offsets[n_reads] =
 *      total labels.  Read offsets[n_reads] back to size the buffer: fcd_packed_result_bytes().
 *   2. fcd_pack_results_dev: labels/path(u32 -> path_bytes = 2 or 4 wide)/out_len/status -> buf.
 *   3. fcd_unpack_results_dev: buf -> fixed-stride arrays of `out` (offsets: workspace of n_reads+1 u64). */
int64_t fcd_packed_result_bytes(int64_t n_reads, int64_t total_labels, int path_bytes);
int fcd_result_offsets_dev(fcd_handle *h, const uint32_t *out_len, int64_t n_reads, int64_t out_stride,
                           uint64_t *offsets);
int fcd_pack_results_dev(fcd_handle *h, const fcd_result *res, int64_t n_reads, int path_bytes,
                         const uint64_t *offsets, uint8_t *buf);
int fcd_unpack_results_dev(fcd_handle *h, const uint8_t *buf, int64_t n_reads, uint64_t *offsets,
                           const fcd_result *out);

/* ---- the multi-GPU step without Python (csrc/comm.hip) ----------------------------------------
 * One process per GPU; reads shard across the ranks with no exchange inside the searches; what is left is ONE
 * gather of every rank's decoded results on a destination rank over RCCL / xGMI.  RCCL is loaded at run time
 * (dlopen), so single-GPU users never need it.
 *   fcd_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId); the host hands it to the other ranks
 *   fcd_comm_create      ncclCommInitRank on the handle's device (collective: every rank calls it)
 *   fcd_comm_wrap        use an ncclComm_t the host already owns (passed as void*); NULL is allowed for world = 1
 *   fcd_gather_results_dev  this rank's `res` (n_reads fixed-stride rows in device memory, e.g. what
 *                        fcd_beam_search_dev wrote) -> on rank `dst`, `out` receives all ranks' rows in global read
 *                        order (counts[k] = reads of rank k; out->out_stride >= res->out_stride).  Steps, all on
 *                        the handle's stream: prefix sums + pack of the used prefixes (u16 times below 65536 rows),
 *                        a 16-byte ncclAllReduce(MAX) of {label total, out_stride} so that all ranks send one size (the
 *                        ranks' out_stride should be equal; the destination's must not be narrower), ONE ncclGather, and on `dst`
 *                        the unpack of every shard with one pair of launches.  The call waits for the device once
 *                        (the agreed size: 16 bytes); the gather and the unpack are left in flight on the stream.  An error a
 *                        rank finds after the all-reduce is reported once its own ncclGather is enqueued: no rank is left hanging.
 *   fcd_comm_synchronize waits for the stream and reports a shard whose header contradicted `counts`
 *                        (FCD_E_INVALID; such a shard's rows are left empty, never read out of bounds). */
#define FCD_COMM_ID_BYTES 128
typedef struct fcd_comm fcd_comm;
int fcd_comm_unique_id(uint8_t id[FCD_COMM_ID_BYTES]);
int fcd_comm_create(fcd_handle *h, int world, int rank, const uint8_t id[FCD_COMM_ID_BYTES], fcd_comm **out);
int fcd_comm_wrap(fcd_handle *h, void *nccl_comm, int world, int rank, fcd_comm **out);
int fcd_comm_destroy(fcd_comm *c);
int fcd_gather_results_dev(fcd_comm *c, const fcd_result *res, int64_t n_reads, const int64_t *counts, int dst,
                           const fcd_result *out);
int fcd_comm_synchronize(fcd_comm *c);
/* The unpack step on its own (what fast_ctc_decode_amd/dist.py calls after torch.distributed's gather):
 * gathered = world packed shards of `stride` bytes each, back to back; first = DEVICE array [world + 1] of
 * prefix sums of the per-rank read counts; offsets = DEVICE workspace of first[world] + world u64; bad = DEVICE
 * int32 that receives 1 + shard for a shard whose header contradicts the counts or the buffer size. */
int fcd_unpack_gathered_dev(fcd_handle *h, const uint8_t *gathered, int64_t stride, int world, const int64_t *first,
                            int64_t n_total, uint64_t *offsets, const fcd_result *out, int32_t *bad);

/* ---- coalescing front door for per-read callers (csrc/coalesce.hip) --------------------------------------
 * The reference decodes ONE read per call and releases the GIL around the search so that callers can run it
 * from many threads (src/lib.rs:199 viterbi_search, :353 beam_search).  On a GPU a lone read is one wavefront;
 * a coalescer turns CONCURRENT per-read calls into batched launches without changing the callers: the first
 * thread to arrive decodes every compatible pending request (same search, alphabet size, beam size, threshold,
 * collapse flag) with one ragged batch on one of the coalescer's own handles; whatever arrives while that launch
 * is in flight forms the next batch.  max_wait_us = 0 (adaptive): before launching, a leader waits briefly for
 * the callers of the previous batches to come back (until as many requests are pending as recent batches held,
 * at most an eighth of the last launch's duration, <= 1 ms); a lone caller never waits.  max_wait_us > 0: every
 * leader waits up to that long for company (or until max_batch requests are pending).  A few
 * leaders run side by side while fewer than four reads are in flight, so a handful of callers overlap like
 * independent per-read calls.  Results are bit-identical to the per-read calls.
 * `read` must describe exactly one (T, N) matrix (n_reads = 1, S <= 1, non-negative strides, no lengths);
 * `out` is a one-read fcd_result with out_stride >= T.  Blocking; any number of threads.  On failure the text
 * is in fcd_coalescer_last_error() (per calling thread). */
typedef struct fcd_coalescer fcd_coalescer;
int fcd_coalescer_create(int device, int max_batch, int max_wait_us, fcd_coalescer **out);
int fcd_coalescer_destroy(fcd_coalescer *c);   /* FCD_E_INVALID while calls are in flight */
int fcd_coalescer_beam_search(fcd_coalescer *c, const fcd_batch *read, int64_t beam_size,
                              float beam_cut_threshold, int collapse_repeats, const fcd_result *out);
int fcd_coalescer_viterbi_search(fcd_coalescer *c, const fcd_batch *read, int collapse_repeats,
                                 const fcd_result *out);
/* r04: the CRF searches (src/search.rs:38-157, :385-423) through the same door: one (T, S, N) read and its init_state
 * (n_init contiguous floats) per call; calls with the same S, N, n_init, beam and threshold share a launch. */
int fcd_coalescer_crf_beam_search(fcd_coalescer *c, const fcd_batch *read, const float *init, int64_t n_init,
                                  int64_t beam_size, float beam_cut_threshold, const fcd_result *out);
int fcd_coalescer_crf_greedy_search(fcd_coalescer *c, const fcd_batch *read, const float *init, int64_t n_init,
                                    const fcd_result *out);
/* r05: the pair searches (src/duplex.rs:443-650, :652-834; the reference's beam_search_duplex /
 * crf_beam_search_duplex, src/lib.rs:401-578, decode ONE pair per call with the GIL released): read1 / read2 as above,
 * `envelope` = read1->T rows of {lo, hi} (u64), logadd_mode = FCD_LOGADD_*.  A lone pair is one wavefront walking a
 * 2000-step chain: per-pair callers get the latency of a whole batch per call -- through this door concurrent calls
 * with the same N (S, init sizes), beam, threshold, collapse flag and log-add flavour share ONE launch. */
int fcd_coalescer_beam_search_duplex(fcd_coalescer *c, const fcd_batch *read1, const fcd_batch *read2,
                                     const uint64_t *envelope, int64_t beam_size, float beam_cut_threshold,
                                     int collapse_repeats, int logadd_mode, const fcd_result *out);
int fcd_coalescer_crf_beam_search_duplex(fcd_coalescer *c, const fcd_batch *read1, const float *init1, int64_t n_init1,
                                         const fcd_batch *read2, const float *init2, int64_t n_init2,
                                         const uint64_t *envelope, int64_t beam_size, float beam_cut_threshold,
                                         int logadd_mode, const fcd_result *out);
int fcd_coalescer_stats(fcd_coalescer *c, int64_t *n_calls, int64_t *n_launches, int64_t *largest_batch);
const char *fcd_coalescer_last_error(void);

/* ---- large HOST batches as a stream of result chunks (csrc/hostjob.hip) -----------------------------------
 * The reference's callers hold posteriors in host memory (src/lib.rs:182,325: &PyArray2<f32>) and want strings
 * and paths back (src/lib.rs:208,361).  A job decodes a host batch chunk by chunk on a few internal lanes (own
 * stream, staging area and tree arena each): the upload of chunk c+1 overlaps the searches of the chunks before
 * it, only the USED prefix of every result row crosses PCIe (times as u16 when T < 65536), and the caller can
 * turn chunk c into its own objects while later chunks are still in flight.
 *   fcd_*_host_begin   same arguments and checks as fcd_*_host (host pointers; they must stay valid until
 *                      fcd_job_end) plus `want`, an OR of FCD_JOB_*; starts the pipeline and returns.
 *   fcd_job_next       blocks until the next chunk (in read order) is in host memory and describes it in *out;
 *                      the view stays valid until the next fcd_job_next / fcd_job_end on the job.  Returns
 *                      FCD_OK, FCD_JOB_DONE after the last chunk, or a negative FCD_E_*.
 *   fcd_job_end        always call: waits for / cancels outstanding work and frees the job.
 * One job at a time per handle; the handle's other entry points may be used again after fcd_job_end.
 * fcd_*_host on large batches (>= 128 reads and >= 16 MB) runs the same pipeline and expands the chunks into the
 * caller's fixed-stride arrays.  Environment: FCD_HOST_LANES (default 3; 1 = no pipelining), FCD_HOST_CHUNK
 * (reads per chunk; default = the batch split evenly over the lanes, at most 2048). */
enum { FCD_JOB_PATH = 1, FCD_JOB_QUAL = 2, FCD_JOB_AMBIGUOUS = 4 };
enum { FCD_JOB_DONE = 1 };
typedef struct fcd_job fcd_job;
typedef struct fcd_chunk {
    int64_t read_begin;        /* index of the chunk's first read in the batch */
    int64_t n_reads;
    const uint32_t *out_len;   /* [n_reads] */
    const int32_t *status;     /* [n_reads] FCD_ST_* */
    const uint64_t *offsets;   /* [n_reads + 1]: read i owns entries offsets[i] .. offsets[i+1] of the arrays below */
    const uint8_t *labels;     /* label indices of read 0, read 1, ... back to back */
    const void *path;          /* u16 (path_bytes == 2) or u32 (4) row indices, same order; NULL unless FCD_JOB_PATH */
    int path_bytes;
    const float *qual;         /* NULL unless FCD_JOB_QUAL (viterbi / crf_greedy) */
    const uint32_t *ambiguous; /* [n_reads][2] tie counters (fcd_result.ambiguous); NULL unless FCD_JOB_AMBIGUOUS */
} fcd_chunk;
int fcd_viterbi_search_host_begin(fcd_handle *h, const fcd_batch *in, int collapse_repeats, int want, fcd_job **job);
int fcd_beam_search_host_begin(fcd_handle *h, const fcd_batch *in, int64_t beam_size, float beam_cut_threshold,
                               int collapse_repeats, int kernel, int want, fcd_job **job);
int fcd_crf_beam_search_host_begin(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                                   int64_t init_stride, int64_t beam_size, float beam_cut_threshold, int kernel,
                                   int want, fcd_job **job);
int fcd_crf_greedy_search_host_begin(fcd_handle *h, const fcd_batch *in, const float *init, int64_t n_init,
                                     int64_t init_stride, int want, fcd_job **job);
/* The same jobs for a batch whose reads are SEPARATE host arrays -- a Python list of ragged matrices, which is how the
 * reference's callers hold them (src/lib.rs:325,352 take one read per call): reads[r] points at read r's contiguous
 * (rows[r], N) matrix of element type `dtype`.  Every chunk is gathered into page-locked memory by its lane (the lanes
 * gather side by side) and leaves as one DMA: no padded copy of the batch on the caller's side.  reads / rows must stay
 * valid until fcd_job_end.  Chunk views are the same (out_stride = the longest read). */
int fcd_viterbi_search_host_ptrs_begin(fcd_handle *h, const void *const *reads, const int64_t *rows, int64_t n_reads,
                                       int64_t N, int dtype, int collapse_repeats, int want, fcd_job **job);
int fcd_beam_search_host_ptrs_begin(fcd_handle *h, const void *const *reads, const int64_t *rows, int64_t n_reads,
                                    int64_t N, int dtype, int64_t beam_size, float beam_cut_threshold,
                                    int collapse_repeats, int kernel, int want, fcd_job **job);
/* Tuning / tests: lanes (0 = default 3 or FCD_HOST_LANES; 1 = never pipeline), reads per chunk (0 = automatic),
 * and the input size from which fcd_*_host takes the pipeline (-1 = default: >= 128 reads and >= 16 MB). */
int fcd_set_host_pipeline(fcd_handle *h, int lanes, int64_t chunk_reads, int64_t min_bytes);
int fcd_job_chunks(const fcd_job *job, int64_t *chunk_reads, int *n_lanes);  /* number of chunks; -1 for NULL */
int fcd_job_next(fcd_job *job, fcd_chunk *out);
int fcd_job_end(fcd_job *job);

/* ---- host-side helpers shared with the language bindings ---- */
/* phred quality character code point for a probability (src/search.rs:31-36) */
uint32_t fcd_phred(float prob, float qscale, float qbias);

#ifdef __cplusplus
}
#endif
#endif /* FCD_H */
