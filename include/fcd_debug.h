/* fcd_debug.h -- test hooks and developer instruments of libfcd_hip.so.
 *
 * NOT part of the drop-in boundary (include/fcd.h): nothing here replaces a function of the reference.  The tests
 * (tests/) and the measurement tools (tools/) use these entry points to reach device routines directly (the
 * quicksort replay, the log-add fast paths, glibc 2.35's libm restatement) and to read cycle accounts out of
 * instrumented kernel instantiations.  A binding generator (bindgen, cgo) should be pointed at fcd.h only. */
#ifndef FCD_DEBUG_H
#define FCD_DEBUG_H

#include "fcd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Which FORM of the two routines Rust's std changed in 2023 the quicksort replay of FCD_TIE_PDQ178 follows (csrc/pdq178.h
 * g_std_form; fcd.h, FCD_PDQ178_STD_FORM): bit 0 = break_patterns' generator as until 2022, bit 1 =
 * partial_insertion_sort's shifting as until 2022; 0 = Rust 1.78 as recalled (the default), 3 = what a compiled
 * rustc-1.65 std does (tools/verify/rust165_pdqsort.py).  The word is process-wide: this call stores it for handles
 * created later and writes it to THIS handle's device at once (other devices that already hold handles keep theirs).
 * For tests and for whoever has just learnt from tools/verify/pdq178_check.rs which form their toolchain carries. */
int fcd_debug_set_pdq178_std_form(fcd_handle *h, int bits);
int fcd_debug_get_pdq178_std_form(void);
/* Test hook: lists (DEVICE u64 [n_lists][stride], lens DEVICE i32 [n_lists]) are sorted in place by the device
 * function the kernels run on tie-flagged steps: descending by the UPPER 32 bits of each element, equal keys in
 * the order Rust 1.78's sort_unstable_by leaves them in; the lower 32 bits ride along. */
int fcd_debug_pdq178_sort_dev(fcd_handle *h, uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens);
/* The same lists through the wave-cooperative form of that routine (csrc/pdq178_wave.h + pdq178_reg.h: what the register
 * kernels run, all 64 lanes on one list): wavefront b sorts list b.  planes = 1, 3, 5 or 8: the instantiation (64 * planes
 * positions; lens[b] must not exceed them); keep: only the first `keep`
 * positions of every list have to come out right (the searches keep beam_size candidates).  Must equal the call
 * above on those positions. */
int fcd_debug_pdq178_coop_sort_dev(fcd_handle *h, uint64_t *lists, int64_t n_lists, int64_t stride, const int32_t *lens,
                                   int planes, int keep);
/* Developer instrument: shader cycles the wavefronts of the call above spent per phase of the routine, summed since the
 * last reset (HOST array): [0] set-up, [1..7] the phases of an LDS partition round (pivot, swap + mode, classification,
 * scans + tables, scatter, children, next segment -- which includes the segments replayed in registers), [8] exit,
 * [9] leaves, [10] segments, [11] calls.  reset: bit 0 = reset afterwards, bit 1 = the wide-beam kernel's IN-PLACE counters
 * of a -DFCD_LANE_TIE_PROF build (tools/dev/lane_tie_prof.sh: [12] list building, [13] replay, [14] ranks handed back, [15]
 * tied steps) instead of the probe kernels'.  (The shipped search kernels carry no stamps.) */
int fcd_debug_pdq178_coop_profile(fcd_handle *h, uint64_t cycles[16], int reset);
/* Developer instrument: wide-beam searches whose worst-case tree arena would exceed 8 GiB (or the workspace
 * limit) run a first pass in slabs of 1/divisor of the worst case (default 2; trees usually reach a third of
 * it) and decode the reads that outgrow their slab again in worst-case slabs carved from the same arena.  A
 * job in which more than a quarter of the reads overflow switches the handle to worst-case slabs for later
 * jobs.  A larger divisor makes the retry path run on small inputs (tests) and pins it; 0 restores the
 * adaptive default. */
int fcd_debug_set_first_pass_divisor(fcd_handle *h, int divisor);
/* Developer instrument: while `cycles` (DEVICE array [n_pairs][16] u32, indexed by the pair's position in the batch)
 * is set, the duplex searches on this handle record a cycle account per pair: shader cycles / 64 spent in
 * [0] envelope + forward-vector extension, [1] LDS tiles, [2] expansion without the window builds, [3] window
 * builds of the new nodes, [4] rank + next beam; [5] window-build loop iterations, [6] new nodes, [7] steps, [8] steps
 * whose extension took the sequential path, [9] nodes that entered the beam, [10] steps with a growing upper bound.
 * NULL switches it off.  The stamps wait for each phase's results (tools/duplex_account.py). */
int fcd_debug_set_duplex_profile(fcd_handle *h, uint32_t *cycles);

/* Test hook: which kernel the duplex searches on this handle run -- 0 automatic (the slot-resident kernel,
 * csrc/duplex_slots.hip, wherever beam_size * N <= 64 and its rings fit the LDS; the any-shape kernel, csrc/duplex.hip,
 * otherwise), 1 the any-shape kernel always, 2 the slot-resident kernel (FCD_E_UNSUPPORTED where it does not fit).  The
 * process default comes from FCD_DUPLEX_KERNEL = "legacy" / "slots".  Results are identical by construction; the
 * tests run both. */
int fcd_debug_set_duplex_kernel(fcd_handle *h, int which);

/* Developer instrument, not part of the drop-in surface: the headline instantiation of the register kernel
 * (beam_size <= 5, N = 5, two reads per wavefront) with a shader-clock stamp after each block of the time
 * step.  cycles: device array [ceil(n_reads / 2)][8] u32 -- per wavefront, cycles summed over the read in
 * blocks 0..6 (row fetch, extensions + push, numbering + stores, key + rank, child-entry upkeep, gather,
 * top + divisions + state) and the step count in [7].  Results in `out` are the search's.  The stamps
 * serialise the blocks, so this measures their dependent latencies (tools/cycle_account.py, profiles/). */
int fcd_beam_search_profile_dev(fcd_handle *h, const fcd_batch *in, int64_t beam_size,
                                float beam_cut_threshold, int collapse_repeats, const fcd_result *out,
                                uint32_t *cycles);

/* Test hook: y[i] = f(x[i]) (device pointers) with the device build of csrc/glibc235_math.h -- which = 0 expf, 1 logf,
 * 2 log1pf -- to be compared with the host's libm (glibc 2.35: identical on every argument). */
int fcd_debug_glibc235_dev(fcd_handle *h, int which, const float *x, float *y, int64_t n);

/* Developer instrument: one wavefront folds n_chain values into an accumulator with the duplex kernel's
 * LogSpace::add, each add waiting for the previous one; cycles[lane] (DEVICE u64[64]) = shader cycles of the chain,
 * sink (DEVICE f32[64]) keeps the result alive.  cycles / n_chain is the dependent latency that bounds the duplex
 * searches (tools/duplex_account.py: the dependent-chain roofline). */
int fcd_logadd_latency_probe_dev(fcd_handle *h, int n_chain, int logadd_mode, uint64_t *cycles, float *sink);
/* Test hook: exhaustive sweep of one fast path of LogSpace::add on the device -- which = 0: exp, 1: ln_1p -- over
 * every f32 bit pattern in [first_bits, last_bits]; counts (DEVICE u64[3]) = arguments, arguments Ziv's test sends
 * to the slow path, arguments whose trusted fast result differs from the library routine's (must be 0). */
int fcd_logadd_sweep_dev(fcd_handle *h, int which, uint32_t first_bits, uint32_t last_bits, uint64_t *counts);

/* Test hook (device pointers, n elements): out_add[i] = LogSpace::add(a[i], b[i]) (src/duplex.rs:42-63)
 * and out_ln[i] = LogSpace::new(a[i]) = ln(a[i]) (:24-26), computed by the very device functions the
 * duplex kernel uses, so the log-space arithmetic can be checked bit for bit against the oracle.
 * logadd_mode: FCD_LOGADD_*, + 4 for the lockstep form of the window-building loop instead of the general one. */
int fcd_logspace_probe_dev(fcd_handle *h, const float *a, const float *b, float *out_add,
                           float *out_ln, int64_t n, int logadd_mode);

#ifdef __cplusplus
}
#endif

#endif /* FCD_DEBUG_H */
