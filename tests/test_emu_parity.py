"""CPU-side logic check of the HIP kernels: the product's csrc/*.hip, compiled unchanged against
tests/hipemu's lockstep wave64 emulation, must reproduce the oracle on the same seeded inputs the -m gpu
parity tests use (a selection sized for the CPU suite; FCD_TEST_EMU=1 python -m pytest tests -m gpu runs
them all).  This is NOT the parity gate -- that is tests -m gpu on the MI355X -- it catches logic errors
in cross-lane protocols, node numbering, arena handling and pruning before a GPU is needed."""
import numpy as np
import pytest

import test_gpu_duplex as D
import test_gpu_envelope as E
import test_gpu_parity as P
from emu_util import emulated_kernels


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    with emulated_kernels():
        yield m


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_beam_kernels(fcd, kernel):
    P.test_beam_thr0(fcd, True, kernel)
    P.test_beam_ragged(fcd, kernel)
    P.test_beam_nan_and_zero_rows(fcd, kernel)
    P.test_beam_ties_and_zeros(fcd, kernel)
    P.test_beam_traceback_segment_boundaries(fcd, kernel)


@pytest.mark.parametrize("N,beam,kernel", [(5, 5, 2), (3, 2, 2), (7, 8, 3), (5, 12, 0), (4, 9, 2)])
def test_beam_wave_shapes(fcd, N, beam, kernel):
    if beam > 8:
        P.test_beam_wave_wide(fcd, N, beam, kernel)
    else:
        P.test_beam_wave_random(fcd, N, beam, kernel)


@pytest.mark.parametrize("N,beam", [(5, 32), (5, 64), (8, 13), (2, 5)])
def test_beam_lane_shapes(fcd, N, beam):
    P.test_beam_lane_random(fcd, N, beam)


@pytest.mark.parametrize("N,beam", [(5, 32), (12, 5)])
def test_beam_generic_shapes(fcd, N, beam):
    P.test_beam_generic_random(fcd, N, beam)


def test_full_length_read_every_kernel(fcd):
    """T = 4000 (BASELINE row count): wave (two reads per wavefront), generic, lane at beam 32 and 64."""
    x = P.gen_batch(4242, 2, 4000, 5)
    P.check_beam(fcd, x, 5, 0.1, kernel=0)
    P.check_beam(fcd, x[:1], 5, 0.1, kernel=1)
    P.check_beam(fcd, x, 32, 0.1, kernel=4)
    P.check_beam(fcd, x[:1], 64, 0.1, kernel=4)


def test_lane_two_pass_retry(fcd):
    P.test_lane_two_pass_retry(fcd)


def test_lane_overlapping_calls(fcd):
    P.test_lane_overlapping_calls(fcd)


def test_lane_slab_pool_reuse(fcd):
    P.test_lane_slab_pool_under_contention(fcd, 32)


def test_overlapping_calls_plain_kernels(fcd):
    P.test_overlapping_calls_every_kernel(fcd, 0, 5)


def test_largest_beam_of_the_lds_kernel(fcd):
    P.test_largest_beam_of_the_lds_kernel(fcd)


def test_ambiguity_counter(fcd):
    P.test_ambiguity_counter(fcd)


@pytest.mark.parametrize("S,beam,thr", [(16, 5, 0.1), (1024, 12, 0.05), (64, 32, 0.1), (8, 64, 0.0)])
def test_crf_many_states(fcd, S, beam, thr):
    P.test_crf_beam_many_states(fcd, S, beam, thr)


def test_crf_kernels_and_fuzz(fcd):
    P.test_crf_beam_kernels(fcd, 2, 5, 0.1)
    P.test_crf_bad_state_parity(fcd)
    for seed in range(3000, 3012):
        P.crf_fuzz_seed(fcd, seed)


def test_viterbi_and_crf(fcd):
    P.test_viterbi_random(fcd)
    P.test_viterbi_qual_bits(fcd)
    P.test_viterbi_whole_tiles_without_quality(fcd)
    P.test_viterbi_time_major_storage(fcd, np.float32)
    P.test_viterbi_time_major_storage(fcd, np.float16)
    P.test_crf_beam_random(fcd, 5, 0.1)
    P.test_crf_greedy_random(fcd)
    P.test_crf_greedy_time_major_storage(fcd)


def test_duplex(fcd):
    D.test_duplex_banded_exact(fcd, D.LSE, True)
    D.test_duplex_banded_exact(fcd, D.MAX, False)
    D.test_duplex_special_values_in_wide_windows(fcd, D.MAX)
    D.test_duplex_special_values_in_wide_windows(fcd, D.LSE)
    for seed in (7000, 7001, 100369, 101027):
        assert D.special_values_case(fcd, seed, D.MAX) and D.special_values_case(fcd, seed, D.LSE)
    D.test_duplex_envelope_errors_and_edges(fcd)
    D.test_duplex_shapes_exact(fcd, 3, 3)
    D.test_duplex_receding_upper_bound(fcd, D.LSE)
    D.test_duplex_tie_counters(fcd, D.MAX)
    from oracle import oracle
    x1, i1, x2, i2 = D.crf_pairs(405, 70, 64)   # duplex::crf_beam_search, banded
    env = D.band(70, 64, 20)
    for mode in (D.LSE, D.MAX):
        want = oracle.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, 5, 0.1, mode | D.CR)
        assert fcd.crf_beam_search_duplex(x1, i1, x2, i2, "NACGT", env, 5, 0.1, logadd_mode=mode) == want


def test_duplex_overlapping_calls(fcd):
    D.test_duplex_overlapping_calls(fcd, D.LSE)


def test_duplex_any_shape_kernel_forced(fcd):
    """csrc/duplex.hip is the fallback since r06 (AUTO runs csrc/duplex_slots.hip wherever it fits): it stays under test."""
    D.test_duplex_each_kernel_forced(fcd, 1)


def test_duplex_glibc235_flavour(fcd):
    """FCD_LOGADD_LOGSUMEXP_GLIBC235 == the oracle on the host's libm (this image: glibc 2.35), every pair"""
    if __import__("platform").libc_ver() != ("glibc", "2.35"):
        pytest.skip("the host links another libm")
    D.test_duplex_glibc235_mode_equals_the_oracle_on_the_hosts_libm(fcd)


def test_envelope(fcd):
    E.test_envelope_equals_model(fcd, 3)


def test_beam_fuzz_slice_and_misc(fcd):
    """A slice of the randomised differential fuzz (random shapes / beams / thresholds / ragged lengths / quantised
    ties) on every kernel selection, failing reads, workspace release."""
    for seed in range(52000, 52030):
        x, beam, thr, collapse, lengths = P._fuzz_case(seed)
        for kernel in (0, 1, 2, 3, 4):
            try:
                P.check_beam(fcd, x, beam, thr, collapse, lengths=lengths, kernel=kernel)
            except RuntimeError as e:  # a forced kernel that does not cover the shape says so
                assert kernel in (2, 3, 4) and " kernel: " in str(e), (seed, kernel, str(e))
    P.test_beam_peaky(fcd, 2)
    P.test_beam_peaky(fcd, 3)
    P.test_release_workspace(fcd)


def test_chunked_host_path(fcd):
    """csrc/hostjob.hip under the emulator: chunk boundaries, ragged and failing reads, the job API, and the
    compiled module's batch functions (csrc/pymodule.cpp linked against the emulator library)."""
    import test_gpu_hostjob as H
    H.test_host_pipeline_equals_one_shot(fcd, 3, 5)
    H.test_host_pipeline_crf(fcd)
    H.test_host_pipeline_strided_views(fcd)
    H.test_job_api_chunks_and_cancel(fcd)
    H.test_compiled_batch_functions_equal_per_read_calls(fcd, 3, 2)
    H.test_compiled_duplex_batch_functions_equal_per_read_calls(fcd)
    H.test_list_paths_reference_counts(fcd)
    H.test_time_major_host_views_through_the_batch_functions(fcd)
    H.test_list_of_ragged_reads_without_a_padded_copy(fcd, 3, 3)


def test_half_precision_inputs(fcd):
    """fcd_batch.dtype: float16 / bfloat16 posteriors converted in the kernels' loads == the upcast float32 input"""
    import test_gpu_halfprec as HP
    HP.test_half_precision_host_inputs_every_kernel(fcd)
    HP.test_half_precision_crf_and_duplex(fcd)


@pytest.mark.parametrize("order", ["reverse", "swap-halves", "random:7"])
def test_results_do_not_depend_on_the_order_the_fibres_run_in(fcd, order, monkeypatch):
    """The emulator runs a lane until its next cross-lane operation, so plain stores between two such operations land
    in scheduling order -- not in lockstep's per-instruction order.  Taking the fibres in another order (hipemu.cpp,
    FCD_EMU_LANE_ORDER) makes code that leans on one order fail here instead of only on the GPU: round 4's lane kernel
    wrote half 0's padding over half 1's ids, invisibly with ascending lanes."""
    import test_gpu_tieorder as TO
    monkeypatch.setenv("FCD_EMU_LANE_ORDER", order)
    TO.test_both_tie_orders_every_kernel(fcd, 5, 5, (0, 1, 2, 3, 4))
    TO.test_both_tie_orders_every_kernel(fcd, 5, 32, (1, 4))
    TO.test_both_tie_orders_every_kernel(fcd, 7, 8, (0, 1, 3, 4))
    P.test_beam_fuzz(fcd, 3)
    P.test_crf_fuzz(fcd, 2)


def test_lane_kernel_node_order_by_all_pairs_too(fcd, monkeypatch):
    """The lane kernel puts a tie-flagged step's candidates in node order with a bitmap over the ids' range (r05) and
    falls back to comparing all pairs when the range does not fit its table: the emulator build can force that path."""
    import test_gpu_tieorder as TO
    monkeypatch.setenv("FCD_EMU_LANE_ALLPAIRS", "1")
    TO.test_both_tie_orders_every_kernel(fcd, 5, 32, (1, 4))
    TO.test_both_tie_orders_every_kernel(fcd, 8, 64, (4,))
    TO.test_both_tie_orders_every_kernel(fcd, 7, 8, (4,))
    TO.test_both_tie_orders_every_kernel(fcd, 5, 12, (4,))


def test_tie_orders(fcd):
    """FCD_TIE_PDQ178 / FCD_TIE_STABLE (tests/test_gpu_tieorder.py) under the emulator: every kernel family under both
    orders on inputs built to tie, the CRF and duplex searches, and the BASELINE reads whose result depends on it."""
    import test_gpu_tieorder as TO
    TO.test_both_tie_orders_every_kernel(fcd, 5, 5, (0, 1, 2, 3, 4))
    TO.test_both_tie_orders_every_kernel(fcd, 5, 32, (1, 4))
    TO.test_both_tie_orders_every_kernel(fcd, 8, 64, (4,))
    TO.test_both_tie_orders_every_kernel(fcd, 4, 5, (2, 4))
    TO.test_tie_order_is_per_handle_too(fcd)
    TO.test_crf_beam_both_orders(fcd, "pdq178")
    TO.test_duplex_both_orders(fcd, D.MAX)
