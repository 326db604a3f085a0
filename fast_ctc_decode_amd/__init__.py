"""fast_ctc_decode_amd -- MI355X-native (gfx950 / CDNA4) batched CTC decoding behind
fast_ctc_decode's Python API.

    from fast_ctc_decode_amd import beam_search, viterbi_search          # drop-in, one read
    from fast_ctc_decode_amd import beam_search_batch, viterbi_search_batch  # many reads per launch

All searching runs in hand-written HIP kernels reached through the C ABI in include/fcd.h;
importing this package never touches a CPU fallback (there is none).
"""
from .api import (  # noqa: F401
    BatchResult,
    __version__,
    beam_search,
    beam_search_batch,
    beam_search_batch_raw,
    beam_search_duplex,
    beam_search_duplex_batch,
    beam_search_duplex_batch_raw,
    set_duplex_logadd_mode,
    set_tie_order,
    tie_order,
    set_overlap,
    overlap_join,
    set_coalescing,
    coalescing_stats,
    crf_beam_search,
    crf_beam_search_batch,
    crf_beam_search_batch_raw,
    crf_beam_search_duplex,
    crf_beam_search_duplex_batch,
    crf_beam_search_duplex_batch_raw,
    crf_greedy_search,
    crf_greedy_search_batch,
    crf_greedy_search_batch_raw,
    estimate_envelope,
    estimate_envelope_batch,
    viterbi_search,
    viterbi_search_batch,
    viterbi_search_batch_raw,
)
from ._native import (  # noqa: F401
    KERNEL_AUTO, KERNEL_GENERIC, KERNEL_LANE, KERNEL_WAVE, KERNEL_WAVE1, LOGADD_LOGSUMEXP, LOGADD_LOGSUMEXP_GLIBC235, LOGADD_MAX,
    TIE_DEFAULT, TIE_PDQ178, TIE_STABLE, default_tie_order, set_default_tie_order,
)
