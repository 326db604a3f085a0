"""Known-answer tests (golden vectors) held by the reference's own test suites.

Each case is a function taking `m`, a module-like object with the reference's Python surface
(beam_search, viterbi_search, crf_beam_search, crf_greedy_search, beam_search_duplex, ...).
They are run against the CPU oracle (tests/test_oracle_kat.py, no GPU) to pin the oracle, and
against the HIP product through its C-ABI (tests/test_gpu_kat.py, -m gpu).

Inputs and expected outputs are data restated from the reference's tests; the ids K1..K18 are
SURVEY.md section 4's.  Source of each vector is cited as /root/reference path:line.
"""
import numpy as np
import pytest

K1_MATRIX = np.array([  # src/search.rs:533-544 == tests/fast_ctc_wasm.test.js:12
    [0.0, 0.4, 0.6],
    [0.0, 0.3, 0.7],
    [0.3, 0.3, 0.4],
    [0.4, 0.3, 0.3],
    [0.4, 0.3, 0.3],
    [0.3, 0.3, 0.4],
    [0.1, 0.4, 0.5],
    [0.1, 0.5, 0.4],
    [0.8, 0.1, 0.1],
    [0.1, 0.1, 0.8],
], np.float32)

K3_MATRIX = np.concatenate([  # src/search.rs:562-576
    np.array([[0.6, 0.2, 0.2], [0.6, 0.2, 0.2]], np.float32),
    K1_MATRIX,
    np.array([[0.4, 0.3, 0.3]], np.float32),
])

NAG = ["N", "A", "G"]


def k1_viterbi(m):  # src/search.rs:546-554
    assert m.viterbi_search(K1_MATRIX, NAG, False, 1.0, 0.0, True) == ("GGAG", [0, 5, 7, 9])
    assert m.viterbi_search(K1_MATRIX, NAG, True, 1.0, 0.0, True) == ("GGAG%$$(", [0, 5, 7, 9])


def k2_beam_wasm(m):  # tests/fast_ctc_wasm.test.js:29-36
    assert m.beam_search(K1_MATRIX, NAG, 5, 0.0, True) == ("GAGAG", [0, 1, 2, 4, 6])


def k3_viterbi_blank_bounds(m):  # src/search.rs:577-594
    assert m.viterbi_search(K3_MATRIX, NAG, False, 1.0, 0.0, True) == ("GGAG", [2, 7, 9, 11])
    assert m.viterbi_search(K3_MATRIX, NAG, True, 1.0, 0.0, True) == ("GGAG%$$(", [2, 7, 9, 11])
    assert m.viterbi_search(K3_MATRIX, NAG, False, 1.0, 0.0, False) == (
        "GGGGGAG", [2, 3, 4, 7, 8, 9, 11])
    assert m.viterbi_search(K3_MATRIX, NAG, True, 1.0, 0.0, False) == (
        "GGGGGAG%&##$$(", [2, 3, 4, 7, 8, 9, 11])


def k4_beam_blank_bounds(m):  # src/search.rs:596-600
    assert m.beam_search(K3_MATRIX, NAG, 5, 0.0, True)[0] == "GAGAG"
    assert m.beam_search(K3_MATRIX, NAG, 5, 0.0, False)[0] == "GGGAGAG"


def _crf_tensor():  # src/search.rs:440-483
    x = np.zeros((7, 4, 5), np.float32)
    x[0, 2, 0] = 1.0
    x[1, 2, 2] = 0.9
    x[2, 1, 4] = 0.7
    x[3, 3, 0] = 1.0
    x[4, 3, 1] = 0.99
    x[5, 0, 1] = 0.9
    x[6, 0, 3] = 0.999
    return x, np.array([0, 0, 1, 0, 0], np.float32)


def k6_crf_greedy(m):  # src/search.rs:485-495
    x, init = _crf_tensor()
    assert m.crf_greedy_search(x, init, "NACGT", False, 1.0, 0.0) == ("CTAAG", [1, 2, 4, 5, 6])
    assert m.crf_greedy_search(x, init, "NACGT", True, 1.0, 0.0) == ("CTAAG+&5+?", [1, 2, 4, 5, 6])


def k6_crf_beam(m):  # src/search.rs:497-509
    x, init = _crf_tensor()
    assert m.crf_beam_search(x, init, "NACGT", 5, 0.01) == ("CTAAG", [1, 2, 4, 5, 6])


def _path_matrix(w=5000):  # tests/test_decode.py:122-131
    x = np.zeros((w, 5), np.float32)
    x[:, 0] = 0.5
    emit = np.arange(0, w, 4)
    for base, pos in enumerate(emit):
        x[pos, base % 4 + 1] = 1.0
    return x, emit


def k7_beam_path(m):  # tests/test_decode.py:122-135
    x, emit = _path_matrix()
    seq, path = m.beam_search(x, "NACGT", 5, 0.1)
    np.testing.assert_array_equal(emit, path)
    assert len(seq) == len(path)


def k7_viterbi_path(m):  # tests/test_decode.py:227-240
    x, emit = _path_matrix()
    seq, path = m.viterbi_search(x, "NACGT")
    np.testing.assert_array_equal(emit, path)
    assert len(seq) == len(path)


def _repeat_matrix():  # tests/test_decode.py:139-147
    x = np.zeros((20, 5), np.float32)
    x[:, 0] = 0.5
    for idx in (6, 13, 18):
        x[idx, 0] = 0.0
        x[idx, 1] = 1.0
    return x


def k8_beam_repeat(m):  # tests/test_decode.py:137-152
    assert m.beam_search(_repeat_matrix(), "NACGT", 5, 0.1) == ("AAA", [6, 13, 18])


def k8_viterbi_repeat(m):  # tests/test_decode.py:242-277
    assert m.viterbi_search(_repeat_matrix(), "NACGT") == ("AAA", [6, 13, 18])
    seq, path = m.viterbi_search(_repeat_matrix(), "NACGT", qstring=True)
    assert seq[len(path):] == "III" and seq[:len(path)] == "AAA" and path == [6, 13, 18]


def _multichar_matrix():  # tests/test_decode.py:157-166
    x = np.zeros((20, 5), np.float32)
    x[:, 0] = 0.5
    for a, idx in enumerate((6, 13, 18)):
        x[idx, 0] = 0.0
        x[idx, a + 1] = 1.0
    return x, ["N", "AAA", "CCC", "GGG", "TTTT"]


def k9_beam_multichar(m):  # tests/test_decode.py:154-171
    x, alpha = _multichar_matrix()
    assert m.beam_search(x, alpha, 5, 0.1) == ("AAACCCGGG", [6, 13, 18])


def k9_viterbi_multichar(m):  # tests/test_decode.py:321-338
    x, alpha = _multichar_matrix()
    assert m.viterbi_search(x, alpha) == ("AAACCCGGG", [6, 13, 18])


def k10_beam_spread(m):  # tests/test_decode.py:173-189
    x = np.zeros((20, 5), np.float32)
    x[:, 0] = 0.5
    for idx in (6, 13, 18):
        x[idx:idx + 3, 0] = 0.0
        x[idx:idx + 3, 1] = 1.0
    assert m.beam_search(x, "NACGT", 5, 0.1) == ("AAA", [6, 13, 18])


def k11_mean_qscores(m):  # tests/test_decode.py:279-319
    x = np.zeros((20, 5), np.float32)
    x[:, 0] = 0.5
    for r, c, v in [(3, 1, 0.99), (4, 1, 0.99), (6, 2, 0.999), (7, 2, 0.999), (9, 4, 0.6),
                    (10, 4, 0.7), (11, 4, 0.8), (13, 4, 0.4), (14, 4, 0.5), (15, 4, 0.6)]:
        x[r, 0] = 0.0
        x[r, c] = v
    seq, path = m.viterbi_search(x, "NACGT", qstring=True)
    assert seq[:len(path)] == "ACTT" and seq[len(path):] == "5?&$"


K12_MATRIX = np.array([  # tests/test_decode.py:342-352
    [0.7, 0.1, 0.2], [0.7, 0.1, 0.2], [0.2, 0.3, 0.5], [0.2, 0.2, 0.6], [0.3, 0.3, 0.4],
    [0.2, 0.2, 0.6], [0.2, 0.3, 0.5], [0.7, 0.1, 0.2], [0.7, 0.1, 0.2]], np.float32)


def k12_viterbi_off_path(m):  # tests/test_decode.py:340-355
    assert m.viterbi_search(K12_MATRIX, "NAB")[0] == "B"


K13_MATRIX = np.array([  # tests/test_decode.py:399-404
    [0.01, 0.98, 0.01], [0.01, 0.34, 0.65], [0.01, 0.98, 0.01], [0.01, 0.01, 0.98]], np.float32)


def k13_beam_abab(m):  # tests/test_decode.py:405
    assert m.beam_search(K13_MATRIX, "NAB")[0] == "ABAB"


K14_MATRIX = np.array([  # tests/test_decode.py:378-393
    [0.01, 0.98, 0.01], [0.01, 0.98, 0.01], [0.01, 0.98, 0.01], [0.01, 0.98, 0.01],
    [0.9, 0.05, 0.05], [0.7, 0.05, 0.35], [0.9, 0.05, 0.05],
    [0.01, 0.98, 0.01], [0.01, 0.98, 0.01], [0.01, 0.98, 0.01],
    [0.01, 0.01, 0.98], [0.01, 0.01, 0.98], [0.01, 0.01, 0.98], [0.01, 0.01, 0.98]], np.float32)


def k14_duplex_identical(m):  # tests/test_decode.py:376-395
    assert m.beam_search_duplex(K14_MATRIX, K14_MATRIX, "NAB") == "AAB"


def k15_duplex_disagreeing(m):  # tests/test_decode.py:397-412
    y = np.array([[0, 1, 0], [0, 1, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    assert m.beam_search_duplex(K13_MATRIX, y, "NAB") == "AB"


def k16_beam_nans(m):  # tests/test_decode.py:100-104
    x = np.full((100, 5), np.nan, np.float32)
    with pytest.raises(RuntimeError, match="Failed to compare values"):
        m.beam_search(x, "NACGT")


def k16_duplex_nans(m):  # tests/test_decode.py:370-374
    rng = np.random.default_rng(5)
    x1 = np.full((100, 5), np.nan, np.float32)
    x2 = rng.random((100, 5), dtype=np.float32)
    x2 /= np.linalg.norm(x2, ord=2, axis=1, keepdims=True)
    with pytest.raises(RuntimeError, match="Failed to compare values"):
        m.beam_search_duplex(x1, x2, "NACGT")


ONE_D_CASES = [k1_viterbi, k2_beam_wasm, k3_viterbi_blank_bounds, k4_beam_blank_bounds,
               k7_beam_path, k7_viterbi_path, k8_beam_repeat, k8_viterbi_repeat,
               k9_beam_multichar, k9_viterbi_multichar, k10_beam_spread, k11_mean_qscores,
               k12_viterbi_off_path, k13_beam_abab, k16_beam_nans]
CRF_CASES = [k6_crf_greedy, k6_crf_beam]
DUPLEX_CASES = [k14_duplex_identical, k15_duplex_disagreeing, k16_duplex_nans]


# ---- the API-shape / validation behaviours of tests/test_decode.py ---------------------------

def reference_style_rows(rng, samples, n):
    """tests/test_decode.py:15-17 with a seeded generator."""
    x = rng.random((samples, n), dtype=np.float32)
    return (x / np.linalg.norm(x, ord=2, axis=1, keepdims=True)).astype(np.float32)


def api_beam_search(m):  # tests/test_decode.py:8-120
    rng = np.random.default_rng(11)
    alphabet, beam_size, thr = "NACGT", 5, 0.1
    probs = reference_style_rows(rng, 100, 5)

    def ok(seq, path, n=4):
        assert len(seq) == len(path)
        assert len(set(seq)) == n

    ok(*m.beam_search(probs, alphabet, beam_size, thr))
    ok(*m.beam_search(probs, list(alphabet), beam_size, thr))
    ok(*m.beam_search(probs, tuple(alphabet), beam_size, thr))
    ok(*m.beam_search(network_output=probs, alphabet=alphabet, beam_size=beam_size,
                      beam_cut_threshold=thr))
    with pytest.raises(TypeError):
        m.beam_search(probs)
    ok(*m.beam_search(probs, alphabet))
    ok(*m.beam_search(probs, "NRUST", beam_size, thr))
    with pytest.raises(ValueError):
        m.beam_search(probs, alphabet, 0, thr)
    ok(*m.beam_search(probs, alphabet, beam_size, 0.0))
    with pytest.raises(ValueError):
        m.beam_search(probs, alphabet, beam_size, -0.1)
    with pytest.raises(ValueError):
        m.beam_search(probs, alphabet, beam_size, 1.0 / len(alphabet))
    with pytest.raises(ValueError):
        m.beam_search(probs, alphabet, beam_size, 1.1)
    with pytest.raises(ValueError):
        m.beam_search(probs, "NAGC", beam_size, thr)
    with pytest.raises(ValueError):
        m.beam_search(probs, "NAGCTX", beam_size, thr)
    short = reference_style_rows(rng, 100, 3)
    ok(*m.beam_search(short, "NAG", beam_size, thr), n=2)
    long_alpha = "NABCDEFGHIJK"
    longp = reference_style_rows(rng, 10000, len(long_alpha))
    ok(*m.beam_search(longp, long_alpha, beam_size, beam_cut_threshold=0.0), n=11)


def api_viterbi_search(m):  # tests/test_decode.py:192-225
    rng = np.random.default_rng(12)
    probs = reference_style_rows(rng, 100, 5)
    seq, path = m.viterbi_search(probs, "NACGT")
    assert len(seq) == len(path) and len(set(seq)) == 4
    seq, path = m.viterbi_search(probs, "NACGT", qstring=True)
    assert len(seq) == len(path) * 2
    with pytest.raises(TypeError):
        m.viterbi_search(probs)
    with pytest.raises(ValueError):
        m.viterbi_search(probs, "NACG")
    with pytest.raises(ValueError):
        m.viterbi_search(probs, "NACGTR")
