"""Committed golden vectors (tests/golden/vectors.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU (-m gpu): the HIP product reproduces them through the
C ABI without needing the oracle at all."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.npz"))
BEAM_CASES = sorted({k.split("/")[0] for k in G.files if k.startswith("beam_")})


@pytest.mark.parametrize("name", BEAM_CASES)
def test_oracle_reproduces_beam_vectors(name):
    from oracle import oracle
    beam, thr, collapse = G[name + "/args"]
    st, labels, path, _ = oracle.beam_search_raw(G[name + "/x"], int(beam), float(thr), bool(collapse))
    assert st == int(G[name + "/status"][0])
    np.testing.assert_array_equal(labels, G[name + "/labels"])
    np.testing.assert_array_equal(path, G[name + "/path"])


def test_oracle_reproduces_other_vectors():
    from oracle import oracle
    labels, path, quals = oracle.viterbi_search_raw(G["viterbi/x"], True)
    np.testing.assert_array_equal(labels, G["viterbi/labels"])
    np.testing.assert_array_equal(quals, G["viterbi/quals"])
    seq, path = oracle.crf_beam_search(G["crf/x"], G["crf/init"], "NACGT", 5, 0.0)
    assert seq.encode() == G["crf/seq"].tobytes() and path == G["crf/path"].tolist()
    for mode, name in ((oracle.LOGSUMEXP | oracle.MATH_CR, "logsumexp_cr"), (oracle.MAXMODE | oracle.MATH_CR, "max_cr"),
                       (oracle.LOGSUMEXP, "logsumexp_libm")):
        s = oracle.beam_search_duplex(G["duplex/x1"], G["duplex/x2"], "NACGT", G["duplex/env"], 5, 0.1, True, mode)
        assert s.encode() == G["duplex/" + name].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
@pytest.mark.parametrize("name", BEAM_CASES)
def test_gpu_reproduces_beam_vectors(name, kernel):
    import fast_ctc_decode_amd as fcd
    from fast_ctc_decode_amd import _native as nat
    beam, thr, collapse = G[name + "/args"]
    x = G[name + "/x"]
    try:
        r = fcd.beam_search_batch_raw(x[None], int(beam), float(thr), bool(collapse), kernel=kernel)
    except nat.NativeError:
        assert kernel in (2, 3) and (beam > 8 or x.shape[1] > 7)  # shape outside the register kernels
        return
    n = int(r.out_len[0])
    assert int(r.status[0]) == int(G[name + "/status"][0])
    if int(r.status[0]) == 0:
        np.testing.assert_array_equal(r.labels[0, :n], G[name + "/labels"])
        np.testing.assert_array_equal(r.path[0, :n], G[name + "/path"])


@pytest.mark.gpu
def test_gpu_reproduces_other_vectors():
    import fast_ctc_decode_amd as fcd
    from fast_ctc_decode_amd import _native as nat
    seq, path = fcd.viterbi_search(G["viterbi/x"], "NACGT", qstring=True)
    n = len(path)
    assert seq[:n].encode() == bytes(b"NACGT"[l] for l in G["viterbi/labels"])
    assert path == G["viterbi/path"].tolist()
    assert [ord(c) for c in seq[n:]] == G["viterbi/quals"].tolist()
    assert fcd.crf_beam_search(G["crf/x"], G["crf/init"], "NACGT", 5, 0.0) == \
        (G["crf/seq"].tobytes().decode(), G["crf/path"].tolist())
    for mode, name in ((nat.LOGADD_LOGSUMEXP, "logsumexp_cr"), (nat.LOGADD_MAX, "max_cr")):
        s = fcd.beam_search_duplex(G["duplex/x1"], G["duplex/x2"], "NACGT", G["duplex/env"], 5, 0.1, True,
                                   logadd_mode=mode)
        assert s.encode() == G["duplex/" + name].tobytes()
