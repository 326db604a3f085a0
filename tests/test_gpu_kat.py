"""The reference's known-answer tests run against the HIP product through the C ABI."""
import numpy as np
import pytest

import kat_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["python-mirror", "compiled-module"])
def fcd(request):
    """Both host layers over the C ABI: the Python mirror and the compiled `fast_ctc_decode`
    module (csrc/pymodule.cpp), the drop-in import name."""
    if request.param == "python-mirror":
        import fast_ctc_decode_amd as m
    else:
        import fast_ctc_decode as m
    return m


@pytest.mark.parametrize("case", kat_cases.ONE_D_CASES + kat_cases.CRF_CASES,
                         ids=lambda f: f.__name__)
def test_kat_1d(fcd, case):
    case(fcd)


def test_api_shape(fcd):
    kat_cases.api_beam_search(fcd)
    kat_cases.api_viterbi_search(fcd)


@pytest.mark.parametrize("case", kat_cases.DUPLEX_CASES, ids=lambda f: f.__name__)
def test_kat_duplex_default_mode(fcd, case):
    case(fcd)


def test_compiled_module_matches_mirror_on_random_reads():
    import fast_ctc_decode as compiled
    import fast_ctc_decode_amd as mirror
    rng = np.random.default_rng(21)
    for n, alpha in ((5, "NACGT"), (3, "NAB"), (12, "NABCDEFGHIJK")):
        x = kat_cases.reference_style_rows(rng, 300, n)
        for beam, thr in ((5, 0.0), (5, 0.05), (32, 0.05)):
            assert compiled.beam_search(x, alpha, beam, thr) == mirror.beam_search(x, alpha, beam, thr)
        assert compiled.viterbi_search(x, alpha, True) == mirror.viterbi_search(x, alpha, True)
