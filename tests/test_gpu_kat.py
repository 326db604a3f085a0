"""The reference's known-answer tests run against the HIP product through the C ABI."""
import numpy as np
import pytest

import kat_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


@pytest.mark.parametrize("case", kat_cases.ONE_D_CASES + kat_cases.CRF_CASES,
                         ids=lambda f: f.__name__)
def test_kat_1d(fcd, case):
    case(fcd)


def test_api_shape(fcd):
    kat_cases.api_beam_search(fcd)
    kat_cases.api_viterbi_search(fcd)
