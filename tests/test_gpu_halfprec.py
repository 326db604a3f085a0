"""Half-precision posteriors read directly by the kernels (fcd_batch.dtype): float16 / bfloat16 convert to float32
exactly, so every search on half-precision input must return EXACTLY what it returns on the upcast float32
matrix -- the reference's result on that matrix -- for every kernel family, host and device inputs."""
import numpy as np
import pytest

from test_gpu_parity import gen_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fcd():
    import fast_ctc_decode_amd as m
    return m


def _same(a, b):
    a, b = a.cpu(), b.cpu()
    assert np.array_equal(a.out_len, b.out_len) and np.array_equal(a.status, b.status)
    for i in range(len(a.out_len)):
        n = int(a.out_len[i])
        assert np.array_equal(a.labels[i, :n], b.labels[i, :n]), i
        if a.path is not None:
            assert np.array_equal(a.path[i, :n], b.path[i, :n]), i
        if a.qual is not None:
            assert np.array_equal(a.qual[i, :n].view(np.uint32), b.qual[i, :n].view(np.uint32)), i


def _bf16_bits(x):
    """float32 -> (bfloat16 bit patterns as uint16, the same values as float32): truncation is enough for a test"""
    u = x.view(np.uint32) >> 16
    return u.astype(np.uint16), (u.astype(np.uint32) << 16).view(np.float32)


def test_half_precision_host_inputs_every_kernel(fcd):
    x32 = gen_batch(41, 9, 333, 5)
    x32[2, 10] = np.float32(6e-8)        # a float16 subnormal survives the trip
    lengths = np.array([333, 0, 1, 200, 333, 64, 65, 257, 300], np.int64)
    h16 = x32.astype(np.float16)
    up16 = h16.astype(np.float32)
    b16, upb = _bf16_bits(x32)
    for half, up, kw in ((h16, up16, {}), (b16, upb, {"input_dtype": "bfloat16"})):
        for kernel, beam in ((0, 5), (1, 5), (3, 7), (4, 24)):
            _same(fcd.beam_search_batch_raw(half, beam, 0.1, True, lengths=lengths, kernel=kernel, **kw),
                  fcd.beam_search_batch_raw(up, beam, 0.1, True, lengths=lengths, kernel=kernel))
        _same(fcd.viterbi_search_batch_raw(half, True, lengths=lengths, qual=True, **kw),
              fcd.viterbi_search_batch_raw(up, True, lengths=lengths, qual=True))
        strided = half[:, ::2, :]          # not C-contiguous: the strided viterbi kernel
        _same(fcd.viterbi_search_batch_raw(strided, True, qual=True, **kw),
              fcd.viterbi_search_batch_raw(np.ascontiguousarray(up[:, ::2, :]), True, qual=True))
    # N = 4 and 7: other tile shapes of the streaming kernel (odd N: a half-used last load)
    for N in (4, 7, 3):
        y = gen_batch(50 + N, 5, 700, N)
        y16 = y.astype(np.float16)
        _same(fcd.viterbi_search_batch_raw(y16, True, qual=True),
              fcd.viterbi_search_batch_raw(y16.astype(np.float32), True, qual=True))
    # the compiled module's batch functions take float16 arrays too
    assert fcd.beam_search_batch(h16, "NACGT", 5, 0.1, lengths=lengths) == \
        fcd.beam_search_batch(up16, "NACGT", 5, 0.1, lengths=lengths)
    with pytest.raises(TypeError):
        fcd.beam_search(h16[0], "NACGT")   # the per-read surface keeps the reference's float32-only rule


def test_half_precision_crf_and_duplex(fcd):
    rng = np.random.default_rng(3)
    x4 = rng.random((4, 90, 4, 5), dtype=np.float32)
    init = rng.random((4, 4), dtype=np.float32)
    h = x4.astype(np.float16)
    up = h.astype(np.float32)
    _same(fcd.crf_beam_search_batch_raw(h, init, 5, 0.0), fcd.crf_beam_search_batch_raw(up, init, 5, 0.0))
    _same(fcd.crf_greedy_search_batch_raw(h, init, qual=True), fcd.crf_greedy_search_batch_raw(up, init, qual=True))
    x1, x2 = gen_batch(7, 3, 80, 5), gen_batch(8, 3, 70, 5)
    h1, h2 = x1.astype(np.float16), x2.astype(np.float16)
    _same(fcd.beam_search_duplex_batch_raw(h1, h2, None, 5, 0.1),
          fcd.beam_search_duplex_batch_raw(h1.astype(np.float32), h2.astype(np.float32), None, 5, 0.1))


def test_half_precision_device_tensors_full_length(fcd):
    """torch float16 / bfloat16 tensors at BASELINE row count: no upcast pass, same results as the upcast input."""
    torch = pytest.importorskip("torch")
    x = torch.from_numpy(gen_batch(9, 64, 4000, 5)).cuda()
    for dt in (torch.float16, torch.bfloat16):
        xh = x.to(dt)
        up = xh.float()
        _same(fcd.viterbi_search_batch_raw(xh, True, qual=True), fcd.viterbi_search_batch_raw(up, True, qual=True))
        _same(fcd.beam_search_batch_raw(xh, 5, 0.1, True), fcd.beam_search_batch_raw(up, 5, 0.1, True))
        _same(fcd.beam_search_batch_raw(xh, 32, 0.1, True), fcd.beam_search_batch_raw(up, 32, 0.1, True))
