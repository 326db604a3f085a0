"""developer scratch: reproduce the lane-kernel / pdq178 mismatch on the GPU and narrow it down"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import fast_ctc_decode_amd as fcd
from oracle import oracle
from tie_util import tie_order
import test_gpu_tieorder as TO


def cmp(x, beam, thr, collapse, lengths, kernel, tag):
    r = fcd.beam_search_batch_raw(x, beam, thr, collapse, lengths=lengths, kernel=kernel)
    bad = []
    for i in range(x.shape[0]):
        xi = x[i] if lengths is None else x[i, :lengths[i]]
        st, labels, path, _ = oracle.beam_search_raw(np.ascontiguousarray(xi), beam, thr, collapse)
        n = int(r.out_len[i])
        ok = int(r.status[i]) == st and (st != 0 or (n == len(labels) and np.array_equal(r.labels[i, :n], labels) and np.array_equal(r.path[i, :n], path)))
        if not ok:
            # first differing position
            m = min(n, len(labels))
            d = next((j for j in range(m) if r.labels[i, j] != labels[j] or r.path[i, j] != path[j]), m)
            bad.append((i, n, len(labels), d, int(path[d]) if d < len(path) else -1))
    print(tag, "bad:", bad, flush=True)


x = TO.quantised(900 + 5 + 32, 6, 160, 5)
x[:, :, 0] = np.maximum(x[:, :, 0], 0.25)
lengths = np.array([160, 159, 1, 0, 100, 33], np.int64)
for order in ("pdq178", "stable"):
    with tie_order(fcd, order):
        cmp(x, 32, 0.1, False, lengths, 4, order + " lane ragged")
        cmp(x, 32, 0.1, False, None, 4, order + " lane full")
        cmp(x, 32, 0.1, True, None, 4, order + " lane full collapse")
        cmp(x[1:2], 32, 0.1, False, None, 4, order + " lane read1 alone")
        cmp(x[:2], 32, 0.1, False, None, 4, order + " lane reads 0,1")
        cmp(x, 33, 0.1, False, lengths, 4, order + " lane beam 33 (one read per wavefront)")
        cmp(x, 32, 0.1, False, lengths, 1, order + " generic ragged")
        cmp(x, 32, 0.05, False, lengths, 4, order + " lane thr 0.05")
