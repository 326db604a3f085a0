"""Regenerates tests/golden/vectors.npz: seeded inputs and the outputs the CPU oracle gives for them.

The reference itself (Rust) cannot run in the build image, so these vectors come from the oracle
(oracle/fcd_oracle.c), which is pinned by every known-answer test the reference holds
(tests/test_oracle_kat.py).  The reference's own golden vectors are restated, with citations, in
tests/kat_cases.py.  The file is data only: inputs + expected outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kat_cases import reference_style_rows  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    rng = np.random.default_rng(20260926)
    out = {}

    def add_beam(name, x, beam, thr, collapse):
        st, labels, path, _ = oracle.beam_search_raw(x, beam, thr, collapse)
        out[name + "/x"] = x
        out[name + "/args"] = np.array([beam, thr, collapse], np.float64)
        out[name + "/status"] = np.array([st])
        out[name + "/labels"] = labels.astype(np.uint8)
        out[name + "/path"] = path.astype(np.uint32)

    add_beam("beam_n5", reference_style_rows(rng, 300, 5), 5, 0.1, True)
    add_beam("beam_n5_thr0_nocollapse", reference_style_rows(rng, 200, 5), 5, 0.0, False)
    add_beam("beam_n3_b2", reference_style_rows(rng, 250, 3), 2, 0.1, True)
    add_beam("beam_n7_b8", reference_style_rows(rng, 200, 7), 8, 0.05, True)
    add_beam("beam_n12_b32", reference_style_rows(rng, 150, 12), 32, 0.02, True)
    xz = reference_style_rows(rng, 80, 5)
    xz[40:] = 0.0
    add_beam("beam_runs_out", xz, 5, 0.1, True)

    x = reference_style_rows(rng, 500, 5)
    labels, path, quals = oracle.viterbi_search_raw(x, True)
    out["viterbi/x"], out["viterbi/labels"] = x, labels.astype(np.uint8)
    out["viterbi/path"], out["viterbi/quals"] = path.astype(np.uint32), quals

    xc = rng.random((300, 4, 5), dtype=np.float32)
    xc /= xc.sum(-1, keepdims=True)
    init = np.array([0, 0, 1, 0], np.float32)
    seq, path = oracle.crf_beam_search(xc.astype(np.float32), init, "NACGT", 5, 0.0)
    out["crf/x"], out["crf/init"] = xc.astype(np.float32), init
    out["crf/seq"], out["crf/path"] = np.frombuffer(seq.encode(), np.uint8), np.array(path, np.uint32)

    x1, x2 = reference_style_rows(rng, 150, 5), reference_style_rows(rng, 140, 5)
    i = np.arange(150)
    env = np.stack([np.maximum(0, i - 20), np.minimum(140, i + 20)], 1).astype(np.uint64)
    out["duplex/x1"], out["duplex/x2"], out["duplex/env"] = x1, x2, env
    for mode, name in ((oracle.LOGSUMEXP | oracle.MATH_CR, "logsumexp_cr"), (oracle.MAXMODE | oracle.MATH_CR, "max_cr"),
                       (oracle.LOGSUMEXP, "logsumexp_libm")):
        s = oracle.beam_search_duplex(x1, x2, "NACGT", env, 5, 0.1, True, mode)
        out["duplex/" + name] = np.frombuffer(s.encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "vectors.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
