"""Writes tools/verify/pdq178_vectors.json: lists of (probability bits, node) pairs -- what the beam searches hand
`sort_unstable_by` (src/search.rs:262-269, src/duplex.rs:620) -- each with the permutation THIS repository's restatement
of Rust 1.78's pattern-defeating quicksort produces (oracle/fcd_oracle.c, DEFINE_PDQSORT; the kernels' csrc/pdq178.h and
csrc/pdq178_wave.h are tested element for element against it).

Nobody in this image could run rustc; a rustc-1.65 build of std found compiled in it pins everything but the 2023 forms
of two routines (tools/verify/rust165_pdqsort.py).  This file and pdq178_check.rs turn the rest into
one command for whoever has the reference's toolchain:

    rustc +1.78.0 -O tools/verify/pdq178_check.rs -o /tmp/pdq178_check && /tmp/pdq178_check tools/verify/pdq178_vectors.json

    python tools/verify/make_pdq178_vectors.py          # regenerates the file (deterministic)

Format: {"meta": {...}, "cases": [{"bits": [u32 ...], "perm": [int ...] (, "perm_g" / "perm_p" / "perm_gp": [int ...])}, ...]}
(the optional ones: the same list under the earlier forms of the two routines std changed in 2023, see meta.alternatives
and tools/verify/rust165_pdqsort.py); the list handed to the sort is
[(f32::from_bits(bits[i]), node = i) for i in 0..n] -- already in ascending node order, like the reference's list after
its stable sort by node (src/search.rs:245) -- and perm[j] = the node the sorted list holds at position j."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from test_pdqsort_restatement import _patterns  # noqa: E402


def lists():
    rng = np.random.default_rng(178)
    # the lengths where pdqsort changes gear: 20 / 21 (insertion sort), 50 (ninther, shifting), 128-element blocks
    for n in list(range(21, 71, 3)) + [100, 127, 128, 129, 160, 255, 256, 257, 258, 320, 511, 512]:
        for p in _patterns(rng, n):
            yield np.ascontiguousarray(p, np.float32)
    # beam-like: mostly distinct probabilities with groups of equal ones ("twin" prefixes give equal children)
    for _ in range(900):
        n = int(rng.integers(21, 161))
        k = int(rng.integers(1, 12))
        vals = rng.random(k, dtype=np.float32)
        p = vals[rng.integers(0, k, n)]
        if rng.random() < 0.6:
            p = np.where(rng.random(n) < 0.3, p, rng.random(n, dtype=np.float32)).astype(np.float32)
        yield np.ascontiguousarray(p, np.float32)
    for _ in range(250):  # twins proper: every value twice, in node order or shuffled
        n = int(rng.integers(22, 161))
        h = rng.random((n + 1) // 2, dtype=np.float32)
        p = np.repeat(h, 2)[:n].copy()
        if rng.random() < 0.5:
            p = p[rng.permutation(n)]
        yield np.ascontiguousarray(p, np.float32)
    for _ in range(200):  # nearly sorted either way (partial_insertion_sort, the reversal), a few displaced
        n = int(rng.integers(21, 300))
        p = np.sort(rng.random(n, dtype=np.float32))
        if rng.random() < 0.5:
            p = p[::-1].copy()
        q = np.round(p * 16) / 16 if rng.random() < 0.4 else p
        q = q.astype(np.float32).copy()
        for _ in range(int(rng.integers(0, 6))):
            i, j = rng.integers(0, n, 2)
            q[i], q[j] = q[j], q[i]
        yield np.ascontiguousarray(q, np.float32)
    for _ in range(120):  # long lists with few distinct values (partition_equal, bad partitions -> break_patterns, heapsort)
        n = int(rng.integers(161, 513))
        k = int(rng.integers(1, 5))
        vals = rng.random(k, dtype=np.float32)
        yield np.ascontiguousarray(vals[rng.integers(0, k, n)], np.float32)


def main():
    cases = []
    differs = 0
    alts = {"perm_g": 0, "perm_p": 0, "perm_gp": 0}
    for p in lists():
        n = len(p)
        with oracle.unstable_sort("pdqsort"):
            _, perm = oracle.pdqsort_desc(p, np.arange(n, dtype=np.int32))
        assert sorted(perm.tolist()) == list(range(n))
        sp = p[perm]
        assert np.all(sp[:-1] >= sp[1:])
        differs += int(not np.array_equal(perm, np.argsort(-p, kind="stable")))
        case = {"bits": p.view(np.uint32).tolist(), "perm": perm.tolist()}
        # the same list under the EARLIER forms of the two routines std changed in 2023 (fcd_oracle.c,
        # fcdo_set_pdq_std_form): given only where the permutation differs
        for key, form in (("perm_g", 1), ("perm_p", 2), ("perm_gp", 3)):
            with oracle.unstable_sort("pdqsort"), oracle.pdq_std_form(form):
                _, alt = oracle.pdqsort_desc(p, np.arange(n, dtype=np.int32))
            if not np.array_equal(alt, perm):
                case[key] = alt.tolist()
                alts[key] += 1
        cases.append(case)
    meta = {
        "what": "lists handed to sort_unstable_by(|a, b| b.probability().partial_cmp(&a.probability())...) and the permutation "
                "this repository's restatement of Rust 1.78.0's core::slice::sort::quicksort produces",
        "reference": "nanoporetech/fast-ctc-decode src/search.rs:262-269, src/duplex.rs:620; toolchain 1.78.0 (.github/workflows/test.yml:16)",
        "check": "rustc +1.78.0 -O tools/verify/pdq178_check.rs -o /tmp/pdq178_check && /tmp/pdq178_check tools/verify/pdq178_vectors.json",
        "cases": len(cases),
        "cases_where_the_order_differs_from_a_stable_sort": differs,
        "generator": "tools/verify/make_pdq178_vectors.py (numpy default_rng(178); oracle/fcd_oracle.c DEFINE_PDQSORT)",
        "alternatives": "perm = Rust 1.78 as recalled (the kernels' order).  Two routines of std's pdqsort changed in 2023; where "
                        "their EARLIER forms give another permutation it is listed too: perm_g = break_patterns drawing two "
                        "32-bit xorshift numbers (13, 17, 5) per usize, perm_p = partial_insertion_sort calling shift_tail(&mut "
                        "v[..i]) and shift_head(&mut v[i..]), perm_gp = both -- which is what the compiled rustc-1.65 std in this "
                        "image produces on every list (tools/verify/rust165_pdqsort.py).  pdq178_check.rs says which one "
                        "your toolchain agrees with",
        "cases_with_alternatives": alts,
    }
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pdq178_vectors.json")
    with open(out, "w") as f:
        f.write('{"meta": %s,\n "cases": [\n' % json.dumps(meta))
        for i, c in enumerate(cases):
            extra = "".join(', "%s": %s' % (k, json.dumps(c[k], separators=(",", ":"))) for k in ("perm_g", "perm_p", "perm_gp") if k in c)
            f.write('  {"bits": %s, "perm": %s%s}%s\n' % (json.dumps(c["bits"], separators=(",", ":")),
                                                        json.dumps(c["perm"], separators=(",", ":")), extra,
                                                        "," if i + 1 < len(cases) else ""))
        f.write(" ]}\n")
    print(out, len(cases), "cases,", differs, "differ from the stable order,", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
