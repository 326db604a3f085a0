"""Long run of the oracle-vs-naive-restatement cross-check (tests/test_naive_crosscheck.py, tests/naive_reference.py):

    python tools/naive_crosscheck.py [cases_per_search]      (CPU only)

Prints one JSON line with the number of cases per search, how many decoded / failed / hit a reference panic, and
the number of disagreements (must be 0)."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_naive_crosscheck as X


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    out = {}
    X.oracle.lib.fcdo_set_unstable_sort(0)  # the naive restatement sorts stably (tests/test_naive_crosscheck.py)
    runs = [("duplex logsumexp", lambda s: X.check_duplex(s, False), 10 ** 6),
            ("duplex max", lambda s: X.check_duplex(s, True), 10 ** 6),
            ("crf duplex logsumexp", lambda s: X.check_crf_duplex(s, False), 2 * 10 ** 6),
            ("crf duplex max", lambda s: X.check_crf_duplex(s, True), 2 * 10 ** 6),
            ("crf_beam_search", X.check_crf_1d, 3 * 10 ** 6),
            ("beam_search", lambda s: X.check_beam_1d(s, False), 6 * 10 ** 6),
            ("beam_search, special posteriors", lambda s: X.check_beam_1d(s, True), 6 * 10 ** 6),
            ("duplex logsumexp, special posteriors", lambda s: X.check_duplex_special(s, False), 4 * 10 ** 6),
            ("duplex max, special posteriors", lambda s: X.check_duplex_special(s, True), 4 * 10 ** 6),
            ("crf duplex logsumexp, special posteriors", lambda s: X.check_crf_duplex_special(s, False), 5 * 10 ** 6),
            ("crf duplex max, special posteriors", lambda s: X.check_crf_duplex_special(s, True), 5 * 10 ** 6)]
    for name, fn, base in runs:
        kinds = collections.Counter()
        bad = 0
        for seed in range(base, base + n):
            try:
                o = fn(seed)
            except AssertionError as e:
                bad += 1
                print("MISMATCH", name, e, file=sys.stderr)
                continue
            o = o if isinstance(o, str) else "ok"
            kinds["panic" if o == "error: panic" else "error" if o.startswith("error") else "decoded"] += 1
        out[name] = {"cases": n, **kinds, "disagreements": bad}
    print(json.dumps(out))
    return 1 if any(v["disagreements"] for v in out.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
