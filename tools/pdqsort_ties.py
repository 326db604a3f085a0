"""What would Rust 1.78's pdqsort change?  (CPU only.)

    python tools/pdqsort_ties.py [reads_config2] [reads_config3] [threads]

Runs the oracle on BASELINE config 2 (beam 5) and on a config-3 shard (beam 32) twice -- with the stable tie
rule (the oracle's default and the kernels' rule) and with the order its restatement of Rust 1.78's
sort_unstable_by leaves above 20 candidates (oracle/fcd_oracle.c: written from memory, pinned since round 5 against a
compiled rustc-1.65 std except for two routines std changed in 2023) -- and counts
the reads whose (labels, path) differ, next to the tie counters that say which reads COULD differ."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from oracle import oracle


def run(x, beam, thr, threads):
    n, T = x.shape[0], x.shape[1]
    out = oracle.batch_outputs(n, T)
    amb = np.zeros((n, 2), np.int64)
    labels, path, lens, status = oracle.beam_search_batch(x, beam, thr, True, threads, out=out, ambiguous=amb)
    return labels.copy(), path.copy(), lens.copy(), status.copy(), amb


def compare(name, x, beam, thr, threads):
    t0 = time.perf_counter()
    a = run(x, beam, thr, threads)
    with oracle.unstable_sort("pdqsort"):
        b = run(x, beam, thr, threads)
    n = x.shape[0]
    differ = []
    for i in range(n):
        L = int(a[2][i])
        same = a[3][i] == b[3][i] and L == int(b[2][i]) and np.array_equal(a[0][i, :L], b[0][i, :L]) \
            and np.array_equal(a[1][i, :L], b[1][i, :L])
        if not same:
            differ.append(i)
    amb = a[4]
    rec = {
        "config": name, "reads": n, "beam_size": beam, "beam_cut_threshold": thr,
        "reads_with_gt20_candidate_kept_tie": int((amb[:, 0] > 0).sum()),
        "reads_with_result_changing_tie": int((amb[:, 1] > 0).sum()),
        "reads_with_both": int(((amb[:, 0] > 0) & (amb[:, 1] > 0)).sum()),
        "reads_differing_under_pdqsort_restatement": len(differ),
        "differing_reads": differ[:64],
        "differing_reads_all_flagged_by_both_counters": bool(all(amb[i, 0] > 0 and amb[i, 1] > 0 for i in differ)),
        "counters_identical_in_both_modes": bool(np.array_equal(amb, b[4])),
        "seconds": round(time.perf_counter() - t0, 1),
        "note": "pdqsort restatement written from memory of Rust 1.78 library/core/src/slice/sort.rs; pinned against a compiled rustc-1.65 std except for two routines std changed in 2023 (tools/verify/rust165_pdqsort.py)",
    }
    print(json.dumps(rec), flush=True)
    return rec


def main():
    n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n3 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    compare("2 (seed 1, first %d reads)" % n2, bench.make_batch(1, n2), 5, 0.1, threads)
    compare("3 (seed 2, first %d reads of rank 0's shard)" % n3, bench.make_batch(2, n3), 32, 0.1, threads)


if __name__ == "__main__":
    main()
