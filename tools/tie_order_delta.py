"""What does the tie order change, and what does following it cost?  (GPU.)

    python tools/tie_order_delta.py [2] [3] [4] [5] [--oracle N]

Decodes BASELINE configs 2 (4096 reads, beam 5), 3 (one rank's shard: 8192 reads, beam 32), 4 (CRF, 4096 reads) and 5
(1024 read pairs, both log-add modes) under both selectable orders of equal probabilities -- FCD_TIE_PDQ178 (default:
Rust 1.78's sort_unstable_by, csrc/pdq178.h) and FCD_TIE_STABLE (ascending node index) -- and prints one JSON line per
config: how many reads / pairs come out differently, how many were flagged by the tie counters, and the kernel time
under each order (HIP events on the launch stream).  --oracle N also decodes the first N differing reads / pairs with
the CPU oracle (test infrastructure, used here only as the checker) under each order and counts the matches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import bench
import fast_ctc_decode_amd as fcd


def timed(fn, reps=3):
    r = fn()
    torch.cuda.synchronize()
    h = r._handle
    h.timing_reset()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    ms, _ = h.timing_mean_ms()
    return r.cpu(), ms


def rows_differing(a, b, with_path=True):
    out = []
    for i in range(len(a.out_len)):
        n = int(a.out_len[i])
        same = int(a.status[i]) == int(b.status[i]) and n == int(b.out_len[i]) and \
            np.array_equal(a.labels[i, :n], b.labels[i, :n]) and \
            (not with_path or np.array_equal(a.path[i, :n], b.path[i, :n]))
        if not same:
            out.append(i)
    return out


def both_orders(run, reps):
    res = {}
    for order in ("stable", "pdq178"):
        fcd.set_tie_order(order)
        res[order] = timed(run, reps)
    fcd.set_tie_order("pdq178")
    return res


def main():
    n_oracle = 0
    args = sys.argv[1:]
    if "--oracle" in args:
        n_oracle = int(args[args.index("--oracle") + 1])
        del args[args.index("--oracle"):args.index("--oracle") + 2]
    which = [int(a) for a in args] or [2, 3, 4, 5]
    for cfg in which:
        if cfg in (2, 3):
            c = bench.CONFIGS[cfg]
            x = bench.make_batch(c["seed"], c["batch"])
            xd = torch.from_numpy(x).cuda()
            res = both_orders(lambda: fcd.beam_search_batch_raw(xd, c["beam"], c["thr"], True), 5)
            amb = fcd.beam_search_batch_raw(xd, c["beam"], c["thr"], True, count_ambiguous=True).cpu().ambiguous
            amb = np.asarray(amb).astype(np.int64)
            differ = rows_differing(res["stable"][0], res["pdq178"][0])
            # the same launch without its tied reads (each replaced by an untied one): what the tie bookkeeping costs
            # every step, apart from what replaying the quicksort costs the reads that need it
            flagged = np.flatnonzero(amb[:, 0] > 0)
            clean = np.flatnonzero(amb[:, 0] == 0)
            xq = xd.clone()
            xq[torch.from_numpy(flagged).cuda()] = xd[int(clean[0])]
            res_q = both_orders(lambda: fcd.beam_search_batch_raw(xq, c["beam"], c["thr"], True), 5)
            del xq
            out = {"config": cfg, "workload": "beam_search beam %d thr %g, %d reads T=4000 N=5" % (c["beam"], c["thr"], c["batch"]),
                   "tied_steps_per_flagged_read": {"max": int(amb[:, 0].max()), "mean": float(amb[flagged, 0].mean()) if len(flagged) else 0.0,
                                                   "reads_tied_on_more_than_half_their_steps": int((amb[:, 0] > 2000).sum())},
                   "kernel_ms_without_the_tied_reads": {o: res_q[o][1] for o in res_q},
                   "reads_with_gt20_candidate_kept_tie": int((amb[:, 0] > 0).sum()),
                   "reads_with_result_changing_tie": int((amb[:, 1] > 0).sum()),
                   "reads_differing_between_orders": len(differ), "differing_reads": differ[:64],
                   "all_differing_reads_flagged_by_both_counters": bool(all(amb[i, 0] > 0 and amb[i, 1] > 0 for i in differ)),
                   "kernel_ms": {o: res[o][1] for o in res},
                   "reads_per_s": {o: c["batch"] / res[o][1] * 1e3 for o in res}}
            if n_oracle:
                from oracle import oracle
                ok = {}
                for order, mode in (("stable", "stable"), ("pdq178", "pdqsort")):
                    r = res[order][0]
                    good = 0
                    with oracle.unstable_sort(mode):
                        for i in differ[:n_oracle]:
                            st, labels, path, _ = oracle.beam_search_raw(x[i], c["beam"], c["thr"], True)
                            n = int(r.out_len[i])
                            good += int(st == int(r.status[i]) and np.array_equal(r.labels[i, :n], labels) and
                                        np.array_equal(r.path[i, :n], path))
                    ok[order] = good
                out["differing_reads_equal_to_oracle_under_same_order"] = {"checked": min(n_oracle, len(differ)), **ok}
            print(json.dumps(out), flush=True)
            del xd, x
        if cfg == 4:
            B = 4096
            g = torch.Generator(device="cuda")
            g.manual_seed(3)
            x = torch.rand((B, 4000, 4, 5), generator=g, device="cuda", dtype=torch.float32)
            x = x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)
            init = torch.zeros((B, 4), device="cuda")
            init[torch.arange(B), torch.arange(B) % 4] = 1.0
            res = both_orders(lambda: fcd.crf_beam_search_batch_raw(x, init, 5, 0.0), 3)
            differ = rows_differing(res["stable"][0], res["pdq178"][0])
            print(json.dumps({"config": 4, "workload": "crf_beam_search beam 5 thr 0.0, %d reads T=4000 S=4 N=5" % B,
                              "reads_differing_between_orders": len(differ), "differing_reads": differ[:64],
                              "kernel_ms": {o: res[o][1] for o in res}}), flush=True)
            del x
        if cfg == 5:
            B, T, w = 1024, 2000, 64
            from bench_configs import rows
            x1, x2 = rows((B, T, 5), 4), rows((B, T, 5), 5)
            i = np.arange(T)
            env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
            envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
            for mode, name in ((0, "logsumexp"), (1, "max")):
                res = both_orders(lambda: fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode), 1)
                amb = fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode,
                                                      count_ambiguous=True).cpu().ambiguous
                amb = np.asarray(amb).astype(np.int64)
                differ = rows_differing(res["stable"][0], res["pdq178"][0], with_path=False)
                out = {"config": 5, "workload": "beam_search_duplex %s beam 5 thr 0.1, %d pairs T1=T2=%d band +-%d" % (name, B, T, w),
                       "pairs_with_gt20_candidate_kept_tie": int((amb[:, 0] > 0).sum()),
                       "pairs_with_result_changing_tie": int((amb[:, 1] > 0).sum()),
                       "pairs_differing_between_orders": len(differ), "differing_pairs": differ[:64],
                       "kernel_ms": {o: res[o][1] for o in res}}
                if n_oracle:
                    from oracle import oracle
                    h1, h2 = x1.cpu().numpy(), x2.cpu().numpy()
                    ok = {}
                    sample = (differ + [p for p in range(B) if p not in set(differ)])[:n_oracle]
                    for order, om in (("stable", "stable"), ("pdq178", "pdqsort")):
                        r = res[order][0]
                        good = 0
                        with oracle.unstable_sort(om):
                            for p in sample:
                                want = oracle.beam_search_duplex(h1[p], h2[p], "NACGT", env, 5, 0.1, True,
                                                                 (oracle.MAXMODE if mode else oracle.LOGSUMEXP) | oracle.MATH_CR)
                                got = "".join("NACGT"[l] for l in r.labels[p, :int(r.out_len[p])])
                                good += int(got == want)
                        ok[order] = good
                    out["sample_equal_to_oracle_under_same_order"] = {"checked": len(sample), **ok}
                print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
