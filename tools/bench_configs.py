"""Secondary measurements for BASELINE.json configs 3-5 (bench.py covers config 2, the metric's
config).  One JSON line per config: kernel-only time from the C ABI's HIP events.

    python tools/bench_configs.py [3] [4] [5] [--check]

--check also decodes the first reads / pairs of every config with the CPU oracle (test infrastructure,
used here only as the checker) and reports how many of them the GPU reproduced exactly.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import fast_ctc_decode_amd as fcd


def rows(shape, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.rand(shape, generator=g, device="cuda", dtype=torch.float32)
    return x / torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True)


def timed(fn, reps=3):
    r = fn()
    torch.cuda.synchronize()
    h = r._handle
    h.timing_reset()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    ms, n = h.timing_mean_ms()
    return r, ms


def check_beam(x, r, n, beam, thr):
    from oracle import oracle
    xs, rc, same = x[:n].cpu().numpy(), r.cpu(), 0
    for i in range(n):
        st, labels, path, _ = oracle.beam_search_raw(xs[i], beam, thr, True)
        k = int(rc.out_len[i])
        same += int(st == int(rc.status[i]) and np.array_equal(rc.labels[i, :k], labels)
                    and np.array_equal(rc.path[i, :k], path))
    return {"oracle_checked": n, "oracle_identical": same}


def check_crf(x, init, r, n, beam, thr):
    from oracle import oracle
    xs, ini, rc, same = x[:n].cpu().numpy(), init[:n].cpu().numpy(), r.cpu(), 0
    for i in range(n):
        seq, path = oracle.crf_beam_search(xs[i], ini[i], "NACGT", beam, thr)
        k = int(rc.out_len[i])
        same += int("".join("NACGT"[l] for l in rc.labels[i, :k]) == seq and rc.path[i, :k].tolist() == path)
    return {"oracle_checked": n, "oracle_identical": same}


def check_duplex(x1, x2, env, r, n, mode):
    from oracle import oracle
    a, b, rc, same = x1[:n].cpu().numpy(), x2[:n].cpu().numpy(), r.cpu(), 0
    omode = (oracle.LOGSUMEXP if mode == 0 else oracle.MAXMODE) | oracle.MATH_CR
    for i in range(n):
        seq = oracle.beam_search_duplex(a[i], b[i], "NACGT", env, 5, 0.1, True, omode)
        same += int("".join("NACGT"[l] for l in rc.labels[i, :int(rc.out_len[i])]) == seq)
    return {"oracle_checked": n, "oracle_identical": same}


def _est(x1, x2):
    env = fcd.estimate_envelope_batch(x1, x2, 64)
    env._handle = fcd.api.nat.default_handle()
    return env


def main():
    check = "--check" in sys.argv
    which = [int(a) for a in sys.argv[1:] if a != "--check"] or [3, 4, 5]
    if 3 in which:  # beam 32, 8192 reads per GPU (65536 over 8 GPUs)
        B = 8192
        x = rows((B, 4000, 5), 2)
        r, ms = timed(lambda: fcd.beam_search_batch_raw(x, 32, 0.1, True), reps=2)
        out = {"config": 3, "workload": "beam_search beam 32 thr 0.1, %d reads/GPU T=4000 N=5" % B,
               "kernel_ms": ms, "reads_per_s": B / ms * 1e3, "ok": int((r.status == 0).sum()),
               "mean_len": float(r.out_len.float().mean())}
        if check:
            out.update(check_beam(x, r, 24, 32, 0.1))
        print(json.dumps(out), flush=True)
        del x
    if 4 in which:  # CRF, S=4 states x 5 symbols, one-hot init
        B = 4096
        x = rows((B, 4000, 4, 5), 3)
        init = torch.zeros((B, 4), device="cuda")
        init[torch.arange(B), torch.arange(B) % 4] = 1.0
        r, ms = timed(lambda: fcd.crf_beam_search_batch_raw(x, init, 5, 0.0), reps=2)
        out = {"config": 4, "workload": "crf_beam_search beam 5 thr 0.0, %d reads T=4000 S=4 N=5" % B,
               "kernel_ms": ms, "reads_per_s": B / ms * 1e3, "ok": int((r.status == 0).sum()),
               "mean_len": float(r.out_len.float().mean())}
        if check:
            out.update(check_crf(x, init, r, 64, 5, 0.0))
        print(json.dumps(out), flush=True)
        del x
    if 5 in which:  # duplex, 1024 pairs, band +-64
        B, T, w = 1024, 2000, 64
        x1, x2 = rows((B, T, 5), 4), rows((B, T, 5), 5)
        i = np.arange(T)
        env = np.stack([np.maximum(0, i - w), np.minimum(T, i + w)], 1).astype(np.uint64)
        envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
        for mode, name in ((0, "logsumexp"), (1, "max")):
            r, ms = timed(lambda: fcd.beam_search_duplex_batch_raw(x1, x2, envs, 5, 0.1, True, logadd_mode=mode), reps=1)
            out = {"config": 5, "workload": "beam_search_duplex %s beam 5 thr 0.1, %d pairs T1=T2=%d band +-%d"
                   % (name, B, T, w), "kernel_ms": ms, "pairs_per_s": B / ms * 1e3,
                   "ok": int((r.status == 0).sum()), "mean_len": float(r.out_len.float().mean())}
            if check:
                out.update(check_duplex(x1, x2, env, r, 6, mode))
            print(json.dumps(out), flush=True)
        # the alignment-band estimator on the same pairs (not a reference function)
        e, ms = timed(lambda: _est(x1, x2), reps=2)
        print(json.dumps({"config": "5 (envelope estimator)", "workload": "estimate_envelope_batch band 64 "
                          "(two greedy decodes + alignment + band), %d pairs T1=T2=%d" % (B, T),
                          "mean_ms_of_its_three_kernel_launches": ms}), flush=True)


if __name__ == "__main__":
    main()
