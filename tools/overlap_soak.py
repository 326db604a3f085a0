"""Differential soak of fcd_set_overlap on the GPU: a random sequence of device-tensor calls -- every beam kernel, wide
beams through the slab pool (sometimes under a workspace limit that leaves a handful of slabs), duplex searches, viterbi
in between, changes of the number of internal streams, results sometimes read at once and sometimes much later -- each
compared with the same call made in stream order beforehand.  python tools/overlap_soak.py [CALLS] [SEED]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fast_ctc_decode_amd as fcd
from fast_ctc_decode_amd import _native as nat

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
h = nat.default_handle()


def rows(*shape):
    x = rng.random(shape, dtype=np.float32) + 0.05
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)


def make_case():
    kind = rng.choice(["wave", "wave1", "generic", "lane", "lane_pool", "duplex", "duplex_max"])
    B, T = int(rng.integers(1, 40)), int(rng.integers(20, 260))
    if kind in ("duplex", "duplex_max"):
        B, T = int(rng.integers(1, 12)), int(rng.integers(30, 140))
        T2 = T + int(rng.integers(-8, 9))
        w = int(rng.integers(6, 20))
        i = np.arange(T)
        env = np.stack([np.maximum(0, i - w), np.minimum(T2, i + w)], 1).astype(np.uint64)
        envs = torch.from_numpy(np.broadcast_to(env, (B, T, 2)).copy().view(np.int64)).cuda()
        a, b = torch.from_numpy(rows(B, T, 5)).cuda(), torch.from_numpy(rows(B, T2, 5)).cuda()
        mode = 1 if kind == "duplex_max" else 0
        return kind, lambda: fcd.beam_search_duplex_batch_raw(a, b, envs, 5, 0.1, True, logadd_mode=mode), None
    x = torch.from_numpy(rows(B, T, 5)).cuda()
    beam, kernel, limit = {"wave": (5, 2, None), "wave1": (8, 3, None), "generic": (int(rng.integers(2, 40)), 1, None),
                           "lane": (int(rng.integers(13, 65)), 4, None),
                           "lane_pool": (int(rng.integers(13, 65)), 4, int(rng.integers(2, 9)) << 20)}[kind]
    return kind, lambda: fcd.beam_search_batch_raw(x, beam, 0.05, True, kernel=kernel), limit


def same(a, b):
    if not (np.array_equal(a.status, b.status) and np.array_equal(a.out_len, b.out_len)):
        return False
    for i in range(len(a.out_len)):
        n = int(a.out_len[i])
        if not np.array_equal(a.labels[i, :n], b.labels[i, :n]):
            return False
        if a.path is not None and not np.array_equal(a.path[i, :n], b.path[i, :n]):
            return False
    return True


def limited(call, limit):
    if limit:
        h.set_workspace_limit(limit)
        h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 6))
    try:
        return call()
    finally:
        if limit:  # (the limit only matters while the call is being enqueued)
            h.set_workspace_limit(0)
            h.check(h.lib.fcd_debug_set_first_pass_divisor(h.ptr, 0))


bad, done, kinds = 0, 0, {}
while done < CALLS:
    # a chunk of cases: first in stream order (the reference), then all of them overlapping, read back in random order
    cases = [make_case() for _ in range(int(rng.integers(4, 25)))]
    h.set_overlap(0)
    wants = [limited(call, limit).cpu() for _, call, limit in cases]
    h.set_overlap(int(rng.integers(2, 9)))
    pending = []
    for (kind, call, limit), want in zip(cases, wants):
        kinds[kind] = kinds.get(kind, 0) + 1
        pending.append((kind, limited(call, limit), want))
        r = rng.random()
        if r < 0.10:  # another entry point in between: it takes the workspace from its start
            fcd.viterbi_search_batch_raw(torch.from_numpy(rows(int(rng.integers(1, 20)), 100, 5)).cuda(), True)
        if r > 0.85:
            k = int(rng.integers(0, len(pending)))
            for kind_, g, w in pending[k:]:
                done += 1
                if not same(g.cpu(), w):
                    bad += 1
                    print("MISMATCH", kind_, flush=True)
            del pending[k:]
    for kind_, g, w in pending:
        done += 1
        if not same(g.cpu(), w):
            bad += 1
            print("MISMATCH", kind_, flush=True)
h.set_overlap(0)
print("overlap soak: %d calls compared with their stream-order results, %d mismatches; kinds %s" % (done, bad, {str(k): v for k, v in kinds.items()}))
sys.exit(1 if bad else 0)
