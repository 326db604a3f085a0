"""Executable specification of the duplex alignment-band estimator (fast_ctc_decode_amd.estimate_envelope*).

This is NOT part of the reference (its docstring only anticipates a non-trivial default envelope,
/root/reference/src/lib.rs:376-378), so there is no reference parity to claim:
the numpy code below defines the algorithm, and the HIP kernel (csrc/envelope.hip) must reproduce it
exactly -- it is integer work.

Algorithm, per pair:
  1. global alignment of the two label sequences under unit-cost edit distance; traceback from the
     end prefers diagonal, then up (consume a label of read 1), then left;
  2. every diagonal step over EQUAL labels is a match (a, b): label a of read 1 was emitted at time
     path1[a], label b of read 2 at path2[b] -> anchor (path1[a], path2[b]); anchors at read-1 time 0
     are dropped, (0, 0) and (T1, T2) are added;
  3. row i of the envelope is centred on the anchors' piecewise-linear interpolation (integer floor):
     lo = max(0, c - band), hi = min(T2, c + band + 1); then lo(0) = 0, hi(T1 - 1) = T2 and
     lo(i) = min(lo(i), hi(i - 1)) so that consecutive rows touch (src/duplex.rs:485-488)."""
import numpy as np


def align_matches(s1, s2):
    """-> list of (a, b), increasing, with s1[a] == s2[b] on the optimal path."""
    L1, L2 = len(s1), len(s2)
    D = np.zeros((L1 + 1, L2 + 1), np.int64)
    D[:, 0] = np.arange(L1 + 1)
    D[0, :] = np.arange(L2 + 1)
    s2a = np.asarray(s2)
    for i in range(1, L1 + 1):
        sub = D[i - 1, :-1] + (s2a != s1[i - 1])
        up = D[i - 1, 1:] + 1
        E = np.minimum(sub, up)
        # D[i][j] = min over j' <= j of (E[j'] + j - j'), E[0] := D[i][0] = i
        F = np.concatenate([[i], E]) - np.arange(L2 + 1)
        D[i, :] = np.minimum.accumulate(F) + np.arange(L2 + 1)
    out = []
    i, j = L1, L2
    while i > 0 or j > 0:
        if i > 0 and j > 0 and D[i, j] == D[i - 1, j - 1] + (s1[i - 1] != s2[j - 1]):
            if s1[i - 1] == s2[j - 1]:
                out.append((i - 1, j - 1))
            i, j = i - 1, j - 1
        elif i > 0 and D[i, j] == D[i - 1, j] + 1:
            i -= 1
        else:
            j -= 1
    return out[::-1]


def envelope(labels1, path1, T1, labels2, path2, T2, band):
    """-> (T1, 2) uint64 array of [lo, hi) column ranges of read 2 for every row of read 1."""
    env = np.zeros((T1, 2), np.uint64)
    if T1 == 0:
        return env
    anchors = [(0, 0)]
    for a, b in align_matches(list(labels1), list(labels2)):
        if int(path1[a]) > 0:
            anchors.append((int(path1[a]), int(path2[b])))
    anchors.append((T1, T2))
    lo = np.zeros(T1, np.int64)
    hi = np.zeros(T1, np.int64)
    k = 0
    for i in range(T1):
        while anchors[k + 1][0] <= i:
            k += 1
        (ta, ua), (tb, ub) = anchors[k], anchors[k + 1]
        c = ua + ((i - ta) * (ub - ua)) // (tb - ta)
        lo[i] = max(0, c - band)
        hi[i] = min(T2, c + band + 1)
    lo[0] = 0
    hi[T1 - 1] = T2
    lo[1:] = np.minimum(lo[1:], hi[:-1])
    env[:, 0], env[:, 1] = lo, hi
    return env
