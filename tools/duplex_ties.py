"""How often could the duplex searches' result depend on the order Rust's unstable sort leaves equal
probabilities in (src/duplex.rs:620,807)?  CPU only: runs the oracle on config-5-shaped pairs and prints its
tie statistics.    python tools/duplex_ties.py [pairs] [T] [band]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import oracle
import test_gpu_duplex as td


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    x1, x2 = td.pairs(5, n, T, T)
    envs = np.stack([td.band(T, T, w)] * n)
    for name, mode in (("logsumexp (correctly rounded)", td.LSE | td.CR), ("max", td.MAX)):
        oracle.duplex_tie_steps(reset=True)
        td.oracle_strings(x1, x2, "NACGT", envs, 5, 0.1, True, mode)
        print("%d pairs T=%d band +-%d, %s:" % (n, T, w, name), oracle.duplex_tie_steps())


if __name__ == "__main__":
    main()
