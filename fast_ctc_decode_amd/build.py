"""Builds libfcd_hip.so (the C-ABI library, include/fcd.h) in-tree with hipcc for gfx950.

    python -m fast_ctc_decode_amd.build [--force]

hipcc cross-compiles without a GPU.  Flags that matter for parity with the reference:
  -ffp-contract=off                       no FMA contraction: the reference rounds mul and add separately
  -fhip-fp32-correctly-rounded-divide-sqrt  IEEE f32 division for the per-step renormalisation
  (no -fgpu-flush-denormals-to-zero)      f32 subnormals are kept, as on the CPU
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfcd_hip.so")
# every translation unit and every header / include file under csrc/ (listed by glob, so that a new file can never be
# missing from the staleness test of needs_build(): a forgotten header once meant a stale libfcd_hip.so loaded silently)
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [
    os.path.join("..", "..", "include", f) for f in sorted(os.listdir(os.path.join(HERE, "..", "include"))) if f.endswith(".h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function",
]
# Per-file additions.  The duplex kernels run ONE wavefront per SIMD, every instruction on the critical path: where the
# window-building loop happens to land relative to the instruction-fetch lines moved the logsumexp search by 4 % between
# two builds of the same source (r06: 109.1 vs 104.5 ms on one box, profiles/r06i_duplex_alignment.txt); aligned loop heads
# take the lottery out.
EXTRA_FLAGS = {"duplex_slots.hip": ["-falign-loops=64"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


PYMOD_SRC = os.path.join(CSRC, "pymodule.cpp")


def pymodule_path():
    import sysconfig
    return os.path.join(HERE, "fast_ctc_decode" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pymodule(force=False, verbose=False):
    """The compiled Python module `fast_ctc_decode` (csrc/pymodule.cpp, pybind11): the C++ host side
    above the C ABI, mirroring the reference's PyO3 layer.  Links libfcd_hip.so via $ORIGIN."""
    import sysconfig

    import pybind11

    out = pymodule_path()
    build(force=False, verbose=verbose)
    deps = [PYMOD_SRC, os.path.join(HERE, "..", "include", "fcd.h"), LIB]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
           "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], PYMOD_SRC, "-o", out,
           "-L", HERE, "-lfcd_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_pymodule(force="--force" in sys.argv, verbose=True))
