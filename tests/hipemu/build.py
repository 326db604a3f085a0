"""TEST INFRASTRUCTURE ONLY: builds tests/hipemu/_build/libfcd_emu.so -- the product's csrc/*.hip compiled
unchanged for the host against the lockstep wave64 emulator in this directory (hip/hip_runtime.h,
hipemu.cpp), exporting the same C ABI (include/fcd.h).  The product never loads it.

    python tests/hipemu/build.py [--force]

The only source rewrite: `extern __shared__ ... T name[];` (dynamic LDS) becomes a pointer to the
emulator's per-block buffer.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "fast_ctc_decode_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libfcd_emu.so")

_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?((?:unsigned\s+)?\w+)\s+(\w+)\[\];")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    d += [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
          os.path.join(ROOT, "include", "fcd.h"), os.path.abspath(__file__)]
    return d


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, opt="-O1"):
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    flags = ["g++", opt, "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w",
             "-DFCD_HIPEMU=1", "-I", HERE, "-I", CSRC]
    procs, objs = [], []
    for s in _sources():
        text = open(os.path.join(CSRC, s)).read()
        text = _DYN.sub(lambda m: "%s *%s = reinterpret_cast<%s *>(hipemu::dyn_lds());" % (m.group(1), m.group(2), m.group(1)), text)
        cpp = os.path.join(OUT, s.replace(".hip", ".emu.cpp"))
        with open(cpp, "w") as f:
            f.write('#line 1 "%s"\n' % os.path.join(CSRC, s))
            f.write(text)
        o = cpp.replace(".cpp", ".o")
        procs.append((s, subprocess.Popen(flags + ["-c", cpp, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    o = os.path.join(OUT, "hipemu.o")
    procs.append(("hipemu.cpp", subprocess.Popen(flags + ["-c", os.path.join(HERE, "hipemu.cpp"), "-o", o],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("g++ failed on %s:\n%s" % (s, out.decode()))
    subprocess.check_call(["g++", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-lm"])
    return LIB


def pymodule_path(portable=False):
    import sysconfig
    d = os.path.join(OUT, "portable") if portable else OUT
    return os.path.join(d, "fast_ctc_decode" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pymodule(force=False, portable=False):
    """The product's csrc/pymodule.cpp (the compiled host layer) linked against libfcd_emu.so instead of
    libfcd_hip.so, so that its batch functions can be exercised without a GPU.  Test infrastructure only.
    portable: with -DFCD_PORTABLE_LISTS=1 -- the list[int] path of interpreters without a plain reference count."""
    import sysconfig

    import pybind11

    lib = build(force=False)
    out = pymodule_path(portable)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(CSRC, "pymodule.cpp")
    deps = [src, os.path.join(ROOT, "include", "fcd.h"), lib]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
                           "-ffp-contract=off", "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
                           src, "-o", out, "-L", OUT, "-lfcd_emu", "-Wl,-rpath," + OUT] +
                          (["-DFCD_PORTABLE_LISTS=1"] if portable else []))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_pymodule(force="--force" in sys.argv))
