"""A non-Python consumer of the C ABI: tests/capi/consumer.c is compiled as C99 against
include/fcd.h, linked with the library and run on the golden vectors dumped as raw files -- `fcd_beam_search_host`,
the `fcd_*_host_begin / fcd_job_next / fcd_job_end` stream and `fcd_viterbi_search_host`, compared byte for byte in C.
CPU suite: linked against the emulator build of the same sources (tests/hipemu); -m gpu: against libfcd_hip.so."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = np.load(os.path.join(HERE, "golden", "vectors.npz"))


def dump_cases(d):
    lines = []
    for name in sorted({k.split("/")[0] for k in G.files if k.startswith("beam_")}):
        x = G[name + "/x"]
        if x.shape[0] > 8000:
            continue
        beam, thr, collapse = G[name + "/args"]
        x.astype("<f4").tofile(os.path.join(d, name + ".x.f32"))
        G[name + "/labels"].astype(np.uint8).tofile(os.path.join(d, name + ".labels.u8"))
        G[name + "/path"].astype("<u4").tofile(os.path.join(d, name + ".path.u32"))
        lines.append("%s %d %d %d %r %d %d %d" % (name, x.shape[0], x.shape[1], int(beam), float(thr), int(collapse),
                                                 int(G[name + "/status"][0]), len(G[name + "/labels"])))
    x = G["viterbi/x"]
    x.astype("<f4").tofile(os.path.join(d, "viterbi.x.f32"))
    G["viterbi/labels"].astype(np.uint8).tofile(os.path.join(d, "viterbi.labels.u8"))
    G["viterbi/path"].astype("<u4").tofile(os.path.join(d, "viterbi.path.u32"))
    lines.append("viterbi %d %d 0 0.0 1 0 %d" % (x.shape[0], x.shape[1], len(G["viterbi/labels"])))
    with open(os.path.join(d, "cases.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(lines)


def build_and_run(tmp_path, libdir, libname, extra_link=(), env=None):
    d = str(tmp_path)
    n = dump_cases(d)
    exe = os.path.join(d, "consumer")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1",
                           "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "capi", "consumer.c"), "-o", exe,
                           "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir] + list(extra_link))
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert ("consumer: %d cases ok" % n) in r.stdout


def test_c_consumer_on_the_emulated_library(tmp_path):
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build as emu_build
    lib = emu_build.build()
    build_and_run(tmp_path, os.path.dirname(lib), "fcd_emu")


@pytest.mark.gpu
def test_c_consumer_on_the_gpu_library(tmp_path):
    from fast_ctc_decode_amd import _native as nat
    from fast_ctc_decode_amd import build as hip_build
    hip_build.build()  # (a no-op when the library is there)
    # the consumer links libfcd_hip.so alone; the HIP runtime it needs at run time is the system's (/opt/rocm/lib)
    rocm = "/opt/rocm/lib"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = rocm + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    build_and_run(tmp_path, os.path.dirname(nat.LIB_PATH), "fcd_hip", extra_link=["-Wl,-rpath-link," + rocm], env=env)
