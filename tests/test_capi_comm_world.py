"""The C-ABI multi-GPU step above world size 1: tests/capi/comm_world.c forks N processes, each with
the emulator build of the library and tests/stubs/librccl_stub.c (five RCCL entry points over POSIX shared memory) in
place of RCCL -- fcd_comm_create, a beam search per rank, ONE fcd_gather_results_dev, rank 0 compares with a
single-process decode.  Covers uneven and empty shards, out_stride across the 65535-row boundary, a shard whose header
contradicts the read counts, an allocation failure on one rank between the size agreement and the gather, and (r06) a
failure on one rank BEFORE the size agreement: every rank must come back with the error, nobody may hang.  No GPU: two ranks cannot share one under RCCL, and no scaling
curve is claimed from this -- it tests the PROTOCOL (comm.hip), the only part of the multi-GPU path with an exchange."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def world_exe(tmp_path_factory):
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build as emu_build
    lib = emu_build.build()
    d = str(tmp_path_factory.mktemp("comm_world"))
    stub = os.path.join(d, "librccl_stub.so")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-O1", "-shared", "-fPIC",
                           os.path.join(HERE, "stubs", "librccl_stub.c"), "-o", stub, "-lpthread", "-lrt"])
    exe = os.path.join(d, "comm_world")
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(HERE, "capi", "comm_world.c"), "-o", exe, "-L", libdir, "-lfcd_emu",
                           "-Wl,-rpath," + libdir])
    env = dict(os.environ)
    env["FCD_RCCL_LIBRARY"] = stub
    env["FCD_EMU_DEVICES"] = "8"  # one process per (emulated) GPU: rank k creates its handle and communicator on device k
    return exe, env


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("scenario", ["uneven", "wide", "mixed", "badheader", "allocfail", "prepfail"])
def test_gather_between_processes(world_exe, world, scenario):
    exe, env = world_exe
    r = subprocess.run([exe, str(world), scenario], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert ("scenario %s at world size %d ok" % (scenario, world)) in r.stdout
    if scenario in ("uneven", "wide", "mixed"):  # rank 0 really compared every gathered read with its own decode
        assert ("comm_world: %d ranks" % world) in r.stdout and "gathered == decoded in one process" in r.stdout


def test_world_one_needs_no_stub(world_exe):
    exe, env = world_exe
    r = subprocess.run([exe, "1", "uneven"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
