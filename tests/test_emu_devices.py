"""Device index > 0 ("what's missing"): a one-GPU box can never run `fcd_create(1)`, the lanes of a host job
on device 1, a coalescer on device 1, or check that every entry point puts the CALLER's current device back.  The
emulator can: with FCD_EMU_DEVICES=2 the current device is per host thread, streams and allocations remember their
device, and touching them under another current device aborts the process (tests/hipemu/hipemu.cpp).  Runs in a
subprocess: the device count is read when the library loads and an abort must not take the test session down."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, os, sys, threading
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from emu_util import emulated_kernels
from fast_ctc_decode_amd import _native as nat
from oracle import oracle
from kat_cases import reference_style_rows

def current(lib):
    return int(C.CDLL(lib._name).hipemu_current_device())

with emulated_kernels() as lib:
    assert lib.fcd_device_count() == 2
    rng = np.random.default_rng(3)
    x = reference_style_rows(rng, 6 * 90, 5).reshape(6, 90, 5)
    want = [oracle.beam_search_raw(x[i], 5, 0.1, True) for i in range(6)]
    h1 = nat.Handle(1)                      # fcd_create(1): stream and workspace on device 1
    assert current(lib) == 0                # ... and the caller's device put back
    B, T = x.shape[0], x.shape[1]
    labels = np.zeros((B, T), np.uint8); path = np.zeros((B, T), np.uint32)
    out_len = np.zeros(B, np.uint32); status = np.zeros(B, np.int32)
    b = nat.Batch(x.ctypes.data, B, T, 1, 5, T * 5, 5, 0, 1, None)
    r = nat.Result(labels.ctypes.data, path.ctypes.data, None, out_len.ctypes.data, status.ctypes.data, T, None)
    for beam in (5, 32):                    # wave kernel, lane kernel (two-pass arena sizing included)
        wantb = [oracle.beam_search_raw(x[i], beam, 0.1, True) for i in range(B)]
        h1.check(lib.fcd_beam_search_host(h1.ptr, C.byref(b), beam, 0.1, 1, nat.KERNEL_AUTO, C.byref(r)))
        assert current(lib) == 0
        for i in range(B):
            st, lab, pth, _ = wantb[i]
            n = int(out_len[i])
            assert int(status[i]) == st == 0 and np.array_equal(labels[i, :n], lab) and np.array_equal(path[i, :n], pth), (beam, i)
    # a host job: its lanes are sub-handles with streams of their own -- on device 1 too
    h1.check(lib.fcd_set_host_pipeline(h1.ptr, 3, 2, 0))
    labels[:] = 0; out_len[:] = 0
    h1.check(lib.fcd_beam_search_host(h1.ptr, C.byref(b), 5, 0.1, 1, nat.KERNEL_AUTO, C.byref(r)))
    assert current(lib) == 0
    for i in range(B):
        st, lab, pth, _ = want[i]
        n = int(out_len[i])
        assert int(status[i]) == 0 and np.array_equal(labels[i, :n], lab) and np.array_equal(path[i, :n], pth), i
    # viterbi and release on device 1, from ANOTHER thread (whose current device starts at 0)
    err = []
    def other():
        try:
            h1.check(lib.fcd_viterbi_search_host(h1.ptr, C.byref(b), 1, C.byref(r)))
            assert current(lib) == 0
            for i in range(B):
                lab, pth = oracle.viterbi_search_raw(x[i], True)[:2]
                n = int(out_len[i])
                assert np.array_equal(labels[i, :n], lab) and np.array_equal(path[i, :n], pth), i
            h1.check(lib.fcd_release_workspace(h1.ptr))
        except Exception as e:
            err.append(repr(e))
    t = threading.Thread(target=other); t.start(); t.join()
    assert not err, err
    # a coalescer on device 1: per-read calls from two threads
    co = nat.Coalescer(1, 8, 20000)
    got = {}
    def caller(i):
        lab1 = np.zeros((1, T), np.uint8); pth1 = np.zeros((1, T), np.uint32)
        ol = np.zeros(1, np.uint32); st1 = np.zeros(1, np.int32)
        xb = nat.Batch(x[i].ctypes.data, 1, T, 1, 5, T * 5, 5, 0, 1, None)
        rr = nat.Result(lab1.ctypes.data, pth1.ctypes.data, None, ol.ctypes.data, st1.ctypes.data, T, None)
        with co:
            co.check(lib.fcd_coalescer_beam_search(co.ptr, C.byref(xb), 5, 0.1, 1, C.byref(rr)))
        got[i] = (int(st1[0]), lab1[0, :int(ol[0])].copy(), pth1[0, :int(ol[0])].copy())
    ts = [threading.Thread(target=caller, args=(i,)) for i in range(B)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for i in range(B):
        st, lab, pth, _ = want[i]
        assert got[i][0] == st and np.array_equal(got[i][1], lab) and np.array_equal(got[i][2], pth), i
    co.close()
    assert current(lib) == 0
    h1.close()
    # device 2 does not exist
    p = C.c_void_p()
    assert lib.fcd_create(2, C.byref(p)) != 0
print("devices ok")
'''


def test_device_one_on_the_emulator():
    env = dict(os.environ, FCD_EMU_DEVICES="2")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0 and "devices ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


def test_the_emulator_catches_a_wrong_current_device():
    """the check itself: a launch on device 1's stream while device 0 is current must abort"""
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from emu_util import emu_lib_path
lib = C.CDLL(emu_lib_path())
# the C++ runtime stand-ins are not extern "C": reach them through a handle made on device 1 and a raw stream swap
from fast_ctc_decode_amd import _native as nat
from emu_util import emulated_kernels
import numpy as np
with emulated_kernels() as l:
    h0, h1 = nat.Handle(0), nat.Handle(1)
    # hand device 1's stream to the device-0 handle: its next enqueue is a wrong-device operation
    l.fcd_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    s1 = C.c_void_p.from_address(h1.ptr.value + 8).value  # fcd_handle {int device; hipStream_t own_stream; ...}
    l.fcd_set_stream(h0.ptr, s1)
    x = np.full((1, 8, 5), 0.2, np.float32)
    lab = np.zeros((1, 8), np.uint8); pth = np.zeros((1, 8), np.uint32); ol = np.zeros(1, np.uint32); st = np.zeros(1, np.int32)
    b = nat.Batch(x.ctypes.data, 1, 8, 1, 5, 40, 5, 0, 1, None)
    r = nat.Result(lab.ctypes.data, pth.ctypes.data, None, ol.ctypes.data, st.ctypes.data, 8, None)
    l.fcd_viterbi_search_dev(h0.ptr, C.byref(b), 1, C.byref(r))
print("not caught")
'''
    env = dict(os.environ, FCD_EMU_DEVICES="2")
    r = subprocess.run([sys.executable, "-c", code % {"root": ROOT}], capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=900)
    assert r.returncode != 0 and "while device 0 is current" in r.stderr and "not caught" not in r.stdout, (r.stdout, r.stderr[-2000:])
