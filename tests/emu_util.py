"""TEST INFRASTRUCTURE: run the product's host layer (fast_ctc_decode_amd.api, numpy inputs through the
C ABI's *_host entry points) on tests/hipemu's lockstep emulation of the HIP kernels, so that kernel
logic can be compared with the oracle on a machine without a GPU.  The product itself only ever loads
libfcd_hip.so; this module swaps the loaded library object inside the test process and puts it back."""
import contextlib
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))


def emu_lib_path():
    import build as emu_build  # tests/hipemu/build.py
    return emu_build.build()


@contextlib.contextmanager
def emulated_kernels():
    from fast_ctc_decode_amd import _native as nat

    lib = nat.bind(C.CDLL(emu_lib_path()))
    saved_lib, saved_tls = nat._lib, getattr(nat._tls, "handles", None)
    nat._lib = lib
    nat._tls.handles = {}
    try:
        yield lib
    finally:
        for h in nat._tls.handles.values():
            h.close()
        nat._lib = saved_lib
        nat._tls.handles = saved_tls
