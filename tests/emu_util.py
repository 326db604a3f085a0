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


_emu_compiled = None


def emu_compiled_module():
    """csrc/pymodule.cpp linked against the emulator library (tests/hipemu/build.py), loaded under a private
    name so that it never shadows the product's `fast_ctc_decode`."""
    global _emu_compiled
    if _emu_compiled is None:
        import importlib.util

        import build as emu_build
        # (a dotted name: the init function is looked up by the last component, while pybind11 caches completed
        # modules by the FULL name -- under the bare name the product's module would be handed back)
        spec = importlib.util.spec_from_file_location("fcd_hipemu.fast_ctc_decode", emu_build.build_pymodule())
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _emu_compiled = mod
    return _emu_compiled


@contextlib.contextmanager
def emulated_kernels():
    from fast_ctc_decode_amd import _native as nat
    from fast_ctc_decode_amd import api

    lib = nat.bind(C.CDLL(emu_lib_path()))
    saved_lib, saved_tls, saved_cm = nat._lib, getattr(nat._tls, "handles", None), api._cm
    nat._lib = lib
    nat._tls.handles = {}
    api._cm = emu_compiled_module()
    try:
        yield lib
    finally:
        for h in nat._tls.handles.values():
            h.close()
        nat._lib = saved_lib
        nat._tls.handles = saved_tls
        api._cm = saved_cm
