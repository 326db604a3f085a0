"""Drop-in import name: `import fast_ctc_decode` resolves to the compiled MI355X module
(fast_ctc_decode_amd/csrc/pymodule.cpp over include/fcd.h).  This shim only orders the loading
(PyTorch's bundled HIP runtime first) and then replaces itself with
the extension module."""
import importlib.util
import os
import sys

from fast_ctc_decode_amd import _native, build as _build

_native.load()  # builds libfcd_hip.so if needed; imports torch first when it is installed
_path = _build.pymodule_path()
if not os.path.exists(_path):
    _build.build_pymodule()
_spec = importlib.util.spec_from_file_location("fast_ctc_decode", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
sys.modules[__name__] = _mod
