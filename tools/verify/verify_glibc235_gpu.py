"""csrc/glibc235_math.h as compiled for gfx950, against the host's libm on EVERY binary32 argument (GPU + about four
minutes of host time; the host side of the same check is tools/verify/verify_glibc235.c).

    python tools/verify/verify_glibc235_gpu.py        -> one JSON line (differences must be 0 on a glibc 2.35 host)"""
import json
import os
import platform
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from fast_ctc_decode_amd import _native as nat
from oracle import oracle


def main():
    h = nat.default_handle()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    chunk = 1 << 26
    bad = {"expf": 0, "logf": 0, "log1pf": 0}
    yd = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for start in range(0, 1 << 32, chunk):
        x = np.arange(start, start + chunk, dtype=np.uint64).astype(np.uint32).view(np.float32)
        xd = torch.from_numpy(x).cuda()
        for which, name in enumerate(("expf", "logf", "log1pf")):
            h.check(h.lib.fcd_debug_glibc235_dev(h.ptr, which, xd.data_ptr(), yd.data_ptr(), chunk))
            torch.cuda.synchronize()
            got = yd.cpu().numpy()
            want = np.empty_like(x)
            oracle.lib.fcdo_libm_apply(which, x.ctypes.data, want.ctypes.data, x.size)
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            bad[name] += int((~same).sum())
    print(json.dumps({"host_libm": " ".join(platform.libc_ver()), "device": torch.cuda.get_device_name(0),
                      "arguments_per_function": 1 << 32, "differences": bad}))


if __name__ == "__main__":
    main()
