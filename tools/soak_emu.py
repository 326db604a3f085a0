"""Randomised differential run of the register beam kernels under the lockstep emulation (tests/hipemu), CPU only:
    python tools/soak_emu.py [first_seed] [seconds]
The emulated build poisons device memory and traps if a child row that was never stored is read back (the
dead-row test of beam_wave.hip / beam_lane.hip)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from emu_util import emulated_kernels
import fast_ctc_decode_amd as fcd
import test_gpu_parity as tp
t0=time.time(); n=0; bad=0
with emulated_kernels():
    seed=int(sys.argv[1]) if len(sys.argv)>1 else 700000
    while time.time()-t0 < float(sys.argv[2]) if len(sys.argv)>2 else 300:
        rng=np.random.default_rng(seed)
        N=int(rng.integers(3,6)); T=int(rng.integers(150,700)); beam=int(rng.choice([2,3,5,5,8,12,16,32]))
        x=tp.gen_batch(seed,int(rng.integers(1,3)),T,N,peaky=bool(rng.integers(0,2)))
        thr=float(rng.choice([0.0,0.01,0.1]))
        lengths=None if rng.integers(0,2) else rng.integers(1,T+1,size=x.shape[0]).astype(np.int64)
        for kernel in (2,3,4):
            try:
                tp.check_beam(fcd,x,beam,thr,True,lengths=lengths,kernel=kernel)
            except RuntimeError as e:
                if " kernel: " not in str(e): bad+=1; print("ERR",seed,kernel,e,flush=True)
            except AssertionError as e:
                bad+=1; print("MISMATCH",seed,kernel,str(e)[:200],flush=True)
        n+=1; seed+=1
print("emu soak: %d cases x 3 kernels, %d failures, %.0f s"%(n,bad,time.time()-t0))
