"""Phases of chain_scalar_kernel (trace build: tools/build_variant.sh sctrace -DIRS_SCALAR_TRACE): wall-clock stamps (100 MHz) of
thread 0 -- 0 start, 1 verdict done, 2 statistics reduced, 3 VD factor done, 4 Adam step computed, 5 parameters written + derived
constants refreshed."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import engine_config
from ir_sgmcmc_amd import _lib as L
from ir_sgmcmc_amd.data_loader import synthetic_pair
from ir_sgmcmc_amd.engine import TransitionEngine

dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
f1, m1 = synthetic_pair((N, N, N), seed=0)
eng = TransitionEngine(engine_config(N, 'gmm', 1), dev)
fd, md = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'}, {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
eng.gmm_init(fd, md)
v = torch.zeros(1, 3, N, N, N, device=dev)
lib = L.load()
fn = lib.irs_debug_scalar_trace
fn.argtypes = [C.POINTER(C.c_ulonglong)]
for it in range(6):
    eng.transition(fd, md, v)
    eng.flush()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    assert fn(buf) == 0
    t = [buf[i] for i in range(6)]
    if buf[6]:
        print('   of the reduce phase: loads + adds %.2f us, block sum %.2f us' % ((buf[6] - t[1]) / 100, (t[2] - buf[6]) / 100))
    print('transition', it, ' '.join('%s %.2f us' % (n, (t[i + 1] - t[i]) / 100) for i, n in enumerate(['verdict', 'reduce', 'alpha', 'adam', 'write+refresh'])), ' total %.2f' % ((t[5] - t[0]) / 100))
