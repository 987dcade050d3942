"""sha256 of the adjoint of the 12 squaring steps on a fixed input: run once per library build (IRS_LIB) and compare -- two builds
that differ only in instruction selection / scheduling must print the same digest."""
import hashlib
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ir_sgmcmc_amd import ops as G

dev = torch.device('cuda', 0)
for N, amp in ((64, 0.8), (96, 0.3)):
    g = torch.Generator(device='cpu').manual_seed(5)
    lo = torch.randn(1, 3, N // 8, N // 8, N // 8, generator=g)
    v = torch.nn.functional.interpolate(lo, size=(N, N, N), mode='trilinear', align_corners=True)
    v = (v * (amp / float(v.abs().max()))).to(dev).contiguous()
    up = torch.randn(1, 3, N, N, N, generator=g).to(dev)
    _, _, steps = G.svf_exp_fwd(v, 12, want_outputs=False)
    gv = G.svf_exp_bwd(v, steps, up)
    print(N, amp, hashlib.sha256(gv.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(steps[-1].cpu().numpy().tobytes()).hexdigest()[:16])
