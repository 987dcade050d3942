"""Where a wave of the radius-1 adjoint squaring step spends a plane step: wall-clock stamps (100 MHz) at the phase boundaries of
the marching loop, from a trace build of the library:

    bash tools/build_variant.sh bwdtrace -DIRS_BWD_TRACE
    IRS_LIB=$PWD/gpurun_variants/bwdtrace.so python tools/bwd_phase_trace.py --size 256

phases: 0 loop top -> 1 loads of the plane arrived -> 2 committed to LDS -> 3 next plane's loads issued -> 4 barrier passed ->
5 gather done -> 6 own term + stores issued -> 7 second barrier passed -> (next) 0
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=256)
    args = ap.parse_args()
    import torch
    from bench import engine_config
    from ir_sgmcmc_amd import _lib as L
    from ir_sgmcmc_amd.data_loader import synthetic_pair
    from ir_sgmcmc_amd.engine import TransitionEngine
    dev = torch.device('cuda', 0)
    lib = L.load()
    N = args.size
    f1, m1 = synthetic_pair((N, N, N), seed=0)
    eng = TransitionEngine(engine_config(N, 'gmm', 1), dev)
    fd, md = eng.prepare({k: v.unsqueeze(0).to(dev) for k, v in f1.items() if k != 'seg'},
                         {k: v.unsqueeze(0).to(dev) for k, v in m1.items() if k != 'seg'})
    eng.gmm_init(fd, md)
    v = torch.zeros(1, 3, N, N, N, device=dev)
    for _ in range(5):
        eng.transition(fd, md, v)
    eng.flush()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 64))()
    fn = lib.irs_debug_bwd_trace
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    t = [[buf[i * 8 + j] for j in range(8)] for i in range(64)]
    t = [r for r in t if r[0]]
    names = ['wait for loads', 'commit to LDS', 'issue next loads', 'barrier 1', 'gather', 'own term + store', 'barrier 2']
    rows = []
    for i in range(len(t) - 1):
        if t[i + 1][0] < t[i][0]:
            break
        rows.append([t[i][j + 1] - t[i][j] for j in range(7)] + [t[i + 1][0] - t[i][7]])
    print(f'{len(rows)} plane steps traced at {N}^3 (ticks of 10 ns)')
    for j, nme in enumerate(names + ['loop back']):
        col = [r[j] for r in rows]
        print(f'  {nme:18s} mean {sum(col) / len(col):8.1f}   min {min(col):6d}   max {max(col):6d}')
    for i, r in enumerate(rows):   # every plane step: the first ones of a workgroup run cold
        print(f'    step {i:2d}: ' + ' '.join(f'{x:6d}' for x in r) + f'   = {sum(r):6d}')
    tot = [sum(r) for r in rows]
    print(f'  per plane step     mean {sum(tot) / len(tot):8.1f}')


if __name__ == '__main__':
    main()
