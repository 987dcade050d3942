"""Where a workgroup of the forward squaring step spends a plane step: time stamps (clock64) of one wave at six points of the
marching loop, from a trace build of the library:

    bash tools/build_variant.sh trace -DIRS_FWD_TRACE
    IRS_LIB=$PWD/gpurun_variants/trace.so python tools/fwd_phase_trace.py --size 128

phases: 0 loop top -> 1 loads of the plane arrived -> 2 committed to LDS -> 3 next plane's loads issued -> 4 barrier passed ->
5 outputs computed and stores issued -> (next) 0
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
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
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 64))()
    fn = lib.irs_debug_fwd_trace
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    t = [[buf[i * 8 + j] for j in range(6)] for i in range(63)]
    t = [r for r in t if r[0]]
    e = [buf[63 * 8 + j] for j in range(5)]   # entry 63: (clock64, 100 MHz wall clock) after the tile set-up and after the loop; plane steps
    # the last launch that traced overwrote the earlier ones: one workgroup's plane steps in order
    names = ['wait for loads', 'commit to LDS', 'issue next loads', 'barrier', 'compute + stores']
    rows = []
    for i in range(len(t) - 1):
        if t[i + 1][0] < t[i][0]:
            break
        rows.append([t[i][j + 1] - t[i][j] for j in range(5)] + [t[i + 1][0] - t[i][5]])
    print(f'{len(rows)} plane steps traced at {N}^3 (ticks of clock64)')
    for j, nme in enumerate(names + ['loop back']):
        col = [r[j] for r in rows]
        print(f'  {nme:18s} mean {sum(col) / len(col):8.1f}   min {min(col):6d}   max {max(col):6d}')
    if os.environ.get('IRS_TRACE_ROWS', '1') != '0':   # every plane step: the first ones of a workgroup run cold (first loads, first pass over each unrolled ring phase)
        for i, r in enumerate(rows):
            print(f'    step {i:2d}: ' + ' '.join(f'{x:6d}' for x in r) + f'   = {sum(r):6d}')
    tot = [sum(r) for r in rows]
    print(f'  per plane step     mean {sum(tot) / len(tot):8.1f}')
    if e[0] and e[2] > e[0] and e[3] > e[1]:
        ticks, wall_ns = e[2] - e[0], (e[3] - e[1]) * 10.0
        print(f'  whole marching loop of this workgroup: {int(e[4])} plane steps, {ticks} ticks = {wall_ns / 1e3:.2f} us '
              f'({ticks / wall_ns * 1e3:.0f} MHz tick rate; {wall_ns / max(int(e[4]), 1):.0f} ns per plane step)')
        if rows:
            print(f'  first plane step starts {t[0][0] - e[0]} ticks after the tile set-up (first prefetches issued in between)')


if __name__ == '__main__':
    main()
