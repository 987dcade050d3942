"""Where a workgroup of the any-radius adjoint (exp_bwd_lds_kernel) spends a tile: wall-clock stamps (100 MHz) of one thread at
the phase boundaries, from a trace build of the library:

    bash tools/build_variant.sh ldstrace -DIRS_LDS_TRACE
    IRS_LIB=$PWD/gpurun_variants/ldstrace.so python tools/lds_phase_trace.py --size 256 --amp 6

phases: 0 tile start -> 1 source box known, accumulators zeroed -> 2 scatter done -> 3 own voxels written
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
    ap.add_argument('--amp', type=float, default=6.0)
    args = ap.parse_args()
    import torch
    from bench import engine_config, initial_velocity
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
    v = initial_velocity('wave', args.amp, N, dev)
    for _ in range(4):
        eng.transition(fd, md, v)
    eng.flush()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 32))()
    fn = lib.irs_debug_lds_trace
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    rows = [[buf[i * 8 + j] for j in range(5)] for i in range(32)]
    rows = [r for r in rows if r[0] and r[3] > r[0]]
    print(f'{len(rows)} tiles of one workgroup at {N}^3, amp {args.amp} (us)')
    for r in rows:
        box = (r[4] & 0xffff, (r[4] >> 16) & 0xffff, (r[4] >> 32) & 0xffff)
        print('  box+zero %6.2f   scatter %6.2f   own %6.2f   total %6.2f   source box %s' %
              ((r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[3] - r[0]) / 100, box))
    if len(rows) > 1:
        print('  first tile start -> last tile end: %.1f us for %d tiles' % ((rows[-1][3] - rows[0][0]) / 100, len(rows)))


if __name__ == '__main__':
    main()
