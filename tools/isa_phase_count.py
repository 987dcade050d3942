"""Static instruction mix of one kernel of the shipped library, from its ISA (hipcc -S of the same sources and flags).

    python tools/isa_phase_count.py [--kernel _ZN3irs20exp_bwd_march_kernelILb0ELi1E] [--src ir_sgmcmc_amd/csrc/exp_kernels.hip]

Prints, for the whole kernel and for every region between two `s_barrier`s, the number of VALU (v_*), LDS (ds_*), vector-memory
(global_* / buffer_* / scratch_*), scalar (s_*) instructions and waits.  For the z-marching adjoint the plane loop is unrolled
over the three slots of its LDS ring, so the loop body appears three times: per plane step = loop totals / 3.  The issue floor
of a kernel that is bound by vector-instruction issue is  VALU per wave-plane-step x 4 cycles x (wave-plane-steps per launch)
/ (1024 SIMDs x clock)  -- see DESIGN.md section 4.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith('v_'):
        return 'valu_quarter' if re.match(r'v_(mul_lo|mul_hi|rcp|rsq|sqrt|exp|log|sin|cos|cvt_.*f64|.*_f64)', op) else 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='_ZN3irs20exp_bwd_march_kernelILb0ELi1E')
    ap.add_argument('--src', default=os.path.join(ROOT, 'ir_sgmcmc_amd', 'csrc', 'exp_kernels.hip'))
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=on', '-fvisibility=hidden', '-S',
                        '--cuda-device-only', '-o', out, args.src], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    body, on = [], False
    for ln in lines:
        if ln.startswith(args.kernel) and re.match(r'^\S+:\s*(;.*)?$', ln):
            on = True
            continue
        if on:
            body.append(ln)
            if 's_endpgm' in ln:
                break
    if not body:
        sys.exit(f'kernel {args.kernel} not found')
    # the loop nest = from the first label that is the target of a backward branch to the last backward branch
    labels = {m.group(1): i for i, ln in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', ln))}
    back = [(i, labels[m.group(1)]) for i, ln in enumerate(body)
            if (m := re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', ln)) and m.group(1) in labels and labels[m.group(1)] < i]
    lo, hi = (min(t for _, t in back), max(i for i, _ in back)) if back else (0, len(body))
    total, regions, cur = collections.Counter(), [], collections.Counter()
    loop = collections.Counter()
    for i, ln in enumerate(body):
        t = ln.strip()
        if not t or t.startswith((';', '.')) or t.endswith(':'):
            continue
        kind = classify(t.split()[0])
        total[kind] += 1
        if lo <= i <= hi:
            loop[kind] += 1
            cur[kind] += 1
            if kind == 'barrier':
                regions.append(dict(cur))
                cur = collections.Counter()
    if cur:
        regions.append(dict(cur))
    res = {'kernel': args.kernel, 'isa_lines': len(body), 'whole_kernel': dict(total), 'loop_nest': dict(loop),
           'barriers_in_loop_nest': loop['barrier'], 'regions_between_barriers': regions}
    print(json.dumps(res, indent=1))
    if args.json:
        json.dump(res, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
