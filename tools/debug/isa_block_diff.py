"""Do the large basic blocks of one kernel survive a source change instruction for instruction?

    python tools/debug/isa_block_diff.py OLD.s NEW.s --kernel _ZN3irs20exp_bwd_march_kernelILb0ELi1E [--min 40]

Splits the kernel of both ISA listings (hipcc -S --cuda-device-only) into basic blocks, reduces every instruction to its
opcode (registers and literals dropped), and reports which blocks of at least `--min` instructions of OLD appear unchanged in
NEW (same opcode sequence), which changed (closest match and the number of differing opcodes) and what NEW has on top.
A kill criterion of the form "the main body must not change by an instruction" is checked with this."""
import argparse
import difflib
import re


def blocks(path, kernel):
    on, cur, out = False, [], []
    for ln in open(path):
        if not on:
            if ln.startswith(kernel) and re.match(r'^\S+:\s*(;.*)?$', ln):
                on = True
            continue
        t = ln.strip()
        if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'):
            continue
        if re.match(r'^\.LBB\d+_\d+:', t):
            if cur:
                out.append(cur)
            cur = []
            continue
        op = t.split()[0]
        cur.append(op)
        if op.startswith(('s_cbranch', 's_branch', 's_endpgm')):
            out.append(cur)
            cur = []
            if op == 's_endpgm':
                break
    if cur:
        out.append(cur)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('old')
    ap.add_argument('new')
    ap.add_argument('--kernel', required=True)
    ap.add_argument('--min', type=int, default=40)
    a = ap.parse_args()
    bo, bn = blocks(a.old, a.kernel), blocks(a.new, a.kernel)
    big_o = [b for b in bo if len(b) >= a.min]
    big_n = [b for b in bn if len(b) >= a.min]
    print(f'old: {len(bo)} blocks ({sum(map(len, bo))} instructions), {len(big_o)} of >= {a.min}; new: {len(bn)} blocks '
          f'({sum(map(len, bn))} instructions), {len(big_n)} of >= {a.min}')
    used = set()
    same = 0
    for i, b in enumerate(big_o):
        hit = next((j for j, c in enumerate(big_n) if j not in used and c == b), None)
        if hit is not None:
            used.add(hit)
            same += 1
            print(f'  old block {i:2d} ({len(b):4d} instr): unchanged')
            continue
        best, bj = 0.0, None
        for j, c in enumerate(big_n):
            if j in used:
                continue
            r = difflib.SequenceMatcher(None, b, c, autojunk=False).ratio()
            if r > best:
                best, bj = r, j
        if bj is not None:
            sm = difflib.SequenceMatcher(None, b, big_n[bj], autojunk=False)
            diff = sum(max(i2 - i1, j2 - j1) for tag, i1, i2, j1, j2 in sm.get_opcodes() if tag != 'equal')
            print(f'  old block {i:2d} ({len(b):4d} instr): CHANGED, closest new block has {len(big_n[bj])} instr, {diff} opcodes differ (ratio {best:.3f})')
            used.add(bj)
        else:
            print(f'  old block {i:2d} ({len(b):4d} instr): GONE')
    extra = [len(c) for j, c in enumerate(big_n) if j not in used]
    print(f'{same} of {len(big_o)} large blocks unchanged; new-only large blocks: {extra}')


if __name__ == '__main__':
    main()
