"""The shape of every marching launch of one transition -- workgroups, threads, segment length, run-in planes, plane steps per
workgroup, workgroups the chip holds at once, rounds -- at 256^3, at 128^3 (one and two chains) and for one rank of eight of a 256^3 slab run (32 owned
planes of 256 x 256), as the launchers themselves report it under IRS_LAUNCH_LOG=1 (csrc/api.hip: log_launch).

    python tools/launch_shapes.py            # on the GPU box; writes gpurun_out/r05_launch_shapes.json and prints a table
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    ('256^3', [sys.executable, 'bench.py', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras']),
    ('128^3', [sys.executable, 'bench.py', '--size', '128', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extras']),
    ('128^3, two chains', [sys.executable, 'tools/two_chain_run.py', '--steps', '2']),
    ('rank of 8 of 256^3', [sys.executable, 'tools/slab_probe.py', '--size', '256', '--worlds', '8']),
]


def main():
    out = {}
    env = dict(os.environ, IRS_LAUNCH_LOG='1')
    for name, cmd in CASES:
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        rows = [json.loads(ln.split('[irs launch] ', 1)[1]) for ln in p.stderr.splitlines() if ln.startswith('[irs launch] ')]
        out[name] = rows
        print(f'== {name}  (rc {p.returncode})')
        print(f'{"kernel":58s} {"tile":>7s} {"wgs":>6s} {"thr":>4s} {"seg":>4s} {"run-in":>6s} {"steps":>5s} {"resident":>8s} {"rounds":>6s} {"overhead":>8s}')
        for r in rows:
            print(f'{r["kernel"]:58s} {r["tile"][0]:>3d}x{r["tile"][1]:<3d} {r["workgroups"]:>6d} {r["threads"]:>4d} {r["seg_len"]:>4d} {r["run_in"]:>6d} '
                  f'{r["plane_steps"]:>5d} {r["resident"]:>8d} {r["rounds"]:>6.2f} {r["run_in_overhead"]:>8.2f}')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'r05_launch_shapes.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
