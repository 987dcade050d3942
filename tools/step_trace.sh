#!/bin/bash
# Per-launch durations of the squaring-step kernels over the LAST transition of a bench run (which variant did each step's work
# and what did it cost).  Run ON THE GPU BOX from the repo root:  bash tools/step_trace.sh <tag> [bench args...]
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/step_trace_$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o k -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extras --steps 6 --warmup 3 "$@" > "$OUT/bench.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/trace/**/k_kernel_trace.csv', recursive=True)[0]
ts = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
fin = [i for i, t in enumerate(ts) if 'finalize_kernel' in t[2]]
part = ts[fin[-2] + 1:fin[-1] + 1]
print('last transition: %d launches, busy %.3f ms, span %.3f ms' % (len(part), sum(e - s for s, e, _ in part) / 1e6, (part[-1][1] - part[0][0]) / 1e6))
for s, e, name in part:
    short = name.split('(')[0].replace('void irs::', '').replace('irs::', '')
    if (e - s) > 8000 or 'exp_' in short:
        print('%9.1f us  %s' % ((e - s) / 1e3, short))
PY
tail -n 1 "$OUT/bench.log" | cut -c1-200
