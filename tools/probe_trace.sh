#!/bin/bash
# Kernel-time sum of one slab rank against its wall time: is a rank of N bound by its kernels or by launch / sync gaps?
# Run ON THE GPU BOX from the repo root:  bash tools/probe_trace.sh <world> [size]   -> gpurun_out/probe_trace_<world>/
set -e
W=${1:-8}
N=${2:-256}
OUT=$PWD/gpurun_out/probe_trace_$W
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 "$REPO/tools/slab_probe.py" --size "$N" --worlds "$W" --steps 30 > "$OUT/probe.log" 2>&1
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
f = glob.glob(out + '/trace/**/k_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the probe runs the fused engine first (5 + 30 transitions), then the slab rank (5 + 30): split at the largest gap
ts = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
gaps = sorted(((ts[i + 1][0] - ts[i][1], i) for i in range(len(ts) - 1)), reverse=True)
cut = gaps[0][1] + 1
for name, part in (('first_phase', ts[:cut]), ('second_phase', ts[cut:])):
    busy = sum(e - s for s, e, _ in part)
    span = part[-1][1] - part[0][0]
    print(json.dumps({'phase': name, 'kernels': len(part), 'busy_ms': busy / 1e6, 'span_ms': span / 1e6, 'busy_frac': busy / span}))
PY
tail -n 3 "$OUT/probe.log"
