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
import csv, glob, sys, json, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/**/k_kernel_trace.csv', recursive=True)[0]
ts = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
# the probe runs the fused engine first, then (after building the slab context: the longest pause between two transitions) the slab
# rank; every transition ends with finalize_kernel
fin = [i for i, t in enumerate(ts) if 'finalize_kernel' in t[2]]
gaps = [(ts[fin[j]][0] - ts[fin[j - 1]][0], j) for j in range(1, len(fin))]
first_slab = fin[max(gaps)[1] - 1] + 1
def last(fins, n):
    part = ts[fins[-n - 1] + 1:fins[-1] + 1]
    busy = sum(e - s for s, e, _ in part)
    return part, {'kernels_per_transition': len(part) / n, 'busy_ms': busy / n / 1e6, 'span_ms': (part[-1][1] - part[0][0]) / n / 1e6}
n = 20
_, fused = last([i for i in fin if i < first_slab], n)
part, slab = last([i for i in fin if i > first_slab], n)
per = collections.Counter()
for s, e, name in part:
    per[name.split('(')[0].replace('void irs::', '').replace('irs::', '')] += (e - s) / n / 1e3
print(json.dumps({'fused': fused, 'slab_rank': slab, 'slab_rank_us_by_kernel': {k: round(v, 1) for k, v in per.most_common(16)}}))
PY
tail -n 3 "$OUT/probe.log"
