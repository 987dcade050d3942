#!/bin/bash
# Round profile of the default bench (256^3 GMM/LCC, C = 1): kernel trace + stats, then two PMC passes (FETCH_SIZE,
# WRITE_SIZE) collected on their own with kernel-trace only, as the MI355X guide prescribes.  Run ON THE GPU BOX from the
# repo root:  bash tools/profile_round.sh <tag>     -> gpurun_out/profile_<tag>/ (copy the summaries into profiles/)
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/profile_$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extras --steps 10 --warmup 2 > "$OUT/trace.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extras --steps 3 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o w -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extras --steps 3 --warmup 1 > "$OUT/pmc_write.log" 2>&1
echo "WRITE_SIZE pass done"
cd "$REPO"
python3 tools/pmc_aggregate.py "$OUT" "$TAG"
