# round 5, batch 32: the SSD workload at 256^3 (BASELINE config 4's loss) kernel by kernel
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/profssd
rocprofv3 --kernel-trace --stats -d gpurun_out/profssd -o s --output-format csv -- python3 bench.py --loss ssd --no-cpu-baseline --no-extras --steps 10 --warmup 2 > gpurun_out/r05_ssd_prof.log 2>&1; rc=$?
grep -l "Memory access fault" gpurun_out/r05_ssd_prof.log && exit 9
f=$(find gpurun_out/profssd -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r05_ssd_256_kernel_stats.csv && head -24 gpurun_out/r05_ssd_256_kernel_stats.csv | cut -c1-70,180-330
rm -rf gpurun_out/profssd
python bench.py --loss ssd --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['stage_ms'])"
exit $rc
