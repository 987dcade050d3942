# round 5, final batch with the FMA forward step as default: full suite, kernel stats, SQ counters, the default bench line
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05_suite_final.txt 2>&1; rc=$?; tail -20 gpurun_out/r05_suite_final.txt
fault gpurun_out/r05_suite_final.txt
[ $rc -ne 0 ] && exit $rc
bash tools/profile_round.sh r05 2>&1 | tail -3
bash tools/pmc_sq.sh 2>&1 | tail -3
python tools/sq_aggregate.py r05 | tail -3
rm -rf gpurun_out/pmc_sq gpurun_out/profile_r05/trace gpurun_out/profile_r05/pmc_fetch gpurun_out/profile_r05/pmc_write
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; python -c "
import json;d=json.load(open('gpurun_out/r05_bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['exp_step_fwd']);print({k:(round(v.get('ms_per_transition',0),4),round(v.get('transitions_per_s',0),1)) for k,v in d['also'].items()})"
