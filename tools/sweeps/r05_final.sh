# the last GPU batch of round 5: the committed library -- full default suite, then the default bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05_suite_final.txt 2>&1; rc=$?; tail -20 gpurun_out/r05_suite_final.txt
grep -l "Memory access fault" gpurun_out/r05_suite_final.txt && exit 9
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; python -c "
import json;d=json.load(open('gpurun_out/r05_bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['exp_step_fwd']);print({k:(round(v.get('ms_per_transition',0),4),round(v.get('transitions_per_s',0),1)) for k,v in d['also'].items()})"
