set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/trend.so
for f in gpurun_variants/trend.so gpurun_variants/notrend.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_trend2_chain_bits.txt 2>&1
fault gpurun_out/r05_trend2_chain_bits.txt; cat gpurun_out/r05_trend2_chain_bits.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/notrend.so gpurun_variants/trend.so 3 --size 128 --steps 300 > gpurun_out/r05_trend2_ab_128.txt 2>&1
fault gpurun_out/r05_trend2_ab_128.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/notrend.so gpurun_variants/trend.so 2 --steps 100 > gpurun_out/r05_trend2_ab_256.txt 2>&1
fault gpurun_out/r05_trend2_ab_256.txt
grep -h -v amdgpu.ids gpurun_out/r05_trend2_ab_128.txt gpurun_out/r05_trend2_ab_256.txt
rm -f gpurun_variants/trend.so
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05_suite_final.txt 2>&1; rc=$?; tail -22 gpurun_out/r05_suite_final.txt
fault gpurun_out/r05_suite_final.txt
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; python -c "
import json;d=json.load(open('gpurun_out/r05_bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac']);print({k:(round(v.get('ms_per_transition',0),4),round(v.get('transitions_per_s',0),1)) for k,v in d['also'].items()})"
