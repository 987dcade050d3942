set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests/test_gpu_transition.py tests/test_gpu_recovery_fuzz.py tests/test_gpu_trainer.py -m gpu -x -q > gpurun_out/r05_t_trend.txt 2>&1; rc=$?; tail -3 gpurun_out/r05_t_trend.txt
fault gpurun_out/r05_t_trend.txt
[ $rc -ne 0 ] && exit $rc
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/trend.so
for f in gpurun_variants/trend.so gpurun_variants/notrend.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_trend_chain_bits.txt 2>&1
fault gpurun_out/r05_trend_chain_bits.txt; cat gpurun_out/r05_trend_chain_bits.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/notrend.so gpurun_variants/trend.so 3 --size 128 --steps 300 > gpurun_out/r05_trend_ab_128.txt 2>&1
fault gpurun_out/r05_trend_ab_128.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/notrend.so gpurun_variants/trend.so 3 --steps 100 > gpurun_out/r05_trend_ab_256.txt 2>&1
fault gpurun_out/r05_trend_ab_256.txt
grep -h -v amdgpu.ids gpurun_out/r05_trend_ab_128.txt gpurun_out/r05_trend_ab_256.txt
IRS_LIB=$PWD/gpurun_variants/trend.so bash tools/step_trace.sh trend128 --size 128 > gpurun_out/r05_trend_step_trace_128.txt 2>&1; rm -rf gpurun_out/step_trace_trend128/trace
grep -v "amdgpu.ids\|rocprofv3" gpurun_out/r05_trend_step_trace_128.txt | tail -45
rm -f gpurun_variants/trend.so
