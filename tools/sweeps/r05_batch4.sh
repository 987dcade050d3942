# round 5, fourth GPU batch: full trimmed suite, phase traces with per-step rows, stencil swizzle A/B (long runs, alternating), 128^3 env sweep
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r05_suite_after.txt 2>&1; rc=$?; tail -32 gpurun_out/r05_suite_after.txt
fault gpurun_out/r05_suite_after.txt
[ $rc -ne 0 ] && exit $rc
IRS_LIB=$PWD/gpurun_variants/other/trace.so timeout -k 10 200 python tools/fwd_phase_trace.py --size 128 > gpurun_out/r05_fwd_trace_128.txt 2>&1
IRS_LIB=$PWD/gpurun_variants/other/bwdtrace.so timeout -k 10 200 python tools/bwd_phase_trace.py --size 128 > gpurun_out/r05_bwd_trace_128.txt 2>&1
fault gpurun_out/r05_fwd_trace_128.txt gpurun_out/r05_bwd_trace_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_fwd_trace_128.txt gpurun_out/r05_bwd_trace_128.txt
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/swzrows4.so 3 --steps 100 > gpurun_out/r05_swz_ab_256.txt 2>&1
fault gpurun_out/r05_swz_ab_256.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/swzrows4.so 3 --size 128 --steps 300 > gpurun_out/r05_swz_ab_128.txt 2>&1
fault gpurun_out/r05_swz_ab_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_swz_ab_256.txt gpurun_out/r05_swz_ab_128.txt
rm -f gpurun_variants/base.so
timeout -k 10 400 bash tools/sweep_env.sh --size 128 --steps 200 < tools/sweeps/r05_seg128.txt > gpurun_out/r05_seg128.txt 2>&1
fault gpurun_out/r05_seg128.txt
grep -v amdgpu.ids gpurun_out/r05_seg128.txt
