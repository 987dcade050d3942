set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/fwdfma.so 3 --steps 60 > gpurun_out/r05_fwdfma_ab_256.txt 2>&1
fault gpurun_out/r05_fwdfma_ab_256.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/fwdfma.so 3 --size 128 --steps 200 > gpurun_out/r05_fwdfma_ab_128.txt 2>&1
fault gpurun_out/r05_fwdfma_ab_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_fwdfma_ab_256.txt gpurun_out/r05_fwdfma_ab_128.txt
rm -f gpurun_variants/base.so
IRS_LIB=$PWD/gpurun_variants/fwdfma.so python -m pytest tests/test_gpu_transition.py tests/test_gpu_ops.py tests/test_trajectory.py tests/test_gpu_fullsize.py tests/test_gpu_ops_fuzz.py tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r05_t_fwdfma.txt 2>&1; tail -12 gpurun_out/r05_t_fwdfma.txt
fault gpurun_out/r05_t_fwdfma.txt
cp gpurun_out/parity_report.json gpurun_out/r05_parity_report_fwdfma.json
