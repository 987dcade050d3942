set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
for f in gpurun_variants/base.so gpurun_variants/schedilp.so gpurun_variants/schedminreg.so gpurun_variants/schedclause.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_sched_chain_bits.txt 2>&1
fault gpurun_out/r05_sched_chain_bits.txt; cat gpurun_out/r05_sched_chain_bits.txt
for rep in 1 2; do timeout -k 10 400 bash tools/sweep_lib.sh --steps 40; done > gpurun_out/r05_sched_256.txt 2>&1
fault gpurun_out/r05_sched_256.txt
for rep in 1 2; do timeout -k 10 400 bash tools/sweep_lib.sh --size 128 --steps 200; done > gpurun_out/r05_sched_128.txt 2>&1
fault gpurun_out/r05_sched_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_sched_256.txt gpurun_out/r05_sched_128.txt
rm -f gpurun_variants/base.so
