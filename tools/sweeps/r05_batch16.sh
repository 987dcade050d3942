set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
for f in gpurun_variants/rec16all.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_rec16all_chain_bits.txt 2>&1
fault gpurun_out/r05_rec16all_chain_bits.txt; cat gpurun_out/r05_rec16all_chain_bits.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/rec16all.so 3 --size 128 --steps 300 > gpurun_out/r05_rec16all_ab_128.txt 2>&1
fault gpurun_out/r05_rec16all_ab_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_rec16all_ab_128.txt
rm -f gpurun_variants/base.so
