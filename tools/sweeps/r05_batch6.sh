# round 5, sixth GPU batch: buffer-load staging of the forward step (and both) A/B, the chain-overlap test on the final default,
# then the round profile: kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes, SQ counter passes
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests/test_gpu_transition.py -m gpu -x -q -k "chain_overlap" > gpurun_out/r05_t_overlap.txt 2>&1; rc=$?; tail -3 gpurun_out/r05_t_overlap.txt
fault gpurun_out/r05_t_overlap.txt
[ $rc -ne 0 ] && exit $rc
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
for f in gpurun_variants/fbufload.so gpurun_variants/bufboth.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_fbufload_chain_bits.txt 2>&1
fault gpurun_out/r05_fbufload_chain_bits.txt
cat gpurun_out/r05_fbufload_chain_bits.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/fbufload.so 3 --steps 60 > gpurun_out/r05_fbufload_ab_256.txt 2>&1
fault gpurun_out/r05_fbufload_ab_256.txt
timeout -k 10 500 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/fbufload.so 3 --size 128 --steps 200 > gpurun_out/r05_fbufload_ab_128.txt 2>&1
fault gpurun_out/r05_fbufload_ab_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_fbufload_ab_256.txt gpurun_out/r05_fbufload_ab_128.txt
rm -f gpurun_variants/base.so
bash tools/profile_round.sh r05 2>&1 | tail -5
bash tools/pmc_sq.sh 2>&1 | tail -4
python tools/sq_aggregate.py r05 | tail -4
rm -rf gpurun_out/pmc_sq gpurun_out/profile_r05/trace gpurun_out/profile_r05/pmc_fetch gpurun_out/profile_r05/pmc_write
ls gpurun_out/profile_r05
