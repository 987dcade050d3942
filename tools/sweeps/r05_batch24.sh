# round 5, batch 24: adjoint segments up to 64 planes by the resident-set rule (IRS_BWD_MAX_SEG 64) against 32: parity at full size, bits, A/B, slab ranks
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests/test_gpu_transition.py tests/test_gpu_slab.py -m gpu -x -q -k "full_size or four_slabs or config4 or one_data_term" > gpurun_out/r05_t_seg64.txt 2>&1; rc=$?; tail -4 gpurun_out/r05_t_seg64.txt
fault gpurun_out/r05_t_seg64.txt
[ $rc -ne 0 ] && exit $rc
for f in gpurun_variants/bwdseg32.so gpurun_variants/bwdseg64.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_seg64_chain_bits.txt 2>&1
fault gpurun_out/r05_seg64_chain_bits.txt
cat gpurun_out/r05_seg64_chain_bits.txt
{
echo "# adjoint step: longest segment the resident-set rule may pick, 32 against 64 planes; one box, alternating"
timeout -k 10 500 bash tools/ab.sh gpurun_variants/bwdseg32.so gpurun_variants/bwdseg64.so 3 --steps 40
for lib in gpurun_variants/bwdseg32.so gpurun_variants/bwdseg64.so gpurun_variants/bwdseg32.so gpurun_variants/bwdseg64.so; do
  IRS_LIB=$PWD/$lib python tools/slab_probe.py --size 256 --worlds 2,4,8 --steps 30 > gpurun_out/s.json 2> gpurun_out/s.err; fault gpurun_out/s.err
  echo "$lib slab ranks (ms per transition of one rank): $(python -c "
import json;d=json.load(open('gpurun_out/s.json'));print({k:round(v['ms'],4) for k,v in d.items() if k.startswith('rank_of')})")"
done
} > gpurun_out/r05_bwd_seg64_ab.txt 2>&1
fault gpurun_out/r05_bwd_seg64_ab.txt
grep -v amdgpu.ids gpurun_out/r05_bwd_seg64_ab.txt
