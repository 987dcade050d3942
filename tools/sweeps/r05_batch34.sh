# round 5, batch 34: own-voxel phase of the any-radius adjoint with compile-time field layouts (ownlay4: 128-register cap, ownlay3: compiled for three waves per SIMD) against HEAD
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
for f in head ownlay4 ownlay3; do echo $f; CHAIN_BITS_DISPLACED=1 IRS_LIB=$PWD/gpurun_variants/$f.so timeout -k 10 300 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids | tail -3; done > gpurun_out/r05_ownlay_chain_bits.txt 2>&1
fault gpurun_out/r05_ownlay_chain_bits.txt
cat gpurun_out/r05_ownlay_chain_bits.txt
one() { IRS_LIB=$PWD/gpurun_variants/$1.so python tools/two_chain_run.py $2 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; fault gpurun_out/s.err; }
{
echo "# own-voxel phase of the any-radius adjoint with compile-time layouts; ms per transition, one box, alternating"
for r in 1 2 3; do
  for f in head ownlay4 ownlay3; do
    echo "$f | 256^3 wave 6 $(one $f '--size 256 --chains 1 --steps 30 --init wave --amp 6') | 256^3 wave 12 $(one $f '--size 256 --chains 1 --steps 30 --init wave --amp 12')"
  done
done
} > gpurun_out/r05_ownlay_ab.txt 2>&1
cat gpurun_out/r05_ownlay_ab.txt
