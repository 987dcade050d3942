# round 5, batch 26: the radius-2 variants of both squaring steps on their own segment lengths (fitted to THEIR resident set) against the
# radius-1 kernel's: bits of displaced chains, then A/B on chains started 3 / 6 / 12 voxels away and at rest
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
for f in gpurun_variants/bwdseg32.so gpurun_variants/bwdseg64.so gpurun_variants/r2seg.so; do echo $f; CHAIN_BITS_DISPLACED=1 IRS_LIB=$PWD/$f timeout -k 10 300 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_r2seg_chain_bits.txt 2>&1
fault gpurun_out/r05_r2seg_chain_bits.txt
cat gpurun_out/r05_r2seg_chain_bits.txt
one() { IRS_LIB=$PWD/$1 python tools/two_chain_run.py $2 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; fault gpurun_out/s.err; }
{
echo "# 256^3, one chain: ms per transition; bwdseg32 = round 5 so far, bwdseg64 = adjoint segments up to 64 planes (the radius-2 variants on the radius-1 kernel's segments), r2seg = + the radius-2 variants on their own; one box, alternating"
for r in 1 2 3; do
  for lib in gpurun_variants/bwdseg32.so gpurun_variants/bwdseg64.so gpurun_variants/r2seg.so; do
    echo "$lib | at rest $(one $lib '--size 256 --chains 1 --steps 40') | wave 3 $(one $lib '--size 256 --chains 1 --steps 30 --init wave --amp 3') | wave 6 $(one $lib '--size 256 --chains 1 --steps 30 --init wave --amp 6') | wave 12 $(one $lib '--size 256 --chains 1 --steps 30 --init wave --amp 12') | 128^3 C=2 wave 6 $(one $lib '--steps 100 --init wave --amp 6')"
  done
done
} > gpurun_out/r05_r2seg_ab.txt 2>&1
cat gpurun_out/r05_r2seg_ab.txt
