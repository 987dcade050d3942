# round 5, batch 33: source boxes of the any-radius adjoint's tiles from a one-wavefront-per-tile kernel (IRS_TILE_BOX=1) against the in-kernel rounds (0): bits, timing, tests
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
for z in 0 1; do echo "IRS_TILE_BOX=$z"; CHAIN_BITS_DISPLACED=1 IRS_TILE_BOX=$z timeout -k 10 300 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_tile_box_chain_bits.txt 2>&1
fault gpurun_out/r05_tile_box_chain_bits.txt
cat gpurun_out/r05_tile_box_chain_bits.txt
python - <<'PY' || exit 7
t=open('gpurun_out/r05_tile_box_chain_bits.txt').read().split('IRS_TILE_BOX=')
a,b=[x.split('\n',1)[1].strip() for x in t[1:3]]
assert a==b and 'v ' in a, 'digests differ'
print('digests equal')
PY
one() { python tools/two_chain_run.py $1 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; fault gpurun_out/s.err; }
{
echo "# any-radius adjoint: source boxes / scales of its tiles from tile_box_kernel (IRS_TILE_BOX=1) against every workgroup its own (0); ms per chain-transition, one box, alternating"
for r in 1 2 3; do
  for z in 0 1; do
    export IRS_TILE_BOX=$z
    echo "IRS_TILE_BOX=$z | 256^3 wave 6 $(one '--size 256 --chains 1 --steps 30 --init wave --amp 6') | 256^3 wave 12 $(one '--size 256 --chains 1 --steps 30 --init wave --amp 12') | 256^3 wave 3 $(one '--size 256 --chains 1 --steps 30 --init wave --amp 3') | 128^3 C=2 wave 6 $(one '--steps 100 --init wave --amp 6') | 256^3 at rest $(one '--size 256 --chains 1 --steps 30')"
  done
done
unset IRS_TILE_BOX
for z in 0 1 0 1; do
  IRS_TILE_BOX=$z python bench.py --loss ssd --no-cpu-baseline --no-extras --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('IRS_TILE_BOX=$z SSD 256^3', round(d['ms_per_step'],4))"
done
} > gpurun_out/r05_tile_box_ab.txt 2>&1
cat gpurun_out/r05_tile_box_ab.txt
python -m pytest tests/test_gpu_transition.py tests/test_gpu_ops.py tests/test_gpu_slab.py -m gpu -x -q -k "displac or large or any_radius or lds or coarse or wave or big" 2>&1 | tail -3
