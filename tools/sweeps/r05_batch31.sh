# round 5, batch 31: forward step of small launches with TWO planes per marching step (IRS_FWD_Z2) against one: bits, then timing
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
for z in 0 1; do echo "IRS_FWD_Z2=$z"; CHAIN_BITS_DISPLACED=1 IRS_FWD_Z2=$z timeout -k 10 300 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_z2_chain_bits.txt 2>&1
fault gpurun_out/r05_z2_chain_bits.txt
cat gpurun_out/r05_z2_chain_bits.txt
python - <<'PY' || exit 7
import re
t=open('gpurun_out/r05_z2_chain_bits.txt').read().split('IRS_FWD_Z2=')
a,b=[x.split('\n',1)[1].strip() for x in t[1:3]]
assert a==b and 'v ' in a, 'digests differ'
print('digests equal')
PY
one() { python tools/two_chain_run.py $1 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; fault gpurun_out/s.err; }
{
echo "# forward step of small launches: two planes per marching step (IRS_FWD_Z2=1) against one (0); ms per chain-transition, one box, alternating"
for r in 1 2 3; do
  for z in 0 1; do
    export IRS_FWD_Z2=$z
    echo "IRS_FWD_Z2=$z | 128^3 C=1 $(one '--chains 1 --steps 300') | 128^3 C=2 $(one '--steps 200') | 96^3 C=2 $(one '--size 96 --steps 200') | 64^3 C=2 $(one '--size 64 --steps 300')"
  done
done
for z in 0 1 0 1; do
  IRS_FWD_Z2=$z python tools/slab_probe.py --size 256 --worlds 4,8 --steps 30 > gpurun_out/s.json 2> gpurun_out/s.err; fault gpurun_out/s.err
  echo "IRS_FWD_Z2=$z slab ranks (ms per transition of one rank): $(python -c "
import json;d=json.load(open('gpurun_out/s.json'));print({k:round(v['ms'],4) for k,v in d.items() if k.startswith('rank_of')})")"
done
} > gpurun_out/r05_fwd_z2_ab.txt 2>&1
cat gpurun_out/r05_fwd_z2_ab.txt
