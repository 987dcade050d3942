# round 5, batch 22: one-row-per-thread forward step up to 1024 tiles (IRS_FWD_SMALL_TILES) against 640: fused engine sizes and one slab rank of 4 / 8
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
one() { IRS_LIB=$PWD/$1 python tools/two_chain_run.py $2 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; fault gpurun_out/s.err; }
{
echo "# ms per chain-transition; one box, alternating"
for r in 1 2 3; do
  for lib in gpurun_variants/small640.so gpurun_variants/small1024.so; do
    echo "$lib | 128^3 C=2 $(one $lib '--steps 200') | 128^3 C=4 $(one $lib '--chains 4 --steps 100') | 128^3 C=1 $(one $lib '--chains 1 --steps 200') | 160^3 C=1 $(one $lib '--size 160 --chains 1 --steps 100') | 96^3 C=2 $(one $lib '--size 96 --steps 200')"
  done
done
for lib in gpurun_variants/small640.so gpurun_variants/small1024.so gpurun_variants/small640.so gpurun_variants/small1024.so; do
  IRS_LIB=$PWD/$lib python tools/slab_probe.py --size 256 --worlds 4,8 --steps 30 > gpurun_out/s.json 2> gpurun_out/s.err; fault gpurun_out/s.err
  echo "$lib slab ranks: $(python -c "
import json;d=json.load(open('gpurun_out/s.json'));print({k:round(v['ms'],4) for k,v in d.items() if k.startswith('rank_of')})")"
done
} > gpurun_out/r05_fwd_small_tiles_ab.txt 2>&1
cat gpurun_out/r05_fwd_small_tiles_ab.txt
