# round 5, batch 23: adjoint step with segments LONGER than 32 planes at 256^3 (one round of 1024 workgroups x 66 steps instead of two of 34)
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" python tools/two_chain_run.py $EXTRA 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; grep -l "Memory access fault" gpurun_out/s.err > /dev/null 2>&1 && { echo "GPU FAULT"; exit 9; }; return 0; }
{
echo "# 256^3, one chain: ms per transition against the adjoint's segment length (IRS_MARCH_SEG; default: 32 by the resident-set rule); one box, three rounds"
EXTRA="--size 256 --chains 1 --steps 40"
for r in 1 2 3; do
  for kn in "IRS_NONE=0" "IRS_MARCH_SEG=64" "IRS_MARCH_SEG=128" "IRS_MARCH_SEG=52"; do
    echo "256^3 C=1 | $kn | $(run $kn)"
  done
done
} > gpurun_out/r05_bwd_long_seg_sweep.txt 2>&1
cat gpurun_out/r05_bwd_long_seg_sweep.txt
