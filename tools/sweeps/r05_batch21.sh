# round 5, batch 21: the forward step's launch shape in the reference's regime (128^3, two chains) and at 192^3 C = 2 / 256^3: segment length x rows per thread
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" python tools/two_chain_run.py --steps 200 $EXTRA 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; grep -l "Memory access fault" gpurun_out/s.err > /dev/null 2>&1 && { echo "GPU FAULT"; exit 9; }; return 0; }
{
echo "# forward step: segment length (IRS_MARCH_SEG_FWD) x one row per thread (IRS_FWD_ROWS1) x planes of prefetch; ms per chain-transition, one box, two rounds"
for r in 1 2; do
  EXTRA=""
  for kn in "IRS_NONE=0" "IRS_MARCH_SEG_FWD=12" "IRS_MARCH_SEG_FWD=16" "IRS_MARCH_SEG_FWD=22" "IRS_MARCH_SEG_FWD=32" "IRS_FWD_ROWS1=1" "IRS_FWD_ROWS1=1 IRS_FWD_PF=1" "IRS_FWD_ROWS1=1 IRS_MARCH_SEG_FWD=8" "IRS_FWD_ROWS1=1 IRS_MARCH_SEG_FWD=32" "IRS_FWD_ROWS1=1 IRS_MARCH_SEG_FWD=22"; do
    echo "128^3 C=2 | $kn | $(run $kn)"
  done
  EXTRA="--size 192 --steps 60"
  for kn in "IRS_NONE=0" "IRS_MARCH_SEG_FWD=32" "IRS_MARCH_SEG_FWD=48" "IRS_FWD_ROWS1=1"; do
    echo "192^3 C=2 | $kn | $(run $kn)"
  done
  EXTRA="--size 256 --chains 1 --steps 40"
  for kn in "IRS_NONE=0" "IRS_MARCH_SEG_FWD=64" "IRS_MARCH_SEG_FWD=43"; do
    echo "256^3 C=1 | $kn | $(run $kn)"
  done
  EXTRA="--size 128 --chains 1 --steps 200"
  for kn in "IRS_NONE=0" "IRS_MARCH_SEG_FWD=16" "IRS_FWD_ROWS1=0 IRS_MARCH_SEG_FWD=16" "IRS_FWD_ROWS1=0 IRS_MARCH_SEG_FWD=32"; do
    echo "128^3 C=1 | $kn | $(run $kn)"
  done
done
} > gpurun_out/r05_fwd_shape_sweep.txt 2>&1
cat gpurun_out/r05_fwd_shape_sweep.txt
