# round 5, batch 20: knob sweep in the reference's regime (128^3, two chains): ms per chain-transition, two rounds
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" python tools/two_chain_run.py --steps 200 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; grep -l "Memory access fault" gpurun_out/s.err > /dev/null 2>&1 && { echo "GPU FAULT"; exit 9; }; return 0; }
{
echo "# 128^3, two chains in one engine, 200 transitions: ms per chain-transition per knob setting (one box, two rounds)"
for r in 1 2; do
  for kn in IRS_NONE=0 IRS_FWD_PF=2 IRS_FWD_ROWS1=1 IRS_MARCH_SEG_FWD=16 IRS_MARCH_SEG_FWD=4 IRS_MARCH_SEG=8 IRS_MARCH_SEG=32 IRS_UPDATE_SEG=4 IRS_UPDATE_SEG=16 IRS_LCC_SEG=8 IRS_STATS_SEG=8 IRS_SOBOLEV_SEG=16 IRS_SOBOLEV_SEG=4 IRS_PS_ROWS=16 IRS_SEG_FIT=0; do
    echo "$kn $(run $kn)"
  done
done
} > gpurun_out/r05_two_chain_knob_sweep.txt 2>&1
cat gpurun_out/r05_two_chain_knob_sweep.txt
