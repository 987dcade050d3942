# round 5, batch 29: knob sweep at 128^3, ONE chain: ms per transition, two rounds
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" python tools/two_chain_run.py --chains 1 --steps 300 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; grep -l "Memory access fault" gpurun_out/s.err > /dev/null 2>&1 && { echo "GPU FAULT"; exit 9; }; return 0; }
{
echo "# 128^3, one chain, 300 transitions (median of three repetitions): ms per transition per knob setting (one box, two rounds)"
for r in 1 2; do
  for kn in IRS_NONE=0 IRS_UPDATE_SEG=8 IRS_UPDATE_SEG=16 IRS_LCC_SEG=8 IRS_LCC_SEG=6 IRS_STATS_SEG=8 IRS_STATS_SEG=2 IRS_SOBOLEV_SEG=8 IRS_SOBOLEV_SEG=16 IRS_PS_ROWS=16 IRS_FWD_PF=1 IRS_MARCH_SEG=16 IRS_MARCH_SEG_FWD=16 IRS_SEG_FIT=0 IRS_SWZ_RUN=0 IRS_NONE=1; do
    echo "$kn $(run $kn)"
  done
done
} > gpurun_out/r05_128_knob_sweep.txt 2>&1
cat gpurun_out/r05_128_knob_sweep.txt
