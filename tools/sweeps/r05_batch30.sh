# round 5, batch 30: knob sweep on ONE slab rank of 4 / 8 of the 256^3 workload (tools/slab_probe.py: the launch sequence of a rank, hand-overs stubbed): ms per transition
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" python tools/slab_probe.py --size 256 --worlds 4,8 --steps 30 > gpurun_out/s.json 2> gpurun_out/s.err; grep -l "Memory access fault" gpurun_out/s.err > /dev/null 2>&1 && { echo "GPU FAULT"; exit 9; }; python -c "
import json;d=json.load(open('gpurun_out/s.json'));print(' '.join(f\"{k} {v['ms']:.4f}\" for k,v in d.items() if k.startswith('rank_of')))"; }
{
echo "# one rank of 4 / 8 of the 256^3 workload: ms per transition per knob setting (one box, two rounds)"
for r in 1 2; do
  for kn in IRS_NONE=0 IRS_FWD_ROWS1=0 IRS_FWD_ROWS1=1 IRS_MARCH_SEG_FWD=16 IRS_MARCH_SEG_FWD=32 IRS_MARCH_SEG=16 IRS_MARCH_SEG=8 IRS_UPDATE_SEG=16 IRS_UPDATE_SEG=8 IRS_LCC_SEG=8 IRS_LCC_SEG=32 IRS_STATS_SEG=8 IRS_STATS_SEG=32 IRS_SOBOLEV_SEG=16 IRS_SLAB_SPLIT=0 IRS_SEG_FIT=0; do
    echo "$kn | $(run $kn)"
  done
done
} > gpurun_out/r05_slab_rank_knob_sweep.txt 2>&1
cat gpurun_out/r05_slab_rank_knob_sweep.txt
