# segment lengths of the stencil kernels at 256^3 (run ON THE GPU BOX from the repo root)
for kv in "IRS_SOBOLEV_SEG=8" "IRS_SOBOLEV_SEG=16" "IRS_SOBOLEV_SEG=32" "IRS_LCC_SEG=8" "IRS_LCC_SEG=16" "IRS_LCC_SEG=32" "IRS_UPDATE_SEG=8" "IRS_UPDATE_SEG=16" "IRS_UPDATE_SEG=32" "IRS_STATS_SEG=16"; do
  env $kv python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('$kv',round(d['ms_per_step'],3),'smooth',round(s['smooth_ms'],3),'data',round(s['data_ms'],3),'upd',round(s['update_ms'],3))"
done
