for s in 16 22 32 43 64; do
  IRS_MARCH_SEG=$s python bench.py --no-cpu-baseline --steps 15 > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));print('bwdseg',$s,round(d['ms_per_step'],3),round(d['roofline']['avg_launch_ms'],4),round(d['exp_step_fwd']['avg_launch_ms'],4))"
done
for s in 16 22 43 64 128; do
  IRS_MARCH_SEG_FWD=$s python bench.py --no-cpu-baseline --steps 15 > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));print('fwdseg',$s,round(d['ms_per_step'],3),round(d['roofline']['avg_launch_ms'],4),round(d['exp_step_fwd']['avg_launch_ms'],4))"
done
