for m in 0 1; do
  IRS_PREDICT_VARIANTS=$m python bench.py --no-cpu-baseline --steps 15 > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('predict',$m,round(d['ms_per_step'],3),'bwd',round(d['roofline']['avg_launch_ms'],4),'fwd',round(d['exp_step_fwd']['avg_launch_ms'],4), round(s['exp_fwd_ms'],3), round(s['exp_bwd_total_ms'],3))"
done
