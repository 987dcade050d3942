for n in 64 128 192 256; do
  python bench.py --no-cpu-baseline --size $n --steps 20 > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('N',$n,'ms',round(d['ms_per_step'],3),'it/s',round(d['value'],1),'fwd',round(s['exp_fwd_ms'],3),'bwd',round(s['exp_bwd_total_ms'],3),'smooth',round(s['smooth_ms'],3),'data',round(s['data_ms'],3),'upd',round(s['update_ms'],3),'roofline_frac',round(d['transition_roofline']['frac_of_8TBps'],3))"
done
