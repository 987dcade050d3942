for d in 8 16 24 6 30; do
  IRS_DBG=$d python bench.py --no-cpu-baseline --steps 10 > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));print('dbg',$d,round(d['ms_per_step'],3),'bwd',round(d['roofline']['avg_launch_ms'],4),'fwd',round(d['exp_step_fwd']['avg_launch_ms'],4))"
done
