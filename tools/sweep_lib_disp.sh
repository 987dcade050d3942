# bench every tuning build under gpurun_variants/ on displaced chain starts (IRS_LIB selects the shared library)
for amp in 6 12; do
for f in gpurun_variants/*.so; do
  IRS_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extras --steps 15 --init wave --init-amp $amp "$@" > gpurun_out/s.json 2>gpurun_out/s.err && python -c "
import json,sys;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('$f amp $amp',round(d['ms_per_step'],3),'exp_bwd',round(s.get('exp_bwd_ms',0),3),'exp_fwd',round(s.get('exp_fwd_ms',0),3))"
done; done
