# transition time vs displacement magnitude (exercises the radius-2 and any-radius variants of the squaring-step kernels)
#   smooth: random low-resolution field, Sobolev-smoothed (high gradients);  wave: one half-wave across the volume (a converged
#   registration: large but smooth)
for init in ${INITS:-smooth wave}; do for a in ${AMPS:-0.5 1.5 3 6 12 24}; do
  python bench.py --no-cpu-baseline --no-extras --steps 10 --init $init --init-amp $a > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('$init amp',$a,'ms',round(d['ms_per_step'],3),'fwd',round(s['exp_fwd_ms'],3),'bwd',round(s['exp_bwd_total_ms'],3))"
done; done
