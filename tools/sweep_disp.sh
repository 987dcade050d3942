# transition time vs displacement magnitude (exercises the radius-2 and fallback variants of the squaring-step kernels)
for a in 0.5 1.5 3 6; do
  python bench.py --no-cpu-baseline --steps 10 --init smooth --init-amp $a > gpurun_out/s.json && python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('amp',$a,'ms',round(d['ms_per_step'],3),'fwd',round(s['exp_fwd_ms'],3),'bwd',round(s['exp_bwd_total_ms'],3))"
done
