# bench one workload under a list of environment settings (one "K=V K=V" group per line of stdin, empty line = defaults);
# run ON THE GPU BOX from the repo root:   bash tools/sweep_env.sh --size 128 < settings.txt
while IFS= read -r kv; do
  env $kv python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5 "$@" > gpurun_out/s.json 2> gpurun_out/s.err && python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('[$kv]',round(d['ms_per_step'],4),'bwd',round(d['roofline']['avg_launch_ms'],4),'fwd',round(d['exp_step_fwd']['avg_launch_ms'],4),'smooth',round(s['smooth_ms'],3),'data',round(s['data_ms'],3),'upd',round(s['update_ms'],3))" || tail -3 gpurun_out/s.err
done
