# bench every tuning build under gpurun_variants/ (IRS_LIB selects the shared library); extra bench args passed through.
# A build whose run dies (a GPU memory fault aborts the process) ENDS the sweep: no further GPU work behind a fault.
for f in gpurun_variants/*.so; do
  IRS_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extras --steps 10 "$@" > gpurun_out/s.json 2> gpurun_out/s.err
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "$f FAILED (rc $rc): $(grep -v amdgpu.ids gpurun_out/s.err | tail -2 | tr '\n' ' ')"
    if [ $rc -ge 128 ] || grep -q "Memory access fault" gpurun_out/s.err; then echo "sweep stopped"; exit $rc; fi
    continue
  fi
  python -c "
import json,sys;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('$f',round(d['ms_per_step'],3),'bwd',round(d['roofline']['avg_launch_ms'],4),'fwd',round(d['exp_step_fwd']['avg_launch_ms'],4),'smooth',round(s['smooth_ms'],3),'data',round(s['data_ms'],3),'upd',round(s['update_ms'],3))"
done
