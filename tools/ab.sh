# A/B of two builds of the library on ONE box, alternating (A B A B ...): `bash tools/ab.sh A.so B.so ROUNDS [bench args]`.
# One line per run: ms per transition, adjoint / forward launch ms, stage ms.  Stops behind a GPU fault.
A=$1; B=$2; R=$3; shift 3
for r in $(seq 1 $R); do
  for f in $A $B; do
    IRS_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/s.json 2> gpurun_out/s.err
    rc=$?
    if [ $rc -ne 0 ]; then echo "$f FAILED (rc $rc): $(grep -v amdgpu.ids gpurun_out/s.err | tail -2 | tr '\n' ' ')"; exit $rc; fi
    python -c "
import json;d=json.load(open('gpurun_out/s.json'));s=d['stage_ms'];print('$f',round(d['ms_per_step'],4),'bwd',round(d['roofline']['avg_launch_ms'],4),'fwd',round(d['exp_step_fwd']['avg_launch_ms'],4),'smooth',round(s['smooth_ms'],3),'data',round(s['data_ms'],3),'upd',round(s['update_ms'],3))"
  done
done
