for f in gpurun_variants/*.so; do
  echo "== $f"
  IRS_LIB=$PWD/$f python tools/lds_phase_trace.py 2>&1 | sed -n '2,5p;$p'
  IRS_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extras --steps 10 --init wave --init-amp 6 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('displaced ms', round(d['ms_per_step'],3))"
done
