# round 5, batch 18: the data terms of all chains as ONE launch (data_batch): its test, the C = 2 parity tests, A/B against the serial form
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests/test_gpu_transition.py tests/test_trajectory.py -m gpu -x -q -k "one_data_term or chain_overlap or fixture or builder_variants or trajectory" > gpurun_out/r05_t_batch.txt 2>&1; rc=$?; tail -5 gpurun_out/r05_t_batch.txt
fault gpurun_out/r05_t_batch.txt
[ $rc -ne 0 ] && exit $rc
python - > gpurun_out/r05_data_batch_ab.txt 2>&1 <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import side_run
dev = torch.device('cuda', 0)
print('# several chains in one engine: the data terms of all chains as ONE launch behind the serial statistics -> step loop (data_batch 1) against one launch per chain (0); one box, alternating; ms per chain-transition')
for rep in range(3):
    for mode in (1, 0):
        from ir_sgmcmc_amd import _lib as L
        L.check(L.load().irs_option_set(None, b'data_batch', mode))
        r = side_run(128, 'gmm', 'identity', 0.0, 200, 20, dev, chains=2)
        r2 = side_run(192, 'gmm', 'identity', 0.0, 60, 10, dev, chains=2)
        r3 = side_run(128, 'gmm', 'identity', 0.0, 120, 20, dev, chains=4)
        print('data_batch', mode, '128^3 C=2:', round(r['ms_per_transition'], 4), '| 192^3 C=2:', round(r2['ms_per_transition'], 4), '| 128^3 C=4:', round(r3['ms_per_transition'], 4), flush=True)
PY
fault gpurun_out/r05_data_batch_ab.txt
grep -v amdgpu.ids gpurun_out/r05_data_batch_ab.txt
