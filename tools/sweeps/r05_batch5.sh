# round 5, fifth GPU batch: the chain-overlap test, C = 2 at 128^3 with / without the overlap, the default bench line with extras
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests/test_gpu_transition.py -m gpu -x -q -k "chain_overlap or fixture or trajectory" > gpurun_out/r05_t_overlap.txt 2>&1; rc=$?; tail -5 gpurun_out/r05_t_overlap.txt
fault gpurun_out/r05_t_overlap.txt
[ $rc -ne 0 ] && exit $rc
python - > gpurun_out/r05_chain_overlap_ab.txt 2>&1 <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import side_run
dev = torch.device('cuda', 0)
for rep in range(3):
    for mode in ('1', '0'):
        os.environ['IRS_CHAIN_OVERLAP'] = mode
        from ir_sgmcmc_amd import _lib as L
        L.load().irs_option_set(None, b'chain_overlap', int(mode))
        r = side_run(128, 'gmm', 'identity', 0.0, 200, 20, dev, chains=2)
        r2 = side_run(192, 'gmm', 'identity', 0.0, 60, 10, dev, chains=2)
        print('chain_overlap', mode, '128^3 C=2 ms per chain-transition:', round(r['ms_per_transition'], 4), '| 192^3 C=2:', round(r2['ms_per_transition'], 4), flush=True)
PY
fault gpurun_out/r05_chain_overlap_ab.txt
grep -v amdgpu.ids gpurun_out/r05_chain_overlap_ab.txt
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
for f in gpurun_variants/base.so gpurun_variants/bufload.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_bufload_chain_bits.txt 2>&1
fault gpurun_out/r05_bufload_chain_bits.txt
cat gpurun_out/r05_bufload_chain_bits.txt
timeout -k 10 400 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/bufload.so 3 --steps 60 > gpurun_out/r05_bufload_ab_256.txt 2>&1
fault gpurun_out/r05_bufload_ab_256.txt
timeout -k 10 400 bash tools/ab.sh gpurun_variants/base.so gpurun_variants/bufload.so 3 --size 128 --steps 200 > gpurun_out/r05_bufload_ab_128.txt 2>&1
fault gpurun_out/r05_bufload_ab_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_bufload_ab_256.txt gpurun_out/r05_bufload_ab_128.txt
rm -f gpurun_variants/base.so
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; rc=$?
fault gpurun_out/r05_bench_default.err
python -c "
import json;d=json.load(open('gpurun_out/r05_bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['exp_step_fwd']['avg_launch_ms']);print({k:(round(v.get('ms_per_transition',0),4),round(v.get('transitions_per_s',0),1)) for k,v in d['also'].items()});print(d['cpu_baseline'])"
exit $rc
