# round 5, seventh GPU batch: XCD run length of the squaring steps, kernel stats at 128^3 and for the displaced start (+ its last
# transition launch by launch), the compute side of one slab rank (1 / 2 / 4 / 8) and concurrent ranks over ipc, BASELINE config 5
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
timeout -k 10 400 bash tools/sweep_env.sh --steps 60 < tools/sweeps/r05_swz.txt > gpurun_out/r05_swz_run.txt 2>&1
fault gpurun_out/r05_swz_run.txt; grep -v amdgpu.ids gpurun_out/r05_swz_run.txt
bash tools/kstats.sh --size 128 > /dev/null 2>&1; cp gpurun_out/kstats/k_kernel_stats.csv gpurun_out/r05_128_kernel_stats.csv; rm -rf gpurun_out/kstats
bash tools/kstats.sh --init wave --init-amp 6 > /dev/null 2>&1; cp gpurun_out/kstats/k_kernel_stats.csv gpurun_out/r05_displaced6_kernel_stats.csv; rm -rf gpurun_out/kstats
bash tools/step_trace.sh displaced6 --init wave --init-amp 6 > gpurun_out/r05_displaced6_step_trace.txt 2>&1; rm -rf gpurun_out/step_trace_displaced6/trace
fault gpurun_out/kstats.log gpurun_out/r05_displaced6_step_trace.txt
cat gpurun_out/r05_displaced6_step_trace.txt | grep -v amdgpu.ids | tail -45
timeout -k 10 300 python tools/slab_probe.py --worlds 1,2,4,8 --ghost-max 8 2> gpurun_out/r05_null_probe_g8.err | grep -v Gloo > gpurun_out/r05_slab_probe_256_g8.json
fault gpurun_out/r05_null_probe_g8.err; cat gpurun_out/r05_slab_probe_256_g8.json | head -50
timeout -k 10 300 python tools/slab_probe.py --transport ipc --worlds 2,4 --ghost-max 8 2> gpurun_out/r05_ipc_probe_g8.err | grep -v Gloo > gpurun_out/r05_slab_ipc_256_g8.json
fault gpurun_out/r05_ipc_probe_g8.err; cat gpurun_out/r05_slab_ipc_256_g8.json | head -40
timeout -k 10 400 python run.py -c configs/experiment1_192_synthetic.json > gpurun_out/r05_config5_192_run.log 2>&1; tail -4 gpurun_out/r05_config5_192_run.log
fault gpurun_out/r05_config5_192_run.log
