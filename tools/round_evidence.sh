# everything profiles/r04_* holds that is not the round profile (tools/profile_round.sh), re-measured in one go on the GPU box
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; tail -2 gpurun_out/r04_gpu_suite.log
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; tail -2 gpurun_out/r04_bench_final.err
python run.py -c configs/experiment1_192_synthetic.json > gpurun_out/r04_config5_192_run.log 2>&1; tail -3 gpurun_out/r04_config5_192_run.log
bash tools/kstats.sh --size 128 > /dev/null 2>&1; cp gpurun_out/kstats/k_kernel_stats.csv gpurun_out/r04_128_kernel_stats.csv
bash tools/kstats.sh --init wave --init-amp 6 > /dev/null 2>&1; cp gpurun_out/kstats/k_kernel_stats.csv gpurun_out/r04_displaced6_kernel_stats.csv
bash tools/probe_round4.sh
