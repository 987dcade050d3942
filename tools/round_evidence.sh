set -x
python -m pytest tests -m gpu -q > gpurun_out/t_full.log 2>&1; tail -2 gpurun_out/t_full.log
bash tools/profile_round.sh r03c > gpurun_out/prof_r03c.log 2>&1
bash tools/kstats.sh --size 128 > /dev/null 2>&1; cp gpurun_out/kstats/k_kernel_stats.csv gpurun_out/r03_128_kernel_stats.csv
bash tools/kstats.sh --init wave --init-amp 6 > /dev/null 2>&1; cp gpurun_out/kstats/k_kernel_stats.csv gpurun_out/r03_displaced6_kernel_stats.csv
python tools/slab_probe.py --worlds 1,2,4,8 --steps 30 > gpurun_out/slab_probe_256.txt 2>&1
python tools/slab_probe.py --loss ssd --worlds 8 --steps 30 > gpurun_out/slab_probe_256_ssd.txt 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -3 gpurun_out/bench_final.err; cut -c1-400 gpurun_out/bench_final.json
