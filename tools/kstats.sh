# per-kernel time summary of the bench (rocprofv3 kernel trace + stats); extra bench args are passed through
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/kstats -o k -- python3 /root/repo/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "$@" > /root/repo/gpurun_out/kstats.log 2>&1
cut -d, -f1-5 /root/repo/gpurun_out/kstats/k_kernel_stats.csv | head -40
