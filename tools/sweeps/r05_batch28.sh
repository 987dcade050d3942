# round 5, batch 28: final library against commit a2a6cbb with the DEFAULT bench loop (20 steps after 3 warm-up) and with 60 steps
set -o pipefail
mkdir -p gpurun_out
{
echo "# default loop (--steps 20 --warmup 3)"
timeout -k 10 400 bash tools/ab.sh gpurun_variants/prev_a2a6cbb.so gpurun_variants/final.so 4 --steps 20 --warmup 3
echo "# --steps 60 --warmup 10"
timeout -k 10 400 bash tools/ab.sh gpurun_variants/prev_a2a6cbb.so gpurun_variants/final.so 2 --steps 60 --warmup 10
} > gpurun_out/r05_final_vs_a2a6cbb_default_loop.txt 2>&1
grep -l "Memory access fault" gpurun_out/r05_final_vs_a2a6cbb_default_loop.txt && exit 9
grep -v amdgpu.ids gpurun_out/r05_final_vs_a2a6cbb_default_loop.txt
