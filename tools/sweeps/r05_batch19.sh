# round 5, batch 19: the reference's regime (128^3, two chains) kernel by kernel -- launch shapes and rocprofv3 kernel stats
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python tools/launch_shapes.py > gpurun_out/r05_launch_shapes.txt 2> gpurun_out/r05_launch_shapes.err; rc=$?
fault gpurun_out/r05_launch_shapes.err
[ $rc -ne 0 ] && { tail -5 gpurun_out/r05_launch_shapes.err; exit $rc; }
grep -A22 "two chains" gpurun_out/r05_launch_shapes.txt
rm -rf gpurun_out/prof2c
rocprofv3 --kernel-trace --stats -d gpurun_out/prof2c -o c2 --output-format csv -- python3 tools/two_chain_run.py --steps 100 > gpurun_out/r05_two_chain_prof.log 2>&1; rc=$?
fault gpurun_out/r05_two_chain_prof.log
tail -2 gpurun_out/r05_two_chain_prof.log
f=$(find gpurun_out/prof2c -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r05_128_two_chains_kernel_stats.csv && head -30 gpurun_out/r05_128_two_chains_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof2c
exit $rc
