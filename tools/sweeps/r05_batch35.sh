# round 5, batch 35: bounds helper of the any-radius adjoint (coarse_minmax_kernel) with the row loads of 1 / 2 / 4 planes of a cell in flight
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
{
echo "# coarse_minmax_kernel: planes of a cell in flight (IRS_CMM_UNROLL_Z 1 / 2 / 4): its own time from rocprofv3 on a chain started 6 voxels away at 256^3, and ms per transition"
for f in cmm1 cmm2 cmm4; do
  rm -rf gpurun_out/profcmm
  IRS_LIB=$PWD/gpurun_variants/$f.so rocprofv3 --kernel-trace --stats -d gpurun_out/profcmm -o c --output-format csv -- python3 tools/two_chain_run.py --size 256 --chains 1 --steps 20 --init wave --amp 6 > gpurun_out/s.log 2>&1
  fault gpurun_out/s.log
  k=$(find gpurun_out/profcmm -name "*kernel_stats.csv" | head -1)
  echo "$f $(grep coarse_minmax $k | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('coarse_minmax calls',r[1],'avg us',round(float(r[3])/1e3,1),'max us',round(float(r[6])/1e3,1))")"
done
rm -rf gpurun_out/profcmm
} > gpurun_out/r05_cmm_unroll_ab.txt 2>&1
cat gpurun_out/r05_cmm_unroll_ab.txt
