# round 5, batch 27: the final library against the library of the 230.2 /s line (commit a2a6cbb), one box, alternating: 256^3 and the reference's regime
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
one() { IRS_LIB=$PWD/$1 python tools/two_chain_run.py $2 2> gpurun_out/s.err | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_transition'],4))"; fault gpurun_out/s.err; }
{
echo "# final library of round 5 against the library of commit a2a6cbb (before the launch-shape changes); one box, alternating"
timeout -k 10 500 bash tools/ab.sh gpurun_variants/prev_a2a6cbb.so gpurun_variants/final.so 3 --steps 40
for r in 1 2 3; do
  for lib in gpurun_variants/prev_a2a6cbb.so gpurun_variants/final.so; do
    echo "$lib | 128^3 C=2 $(one $lib '--steps 200') | 128^3 C=1 $(one $lib '--chains 1 --steps 200') | 192^3 C=2 $(one $lib '--size 192 --steps 60') | 256^3 wave 6 $(one $lib '--size 256 --chains 1 --steps 30 --init wave --amp 6')"
  done
done
} > gpurun_out/r05_final_vs_a2a6cbb.txt 2>&1
fault gpurun_out/r05_final_vs_a2a6cbb.txt
grep -v amdgpu.ids gpurun_out/r05_final_vs_a2a6cbb.txt
