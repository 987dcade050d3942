# round 5, third GPU batch: the corrected LDS tap probe, chain digests base (IRS_FWD_TAPS=1) vs taps0, variants at 256^3 / 128^3,
# forward phase trace, the full trimmed suite with durations
set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probes/lds_tap_probe.hip -o /tmp/lds_tap_probe 2> gpurun_out/r05_probe_build.txt \
  && timeout -k 10 300 /tmp/lds_tap_probe 3000 > gpurun_out/r05_lds_tap_probe.txt 2>&1
fault gpurun_out/r05_lds_tap_probe.txt
grep "run_len  1\|run_len  4" gpurun_out/r05_lds_tap_probe.txt | cut -c1-140
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
for f in gpurun_variants/base.so gpurun_variants/taps0.so gpurun_variants/bwdtaps1.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r05_taps_chain_bits.txt 2>&1
fault gpurun_out/r05_taps_chain_bits.txt
cat gpurun_out/r05_taps_chain_bits.txt
for rep in 1 2; do timeout -k 10 400 bash tools/sweep_lib.sh --steps 20; done > gpurun_out/r05_taps_ab_256.txt 2>&1
fault gpurun_out/r05_taps_ab_256.txt
for rep in 1 2; do timeout -k 10 400 bash tools/sweep_lib.sh --size 128 --steps 60; done > gpurun_out/r05_taps_ab_128.txt 2>&1
fault gpurun_out/r05_taps_ab_128.txt
grep -h -v amdgpu.ids gpurun_out/r05_taps_ab_256.txt gpurun_out/r05_taps_ab_128.txt
rm -f gpurun_variants/base.so
IRS_LIB=$PWD/gpurun_variants/other/trace.so timeout -k 10 200 python tools/fwd_phase_trace.py --size 128 > gpurun_out/r05_fwd_trace_128.txt 2>&1
IRS_LIB=$PWD/gpurun_variants/other/trace.so timeout -k 10 200 python tools/fwd_phase_trace.py --size 256 > gpurun_out/r05_fwd_trace_256.txt 2>&1
fault gpurun_out/r05_fwd_trace_128.txt gpurun_out/r05_fwd_trace_256.txt
cat gpurun_out/r05_fwd_trace_128.txt gpurun_out/r05_fwd_trace_256.txt | grep -v amdgpu.ids
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r05_suite_after.txt 2>&1; rc=$?; tail -32 gpurun_out/r05_suite_after.txt
fault gpurun_out/r05_suite_after.txt
exit $rc
