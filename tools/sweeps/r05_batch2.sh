# round 5, second GPU batch: LDS tap probe (built on the box), the trimmed suite with durations, tap / tile variants of the forward
# step at 256^3 and 128^3, the forward phase trace at 128^3, the launch-shape table
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probes/lds_tap_probe.hip -o /tmp/lds_tap_probe 2> gpurun_out/r05_probe_build.txt \
  && timeout -k 10 300 /tmp/lds_tap_probe 3000 > gpurun_out/r05_lds_tap_probe.txt 2>&1
tail -14 gpurun_out/r05_lds_tap_probe.txt
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/base.so
for rep in 1 2; do timeout -k 10 400 bash tools/sweep_lib.sh; done > gpurun_out/r05_taps_ab_256.txt 2>&1
for rep in 1 2; do timeout -k 10 400 bash tools/sweep_lib.sh --size 128 --steps 60; done > gpurun_out/r05_taps_ab_128.txt 2>&1
for rep in 1 2; do IRS_FWD_ROWS1=1 timeout -k 10 400 bash tools/sweep_lib.sh --size 128 --steps 60; done > gpurun_out/r05_taps_ab_128_rows1.txt 2>&1
grep -h -v amdgpu.ids gpurun_out/r05_taps_ab_256.txt gpurun_out/r05_taps_ab_128.txt gpurun_out/r05_taps_ab_128_rows1.txt
rm -f gpurun_variants/base.so
IRS_LIB=$PWD/gpurun_variants/other/trace.so timeout -k 10 200 python tools/fwd_phase_trace.py --size 128 > gpurun_out/r05_fwd_trace_128.txt 2>&1
IRS_LIB=$PWD/gpurun_variants/other/trace.so timeout -k 10 200 python tools/fwd_phase_trace.py --size 256 > gpurun_out/r05_fwd_trace_256.txt 2>&1
cat gpurun_out/r05_fwd_trace_128.txt gpurun_out/r05_fwd_trace_256.txt | grep -v amdgpu.ids
timeout -k 10 300 python tools/launch_shapes.py > gpurun_out/r05_launch_shapes.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_launch_shapes.txt
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r05_suite_after.txt 2>&1; tail -30 gpurun_out/r05_suite_after.txt
