# round 5, first GPU batch: the LDS tap probe, the run-in peel of the adjoint A/B (128^3, 256^3, digests), then the suite with durations
mkdir -p gpurun_out
timeout -k 10 240 gpurun_out/lds_tap_probe 3000 > gpurun_out/r05_lds_tap_probe.txt 2>&1 || echo "probe failed" >> gpurun_out/r05_lds_tap_probe.txt
tail -3 gpurun_out/r05_lds_tap_probe.txt
cp ir_sgmcmc_amd/csrc/libirsgmcmc.so gpurun_variants/peel.so
for rep in 1 2 3; do timeout -k 10 300 bash tools/sweep_lib.sh --size 128 --steps 60; done > gpurun_out/r05_peel_ab_128.txt 2>&1
for rep in 1 2; do timeout -k 10 300 bash tools/sweep_lib.sh; done > gpurun_out/r05_peel_ab_256.txt 2>&1
cat gpurun_out/r05_peel_ab_128.txt gpurun_out/r05_peel_ab_256.txt
for f in gpurun_variants/*.so; do echo $f; IRS_LIB=$PWD/$f timeout -k 10 200 python tools/debug/chain_bits.py; done > gpurun_out/r05_peel_chain_bits.txt 2>&1
cat gpurun_out/r05_peel_chain_bits.txt
rm -f gpurun_variants/peel.so
python -m pytest tests -m gpu -x -q --durations=80 > gpurun_out/r05_suite_before.txt 2>&1; tail -3 gpurun_out/r05_suite_before.txt
