#!/bin/bash
# Host-side AddressSanitizer run of the library's pure-host code (round planner, slab layout, schedule builder, ABI checks):
# the device code is compiled as usual, only the host pass is instrumented (GPU ASan is not available on this pool).
# Runs WITHOUT a GPU:  bash tools/asan_host.sh      -> builds /tmp/irs_asan/libirsgmcmc_asan.so, runs the CPU schedule tests on it
set -e
OUT=/tmp/irs_asan
mkdir -p "$OUT"
cd "$(dirname "$0")/../ir_sgmcmc_amd/csrc"
for f in field_kernels exp_kernels data_kernels stencil_kernels scalar_kernels api comm ipc slab; do
  hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fvisibility=hidden -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $f.hip -o "$OUT/$f.o" &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address -shared-libsan -Wl,--version-script=exports.map -o "$OUT/libirsgmcmc_asan.so" "$OUT"/*.o -ldl -lrt
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd ../..
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_PRELOAD=$RT IRS_LIB=$OUT/libirsgmcmc_asan.so python -m pytest tests/test_slab_schedule.py tests/test_abi.py -x -q
