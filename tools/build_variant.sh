# build an alternative libirsgmcmc.so with extra compile flags into gpurun_variants/<name>.so (for tools/sweep_lib.sh)
#   bash tools/build_variant.sh planar_lds -DIRS_FWD_LDS_PLANAR
set -e
name=$1; shift
src=ir_sgmcmc_amd/csrc
out=gpurun_variants
mkdir -p $out /tmp/irs_variant_$name
for f in field_kernels exp_kernels data_kernels stencil_kernels scalar_kernels api comm ipc slab; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -fvisibility=hidden "$@" -c $src/$f.hip -o /tmp/irs_variant_$name/$f.o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $out/$name.so /tmp/irs_variant_$name/*.o -Wl,--version-script=$src/exports.map -ldl -lrt
echo built $out/$name.so
