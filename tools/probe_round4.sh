set -e
mkdir -p gpurun_out
for g in 4 8 12; do
  timeout -k 10 280 python tools/slab_probe.py --transport ipc --worlds 2,4 --ghost-max $g 2> gpurun_out/r04_ipc_probe_g$g.err | grep -v Gloo > gpurun_out/r04_ipc_probe_g$g.json
  timeout -k 10 280 python tools/slab_probe.py --worlds 1,2,4,8 --ghost-max $g 2> gpurun_out/r04_null_probe_g$g.err | grep -v Gloo > gpurun_out/r04_null_probe_g$g.json
done
timeout -k 10 280 python tools/slab_probe.py --worlds 8 --ghost-max 8 --loss ssd 2>/dev/null | grep -v Gloo > gpurun_out/r04_null_probe_ssd_g8.json
timeout -k 10 280 python tools/slab_probe.py --worlds 8 --ghost-max 8 --size 128 2>/dev/null | grep -v Gloo > gpurun_out/r04_null_probe_128_g8.json
