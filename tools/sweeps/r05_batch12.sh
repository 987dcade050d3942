set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
for fl in host device; do
  echo "== flags $fl"
  IRS_IPC_FLAGS=$fl timeout -k 10 200 python tools/comm_probe.py 2> gpurun_out/r05_comm_probe_$fl.err | grep -v Gloo | tee gpurun_out/r05_comm_probe_$fl.json
  fault gpurun_out/r05_comm_probe_$fl.err
  IRS_IPC_FLAGS=$fl timeout -k 10 300 python tools/slab_probe.py --transport ipc --worlds 2,4 --ghost-max 8 2> gpurun_out/r05_ipc_probe_$fl.err | grep -v Gloo > gpurun_out/r05_slab_ipc_256_g8_$fl.json
  fault gpurun_out/r05_ipc_probe_$fl.err
  python -c "
import json;d=json.load(open('gpurun_out/r05_slab_ipc_256_g8_$fl.json'));print({k:(round(v['ipc_ms'],3),round(v['moves_nothing_ms'],3),round(v['handover_cost_ms'],3),v['transport'][:90]) for k,v in d.items() if k.startswith('ranks')})"
done
timeout -k 10 300 bash tools/sweep_env.sh --size 128 --steps 300 < tools/sweeps/r05_kernarg.txt > gpurun_out/r05_kernarg_128.txt 2>&1
fault gpurun_out/r05_kernarg_128.txt; grep -v amdgpu.ids gpurun_out/r05_kernarg_128.txt
