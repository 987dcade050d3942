set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
bash tools/probe_trace.sh 8 > gpurun_out/r05_probe_trace_8.txt 2>&1; rm -rf gpurun_out/probe_trace_8/trace
fault gpurun_out/r05_probe_trace_8.txt gpurun_out/probe_trace_8/probe.log
grep -v amdgpu.ids gpurun_out/r05_probe_trace_8.txt | tail -6
bash tools/probe_trace.sh 4 > gpurun_out/r05_probe_trace_4.txt 2>&1; rm -rf gpurun_out/probe_trace_4/trace
grep -v amdgpu.ids gpurun_out/r05_probe_trace_4.txt | tail -6
