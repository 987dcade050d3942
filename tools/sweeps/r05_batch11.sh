set -o pipefail
mkdir -p gpurun_out
fault() { grep -l "Memory access fault" "$@" 2>/dev/null && { echo "GPU FAULT in $*"; exit 9; }; return 0; }
python -m pytest tests/test_gpu_transition.py -m gpu -x -q -k "capture" > gpurun_out/r05_t_capture.txt 2>&1; rc=$?; tail -15 gpurun_out/r05_t_capture.txt
fault gpurun_out/r05_t_capture.txt
[ $rc -ne 0 ] && exit $rc
IRS_IPC_FLAGS=device IRS_IPC_TIMEOUT_S=5 timeout -k 10 200 python -m pytest tests/test_gpu_slab.py -m gpu -x -q -k "exchange_ghost_planes and ipc and 32-2-4" > gpurun_out/r05_t_device_flags_one_gpu.txt 2>&1; echo "device flags, two ranks on one GPU: rc $?"; tail -4 gpurun_out/r05_t_device_flags_one_gpu.txt
fault gpurun_out/r05_t_device_flags_one_gpu.txt
IRS_LONG=1 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r05_suite_long.txt 2>&1; rc=$?; tail -16 gpurun_out/r05_suite_long.txt
fault gpurun_out/r05_suite_long.txt
exit $rc
