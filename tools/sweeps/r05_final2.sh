# the last GPU batches of round 5 (after the launch-shape changes): full default suite, default bench line, round profile (trace + PMC
# passes), the two-chain regime kernel by kernel, launch shapes
set -o pipefail
mkdir -p gpurun_out
bash tools/sweeps/r05_final.sh || exit $?
bash tools/profile_round.sh r05b > gpurun_out/r05b_profile.log 2>&1; rc=$?; tail -3 gpurun_out/r05b_profile.log
grep -l "Memory access fault" gpurun_out/profile_r05b/*.log && exit 9
[ $rc -ne 0 ] && exit $rc
bash tools/sweeps/r05_batch19.sh > gpurun_out/r05_batch19.log 2>&1; rc=$?; tail -3 gpurun_out/r05_batch19.log
exit $rc
