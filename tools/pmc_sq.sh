# SQ counter passes over the default bench (separate --pmc runs, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_sq -o p$i -- python3 /root/repo/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /root/repo/gpurun_out/pmc_sq_$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
