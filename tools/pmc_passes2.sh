cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_PERF_SEL_TOTAL_READ"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_$i -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc2_$i.log 2>&1
done
