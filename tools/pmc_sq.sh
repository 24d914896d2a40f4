# usage: bash tools/pmc_sq.sh TAG — SQ issue/occupancy counters of the default bench command, 2 passes (run on the GPU box)
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
         "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_$i -o p -- python3 $R/bench.py --steps 32 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_$i.log 2>&1 || echo pass $i failed
  python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_$i
done
