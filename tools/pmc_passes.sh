cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  case $i in
    1) C="FETCH_SIZE";;
    2) C="TCC_HIT_sum TCC_MISS_sum";;
    3) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY";;
  esac
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r01_$i -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$i.log 2>&1
done
ls -R $R/gpurun_out/pmc_r01_1 | head
