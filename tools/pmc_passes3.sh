cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
         "TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_$i -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc3_$i.log 2>&1 || echo "pass $i failed"
done
