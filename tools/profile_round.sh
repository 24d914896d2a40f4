# usage: bash tools/profile_round.sh r01   — kernel trace + HBM traffic counters of the default bench command (run on the GPU box)
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o p -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_trace.log 2>&1 || echo trace failed
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -o p -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_fetch.log 2>&1 || echo fetch failed
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -o p -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_write.log 2>&1 || echo write failed
ls $R/gpurun_out/${TAG}_trace
