# usage (on the GPU box): bash tools/profile_round2.sh TAG SCENE [extra bench.py args]
#   kernel trace + the PMC passes of `python3 bench.py --scene SCENE --no-cpu-baseline --repeats 2 ...`, each in its OWN run
#   (gpurun refuses --pmc combined with sys/hip traces; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Scratch output under
#   gpurun_out/TAG_SCENE_*; tools/profile_counters.py TAG SCENE turns it into profiles/TAG_*_SCENE.*
TAG=${1:-r02}; SCENE=${2:-s1}; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_${SCENE}
SC=${SCENE%_sky}; ENVARG=""; [ "$SC" != "$SCENE" ] && ENVARG="--env sky"
ARGS="--scene $SC $ENVARG --no-cpu-baseline --repeats 2 $*"
echo "bench.py $ARGS" > ${O}_cmd.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_trace -o p -- python3 $R/bench.py $ARGS > ${O}_trace.log 2>&1 || echo trace failed
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${O}_fetch -o p -- python3 $R/bench.py $ARGS > ${O}_fetch.log 2>&1 || echo fetch failed
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${O}_write -o p -- python3 $R/bench.py $ARGS > ${O}_write.log 2>&1 || echo write failed
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d ${O}_sq -o p -- python3 $R/bench.py $ARGS > ${O}_sq.log 2>&1 || echo sq failed
tail -c 600 ${O}_trace.log
