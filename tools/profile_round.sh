# usage (on the GPU box): bash tools/profile_round.sh TAG SCENE [PASSES] [extra bench.py args]
#   PASSES = comma list out of trace,fetch,write,sq,valu,mem,mem2 (default: all).  Kernel trace + PMC passes of
#   `python3 bench.py --scene SCENE --no-cpu-baseline --repeats 2 ...`, each in its OWN run (gpurun refuses --pmc combined with
#   sys/hip traces; the counters of one pass have to fit the hardware's per-block limits).  MSNE_SERIAL=1: kernels in stream order, so
#   per-kernel durations are not blurred by co-residency.  Scratch output under gpurun_out/TAG_SCENE_*; tools/profile_counters.py TAG SCENE
#   turns it into profiles/TAG_*_SCENE.*
TAG=${1:-r03}; SCENE=${2:-s1}; PASSES=${3:-trace,fetch,write,sq,valu,mem,mem2}
[ $# -ge 3 ] && shift 3 || shift $#
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export MSNE_SERIAL=1
O=$R/gpurun_out/${TAG}_${SCENE}
SC=${SCENE%_sky}; ENVARG=""; [ "$SC" != "$SCENE" ] && ENVARG="--env sky"
ARGS="--scene $SC $ENVARG --no-cpu-baseline --no-other-configs --sustain-seconds 0 --no-gpu-visits --repeats 2 $*"
echo "MSNE_SERIAL=1 bench.py $ARGS" > ${O}_cmd.txt
pmc() {  # name, counters...
    n=$1; shift
    case ",$PASSES," in *",$n,"*) ;; *) return;; esac
    timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d ${O}_$n -o p -- python3 $R/bench.py $ARGS > ${O}_$n.log 2>&1 || echo "$n failed"
}
case ",$PASSES," in *",trace,"*) timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_trace -o p -- python3 $R/bench.py $ARGS > ${O}_trace.log 2>&1 || echo trace failed;; esac
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
pmc valu SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU
pmc mem TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc mem2 TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD
for n in trace fetch write sq valu mem mem2; do [ -f ${O}_$n.log ] && { echo "== $n"; tail -c 300 ${O}_$n.log | tr '\n' ' ' | cut -c1-300; echo; }; done
