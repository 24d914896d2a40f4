# usage (on the GPU box): bash tools/valu_class_probe.sh   → gpurun_out/r03_valu_classes.txt
# Which SQ_INSTS_VALU_* class does the profiler book each instruction kind of tools/valu_microbench.hip under?  (bench.py prices a kernel's instructions by class.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/valu_microbench "$R/tools/valu_microbench.hip" || exit 1
/tmp/valu_microbench > /tmp/vm_names.txt 2>&1
rm -rf /tmp/vc1 /tmp/vc2
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d /tmp/vc1 -o p -- /tmp/valu_microbench > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --kernel-trace --output-format csv -d /tmp/vc2 -o p -- /tmp/valu_microbench > /dev/null 2>&1
python3 - <<'PY' > "$R/gpurun_out/r03_valu_classes.txt"
import csv, glob, re, collections
names = []
for l in open('/tmp/vm_names.txt'):
    if l.startswith('#') or l.startswith('instruction'): continue
    n = l[:80].strip()
    if n and n not in names: names.append(n)
val = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ('/tmp/vc1', '/tmp/vc2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'<\(?(?:Kind\))?(\d+)>', r['Kernel_Name'])
            if not m: continue
            val[int(m.group(1))][(d, r['Counter_Name'])] += float(r['Counter_Value'])
print('# share of a kernel\'s SQ_INSTS_VALU that each class counter books, per instruction kind of tools/valu_microbench.hip (loop overhead: < 2 %)')
print('%-72s %7s %7s %7s %7s %7s %7s' % ('kind', 'INT32', 'CVT', 'TRANS', 'FMA', 'MUL', 'ADD'))
for k in sorted(val):
    v = val[k]; t1 = v[('/tmp/vc1', 'SQ_INSTS_VALU')] or 1.0; t2 = v[('/tmp/vc2', 'SQ_INSTS_VALU')] or 1.0
    nm = names[k] if k < len(names) else str(k)
    print('%-72s %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f' % (nm[:72], v[('/tmp/vc1', 'SQ_INSTS_VALU_INT32')] / t1, v[('/tmp/vc1', 'SQ_INSTS_VALU_CVT')] / t1, v[('/tmp/vc1', 'SQ_INSTS_VALU_TRANS_F32')] / t1,
          v[('/tmp/vc2', 'SQ_INSTS_VALU_FMA_F32')] / t2, v[('/tmp/vc2', 'SQ_INSTS_VALU_MUL_F32')] / t2, v[('/tmp/vc2', 'SQ_INSTS_VALU_ADD_F32')] / t2))
PY
head -3 /tmp/vc1/*/*counter_collection.csv 2>/dev/null | cut -c1-300 >> "$R/gpurun_out/r03_valu_classes.txt"
