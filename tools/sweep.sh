# usage: bash tools/sweep.sh 8 16 24 ...   (MSNE_REFILL values) — prints Mrays/s and kernel ms per setting
for t in "$@"; do
  MSNE_REFILL=$t python bench.py --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('refill=$t', 'Mrays/s=%.1f'%d['value'], 'ms/step=%.2f'%d['ms_per_step'], 'closest=%.1f shadow=%.1f shade=%.1f render=%.1f'%(k['trace_closest'],k['trace_shadow'],k['shade'],k['render']))"
done
