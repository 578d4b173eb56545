python bench.py --no-cpu-baseline --no-track-leg --no-elas-leg --steps 10 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(' '.join('%s %.4f' % (k, v['avg_ms']) for k, v in d['kernels'].items()), 'step %.3f' % d['ms_per_step'])
"
