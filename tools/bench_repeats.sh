# the default `python bench.py` N times in a row on one box: the scalars of every line (run-to-run spread)
N=${1:-5}
for i in $(seq 1 $N); do python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
keep=('value','ms_per_step','frontend','multi_sequence','sharded','semantic_elas','elas','msa','host_feed','host_feed_pageable','frontend_host_feed','with_null_stream_cotenant','with_pooled_stream_cotenant','two_contexts_one_gpu','frame_period_us')
o={k:d.get(k) for k in keep}; o['roofline_frac']=d['roofline']['frac']; o['kernel_seconds_per_launch']=d['roofline']['kernel_seconds_per_launch']; o['cpu_baseline']=d['cpu_baseline']['value']; o['checks_all_true']=all(v is True or v==0 for v in d['checks'].values())
print(json.dumps(o))"; done
