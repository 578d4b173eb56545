#!/bin/bash
# The tail legs' child process under different GPU_MAX_HW_QUEUES
cd "$(dirname "$0")/.."
show() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k:(round(v['value']) if 'value' in v else {m:round(x['value']) for m,x in v.items()}) for k,v in d.items()})
" "$1" "$2"; }
for q in "$@"; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --tail-leg-child multi_sequence > gpurun_out/lq_$q.json 2>/dev/null; show gpurun_out/lq_$q.json queues=$q
done
