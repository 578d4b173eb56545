#!/bin/bash
# Which neighbour halves the many-sequence leg?  (1) alone, (2) beside a one-thread CPU burner, (3) beside an idle process
# that holds a GPU context with a dozen streams.
cd "$(dirname "$0")/.."
show() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k:(round(v['value']) if 'value' in v else {m:round(x['value']) for m,x in v.items()}) for k,v in d.items()})
" "$1" "$2"; }
timeout 300 python bench.py --tail-leg-child multi_sequence > gpurun_out/li_alone.json 2>/dev/null; show gpurun_out/li_alone.json alone
python3 -c "
while True: pass
" & B=$!
timeout 300 python bench.py --tail-leg-child multi_sequence > gpurun_out/li_burn.json 2>/dev/null; show gpurun_out/li_burn.json cpu_burner
kill $B
python3 -c "
import time, torch
ss=[torch.cuda.Stream() for _ in range(16)]
x=torch.zeros(1<<20,device='cuda')
for s in ss:
    with torch.cuda.stream(s): x+=1
torch.cuda.synchronize()
time.sleep(600)
" & B=$!
sleep 20
timeout 300 python bench.py --tail-leg-child multi_sequence > gpurun_out/li_idlectx.json 2>/dev/null; show gpurun_out/li_idlectx.json idle_gpu_process
kill $B
