#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 600 python bench.py --train3d-b1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2n/b1.json
python - <<PY
import json
d=json.loads(open('gpurun_out/r2n/b1.json').read())
for k,v in d.items(): print(k, {a:b for a,b in v.items() if a in ('ms_per_step','host_enqueue_ms_per_step','error')})
PY
FIND_TUNING=fused_max_units=0 timeout 600 python bench.py --train3d-b1 --no-cpu-baseline --no-graph 2>/dev/null | tail -1 > gpurun_out/r2n/b1_unfused.json
python - <<PY
import json
d=json.loads(open('gpurun_out/r2n/b1_unfused.json').read())
for k,v in d.items(): print('unfused', k, {a:b for a,b in v.items() if a in ('ms_per_step','host_enqueue_ms_per_step','error')})
PY
