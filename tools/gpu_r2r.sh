#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed\|FAILED\|Error" | head
timeout 900 python bench.py > gpurun_out/r2r/bench.json 2> gpurun_out/r2r/bench.err
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu" gpurun_out/r2r/bench.err | tail -5
python - <<PY
import json
d=json.loads(open('gpurun_out/r2r/bench.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['value'], d.get('x_cpu_baseline'), d['roofline']['frac'])
for k,v in d['records'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','host_enqueue_ms_per_step','error','x_cpu_baseline','step_frac_of_fp32_mfma_peak_executed')})
PY
