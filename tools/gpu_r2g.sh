#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_pipeline.py tests/test_gpu_texture.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r2g/pytest.log
tail -4 gpurun_out/r2g/pytest.log
timeout 900 bash tools/prof_raster.sh r2g/raster > gpurun_out/r2g/raster.txt 2>&1
cat gpurun_out/r2g/raster.txt
timeout 300 python bench.py --subpaths --no-cpu-baseline 2>/dev/null | head -3 > gpurun_out/r2g/subpaths.jsonl
python - <<PY
import json
for l in open('gpurun_out/r2g/subpaths.jsonl'):
	d=json.loads(l); print(d['workload'], 'fwd %.2f ms  fwd+bwd %.2f ms'%(d['ms_fwd'],d['ms_fwd_bwd']))
PY
