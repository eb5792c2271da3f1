#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_pipeline.py tests/test_gpu_texture.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r2j/pytest.log
cat gpurun_out/r2j/pytest.log
timeout 900 bash tools/prof_raster.sh r2j/raster > gpurun_out/r2j/raster.txt 2>&1
cat gpurun_out/r2j/raster.txt | head -60
timeout 300 python bench.py --subpaths --no-cpu-baseline 2>/dev/null | head -3 > gpurun_out/r2j/subpaths.jsonl
cat gpurun_out/r2j/subpaths.jsonl | cut -c1-900
