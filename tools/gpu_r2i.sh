#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_pipeline.py tests/test_gpu_texture.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r2i/pytest.log
tail -4 gpurun_out/r2i/pytest.log
timeout 300 python tools/prof_raster_ablate.py 6890 256 0 1 2 3 4 0 2>&1 | grep ablate
timeout 300 python tools/prof_raster_ablate.py 6890 512 0 3 4 2>&1 | grep ablate
timeout 600 bash tools/prof_raster.sh r2i/raster quick 2>&1 | head -8
