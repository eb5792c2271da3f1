#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_train3d.py tests/test_gpu_pipeline.py tests/test_gpu_mlp_f16.py tests/test_gpu_distributed.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r2l/pytest.log
tail -25 gpurun_out/r2l/pytest.log
timeout 600 python tools/fused_ab.py 2>&1 | grep -v amdgpu
