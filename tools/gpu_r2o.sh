#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_train3d.py -q -m gpu -x 2>&1 | tail -3
bash tools/gpu_r2m.sh 2>&1 | grep "fused_chain\|dw2_kernel\|gemm4"
bash tools/gpu_r2n.sh 2>&1 | grep -v amdgpu
