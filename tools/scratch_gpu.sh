#!/bin/bash
set -x
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_train3d.py tests/test_gpu_pipeline.py tests/test_gpu_mlp_f16.py -x -q 2>&1 | tail -15 > $O/tests_new.txt
timeout 900 python tools/fused_ab.py > $O/fused_ab.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
