#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_eval.py tests/test_gpu_geom.py -q -m gpu 2>&1 | grep "passed\|failed\|AssertionError:" | head -3; done
