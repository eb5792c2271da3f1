#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k two_ranks 2>&1 | grep "passed\|failed\|AssertionError" | head -3; done
echo "== fused off"
for i in 1 2; do FIND_TUNING=fused_max_units=0 timeout 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k two_ranks 2>&1 | grep "passed\|failed\|AssertionError" | head -3; done
echo "== fwd_streams off"
for i in 1 2; do FIND_TUNING=fwd_streams=0 timeout 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k two_ranks 2>&1 | grep "passed\|failed\|AssertionError" | head -3; done
