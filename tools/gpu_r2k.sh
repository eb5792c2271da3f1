#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 > gpurun_out/r2k/pytest_full.log
grep -n "backward\|pix_to_face\|mask:\|grad:\|worst\|passed\|failed\|FAILED\|Error" gpurun_out/r2k/pytest_full.log | head -40
