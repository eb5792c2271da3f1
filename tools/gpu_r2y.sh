#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed\|FAILED\|Error" | head
bash tools/profile_round2.sh
