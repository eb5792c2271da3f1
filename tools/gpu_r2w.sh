#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "lds_exclusive=0" "lds_exclusive=0 reduce_exclusive=1" "lds_exclusive=0 bwd_streams=0" "lds_exclusive=0 reduce_stream=0"; do
  echo "== $cfg (250 passes, 16 x 6890, fused chains on)"
  timeout 400 python tools/check_determinism.py 250 16 6890 0 $cfg 2>&1 | grep -v amdgpu | grep "mismatches\|differs" | cut -c1-150 | tail -4
done
