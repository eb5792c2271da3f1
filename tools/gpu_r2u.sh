#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== A: lds_exclusive=0, fused off (the round-1 reproduction), 60 passes at 16 x 6890"
timeout 300 python tools/check_determinism.py 60 16 6890 0 lds_exclusive=0 fused_max_units=0 2>&1 | grep -v amdgpu | grep "mismatches\|differs" | tail -4
echo "== B: same with the stage verification of dw2 switched on"
cat > /tmp/ver.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from find_amd import _lib
torch.zeros(1, device='cuda')
log = torch.zeros(8 + 64 * 8, dtype=torch.int64, device='cuda')
_lib.set_tuning('dw2_verify', log.data_ptr())
sys.argv = ['x', '60', '16', '6890', '0', 'lds_exclusive=0', 'fused_max_units=0']
exec(open(os.environ['GRAFT_REPO_ROOT'] + '/tools/check_determinism.py').read())
L = log.cpu().tolist()
print('stage verification:', L[1], 'stages,', L[2], 'with LDS base != 0,', L[0], 'mismatching pieces')
PY
timeout 300 python /tmp/ver.py 2>&1 | grep -v amdgpu | grep "mismatches\|differs\|verification" | tail -5
echo "== C: lds_exclusive=0 with the fused chains (default path)"
timeout 300 python tools/check_determinism.py 60 16 6890 0 lds_exclusive=0 2>&1 | grep -v amdgpu | grep "mismatches\|differs" | tail -3
echo "== D: default (reservation on), 100 passes"
timeout 300 python tools/check_determinism.py 100 16 6890 0 2>&1 | grep -v amdgpu | grep "mismatches\|differs" | tail -3
