#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/ver.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch
from find_amd import _lib
torch.zeros(1, device='cuda')
log = torch.zeros(8 + 64 * 8, dtype=torch.int64, device='cuda')
_lib.set_tuning('dw2_verify', log.data_ptr())
sys.argv = ['x', '200', '16', '6890', '0', 'lds_exclusive=0']
exec(open(os.environ['GRAFT_REPO_ROOT'] + '/tools/check_determinism.py').read())
L = log.cpu().tolist()
print('stage verification:', L[1], 'stages,', L[2], 'with LDS base != 0,', L[0], 'mismatching pieces')
import struct
for n in range(min(L[0], 24)):
	r = L[8 + n * 8: 16 + n * 8]
	split, chunk = r[0] & 0xFFFFFFFF, (r[0] >> 32) & 0xFFFFFFFF
	stage, tid, piece = r[1] & 0xFF, (r[1] >> 8) & 0xFFFF, (r[1] >> 24) & 0xFF
	alloc = r[2] & 0xFFFFFFFF
	got = struct.unpack('f', struct.pack('I', r[4] & 0xFFFFFFFF))[0]
	want = struct.unpack('f', struct.pack('I', (r[4] >> 32) & 0xFFFFFFFF))[0]
	print(f'  split {split:4d} chunk {chunk:3d} stage {stage} row {tid >> 3:2d} piece {tid & 7} +{piece * 16:3d}B  LDS_ALLOC {alloc:#010x}  hw_id {r[3] & 0xFFFFFFFF:#010x} xcc {r[6] & 0xF}  got {got:+.5e} want {want:+.5e}  stale={r[5]}')
PY
timeout 600 python /tmp/ver.py 2>&1 | grep -v amdgpu | grep "mismatches\|differs\|verification\|split" | tail -40
