"""Diagnosis of the co-residence fault (mlp.hip; it turned out to be about waves that own the full accumulator set in a partial register allocation, not LDS): dw2_repro_kernel compares every published
ring stage with the same bytes in HBM and logs the mismatching 16-byte pieces with the workgroup's HW_REG_LDS_ALLOC (LDS base / size),
HW_ID and whether the piece still holds what the stage held three chunks ago.  python tools/probe_lds_fault.py [passes] [lds_exclusive]"""
import os, sys
os.environ.setdefault('FIND_DIAG', '1')   # laboratory build (include/find_hip_diag.h): this tool uses what the product library does not carry
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd import _lib
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
excl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')
model, params, step = bench.build_step(dev, 0)
_lib.set_tuning('lds_exclusive', excl)
_lib.set_tuning('dw_lds_free', 3)   # the probe lives in dw2's body; 3 = dw2_repro_kernel, the kernel as it was when the fault was found
_lib.set_tuning('fused_max_units', 0)
log = torch.zeros(8 + 64 * 8, dtype=torch.int64, device=dev)
_lib.set_tuning('dw2_verify', log.data_ptr())
for _ in range(passes):
	step()
torch.cuda.synchronize()
L = log.cpu().tolist()
print(f'lds_exclusive={excl}: {passes} passes, {L[1]} stages checked, {L[2]} of them in workgroups with LDS base != 0, {L[0]} mismatching 16-byte pieces')
for n in range(min(L[0], 64)):
	r = L[8 + n * 8: 16 + n * 8]
	split, chunk = r[0] & 0xFFFFFFFF, (r[0] >> 32) & 0xFFFFFFFF
	stage, tid, piece = r[1] & 0xFF, (r[1] >> 8) & 0xFFFF, (r[1] >> 24) & 0xFF
	alloc = r[2] & 0xFFFFFFFF
	import struct
	got = struct.unpack('f', struct.pack('I', r[4] & 0xFFFFFFFF))[0]
	want = struct.unpack('f', struct.pack('I', (r[4] >> 32) & 0xFFFFFFFF))[0]
	print(f'  split {split:4d} chunk {chunk:3d} stage {stage} row {tid >> 3:2d} piece {tid & 7} +{piece * 16:3d}B  lds_base {alloc & 0xFF:3d} (x256B?) lds_size {(alloc >> 12) & 0x1FF:3d}'
		  f'  hw_id {r[3] & 0xFFFFFFFF:#010x} xcc {r[6] & 0xF}  got {got:+.5e} want {want:+.5e}  stale={r[5]}')
_lib.set_tuning('dw2_verify', 0); _lib.set_tuning('lds_exclusive', 0); _lib.set_tuning('dw_lds_free', 1)
