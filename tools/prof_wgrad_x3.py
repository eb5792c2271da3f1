#!/usr/bin/env python3
"""dW = dZ^T X in isolation (find_linear_wgrad) in the three arithmetics: time per call (kernel + slab reduce) and error against float64."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_feet = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 6890
L = _lib.lib()
rows = n_feet * n_pts
g = torch.Generator().manual_seed(0)
dz = (torch.randn(rows, 256, generator=g) * 0.1).cuda()
x = torch.relu(torch.randn(rows, 256, generator=g)).cuda()
dw = torch.empty(256, 256, device='cuda'); db = torch.empty(256, device='cuda')
nb = L.find_linear_wgrad_scratch_bytes(n_feet)
scratch = torch.empty(nb // 4, device='cuda')
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = (dz.double().t() @ x.double()); refb = dz.double().sum(0)
scale = ref.abs().max().item()
for mode, name in ((0, 'dw4 (fp32 MFMA)'), (2, 'dw6 (bf16x3)'), (1, 'dw3 (fp16 operands)')):
	_lib.set_tuning('mlp_f16', mode)
	def launch():
		_lib.check(L.find_linear_wgrad(_lib.ctx(), _lib.ptr(dz), _lib.ptr(x), n_feet, n_pts, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch), nb, s), 'wgrad')
	for _ in range(60):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters):
		launch()
	e1.record(); e1.synchronize()
	ms = e0.elapsed_time(e1) / iters
	d = (dw.double() - ref).abs()
	print(f'{name:24s} {n_feet}x{n_pts}: {ms * 1e3:7.1f} us  {2.0 * rows * 65536 / ms / 1e9:7.1f} TF/s (fp32-equivalent)  max |err| {d.max().item() / scale:.2e} of the largest entry, mean {d.mean().item() / scale:.2e}; '
		  f'bias {((db.double() - refb).abs().max() / refb.abs().max()).item():.1e}')
_lib.set_tuning('mlp_f16', 0)
