#!/usr/bin/env python3
"""The bf16x3 Linear kernel (gemm7; gemm6 with gemm7=0) under its ablation variants ("x3_abl" of libfind_hip_diag.so): which resource bounds it?
1 = no split arithmetic / LDS writes, 2 = no LDS fragment reads after the first, 4 = no activation loads after the prologue, 8 = no MFMAs; "ablate"
512 / 1024 = every unit reads / writes the range's first rows (results are wrong under every one of them: timing only).
Arguments: further x3_abl values."""
import ctypes, os, sys
os.environ.setdefault('FIND_DIAG', '1')   # laboratory build (include/find_hip_diag.h): this tool uses what the product library does not carry
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib
L = _lib.lib()
n_feet, n_pts = 16, 6890
rows = n_feet * n_pts
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(rows, 256, generator=g)).cuda()
w = (torch.randn(256, 256, generator=g) / 16).cuda()
b = torch.randn(256, generator=g).cuda()
y = torch.empty_like(x)
s = torch.cuda.current_stream()
_lib.set_tuning('mlp_f16', 2)
for abl, bits in [(0, 0), (0, 0), (0, 512), (0, 1024), (0, 1536), (12, 0), (3, 0), (15, 0)] + [(int(a), 0) for a in sys.argv[1:]]:
	_lib.set_tuning('x3_abl', abl); _lib.set_tuning('ablate', bits)
	def launch():
		_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
	for _ in range(60):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(40):
		launch()
	e1.record(); e1.synchronize()
	print(f'x3_abl {abl:2d} ablate {bits:4d}: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us')
_lib.set_tuning('x3_abl', 0); _lib.set_tuning('ablate', 0); _lib.set_tuning('mlp_f16', 0)
