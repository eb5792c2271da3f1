#!/usr/bin/env python3
"""bf16x3 (fp32-faithful on the bf16 matrix pipe, mlp_gemm6.h) beside the exact-fp32 MFMA kernel (gemm4) and the fp16 mode (gemm5):
time per launch of find_linear_relu_fwd at a given shape and the error of each against a float64 product.
Usage: python3 tools/prof_x3.py [iters] [n_feet] [n_pts]"""
import ctypes
import os
os.environ.setdefault('FIND_DIAG', '1')   # laboratory build (include/find_hip_diag.h): this tool uses what the product library does not carry
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_feet = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 6890
L = _lib.lib()
rows = n_feet * n_pts
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(rows, 256, generator=g)).cuda()          # post-ReLU activations, as the layers see them
x[:, :8] *= torch.logspace(-6, 3, 8).cuda()                          # a few columns of very small / large magnitude
w = (torch.randn(256, 256, generator=g) / 16).cuda()
b = torch.randn(256, generator=g).cuda()
y = torch.empty_like(x)
s = torch.cuda.current_stream()
sub = slice(0, 8192)
ref = torch.relu(x[sub].double() @ w.double().t() + b.double())
scale = ref.abs().max().item()


def run(mode, name):
	_lib.set_tuning('mlp_f16', mode)
	def launch():
		_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
	for _ in range(100):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters):
		launch()
	e1.record()
	e1.synchronize()
	ms = e0.elapsed_time(e1) / iters
	d = (y[sub].double() - ref).abs()
	rel = (d / ref.abs().clamp(min=1e-30))[ref.abs() > 1e-3 * scale]
	print(f'{name:28s} {n_feet}x{n_pts}: {ms * 1e3:7.1f} us  {2.0 * rows * 65536 / ms / 1e9:7.1f} TF/s (fp32-equivalent)   max |err| {d.max().item():.3e} '
		  f'({d.max().item() / scale:.2e} of the largest output), mean |err| {d.mean().item():.3e}, max relative err {rel.max().item():.2e}')
	_lib.set_tuning('mlp_f16', 0)
	return y.clone()


y4 = run(0, 'gemm4 (fp32 MFMA)')
_lib.set_tuning('gemm7', 0)
y6 = run(2, 'gemm6 (bf16x3, W in LDS)')
_lib.set_tuning('gemm7', 1)
y7 = run(2, 'gemm7 (bf16x3, W in regs)')
print('gemm7 vs gemm6: max |diff| %.3e' % (y7 - y6).abs().max().item())
y5 = run(1, 'gemm5 (fp16 operands)')
print('gemm6 vs gemm4: max |diff| %.3e' % (y6 - y4).abs().max().item())

if os.environ.get('FIND_DBG'):
	_lib.set_tuning('mlp_f16', 2)
	dbg = torch.zeros(256 * 4, dtype=torch.int64, device='cuda')
	_lib.set_tuning('dbg', dbg.data_ptr())
	_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
	torch.cuda.synchronize()
	_lib.set_tuning('dbg', 0); _lib.set_tuning('mlp_f16', 0)
	d = dbg.view(256, 4).double().cpu() * 10.0   # 100 MHz ticks -> ns
	print('gemm7 per workgroup (wave 0), ns: total %.0f (min %.0f max %.0f); at the barrier %.0f (%.1f%%), k loop %.0f (%.1f%%), epilogue %.0f (%.1f%%)' % (
		d[:, 0].mean(), d[:, 0].min(), d[:, 0].max(), d[:, 1].mean(), 100 * d[:, 1].sum() / d[:, 0].sum(), d[:, 3].mean(), 100 * d[:, 3].sum() / d[:, 0].sum(),
		d[:, 2].mean(), 100 * d[:, 2].sum() / d[:, 0].sum()))
