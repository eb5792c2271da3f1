#!/usr/bin/env python3
"""Launch only the dominant kernel (find_linear_relu_fwd at the C2 shape) N times -- a clean target for rocprofv3 --pmc.
Usage: python3 tools/prof_linear.py [iters] [n_feet] [n_pts] [mode] [ablate]   mode 4 = gemm4 (default routing), 3 = gemm3 for every size"""
import ctypes
import os
os.environ.setdefault('FIND_DIAG', '1')   # laboratory build (include/find_hip_diag.h): this tool uses what the product library does not carry
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_feet = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 6890
L = _lib.lib()
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 4
if mode == 3:
	_lib.set_tuning('gemm4_min_units', 10 ** 12)
	_lib.set_tuning('gemm4_small', 0)
_lib.set_tuning('ablate', int(sys.argv[5]) if len(sys.argv) > 5 else 0)
rows = n_feet * n_pts
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, 256, generator=g).cuda()
w = (torch.randn(256, 256, generator=g) / 16).cuda()
b = torch.randn(256, generator=g).cuda()
y = torch.empty_like(x)
s = torch.cuda.current_stream()
for _ in range(3):
	_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
	_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / iters
ref = torch.relu(x[:4096] @ w.t() + b)
err = (y[:4096] - ref).abs().max().item()
if os.environ.get('FIND_DBG'):
	dbg = torch.zeros(256 * 4, dtype=torch.int64, device='cuda')
	_lib.set_tuning('dbg', dbg.data_ptr())
	_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
	torch.cuda.synchronize()
	_lib.set_tuning('dbg', 0)
	d = dbg.view(256, 4).double().cpu()
	print('cycles per WG: total %.0f  wait+barrier %.0f (%.1f%%)  epilogue %.0f (%.1f%%)  lgkm %.0f (%.1f%%)' % (
		d[:, 0].mean(), d[:, 1].mean(), 100 * d[:, 1].sum() / d[:, 0].sum(), d[:, 2].mean(), 100 * d[:, 2].sum() / d[:, 0].sum(),
		d[:, 3].mean(), 100 * d[:, 3].sum() / d[:, 0].sum()), ' max total %.0f min total %.0f' % (d[:, 0].max(), d[:, 0].min()))
ref_all = torch.relu(x @ w.t() + b)
err = (y - ref_all).abs().max().item()
print(f'mode {mode} linear_relu {n_feet}x{n_pts}: {ms*1e3:.1f} us  {2.0*rows*65536/ms/1e9:.1f} TF/s  max_err_vs_torch {err:.2e}')
