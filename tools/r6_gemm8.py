"""gemm8 against gemm7: the isolated Linear + bias + ReLU launch at the headline shape (find_linear_relu_fwd), both kernels' distance from a float64
evaluation, and their time.  python tools/r6_gemm8.py [n_feet n_pts]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib
L = _lib.lib()
n_feet, n_pts = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 6890)
rows = n_feet * n_pts
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(rows, 256, generator=g)).cuda()
w = (torch.randn(256, 256, generator=g) / 16).cuda()
b = torch.randn(256, generator=g).cuda()
s = torch.cuda.current_stream()
_lib.set_tuning('mlp_f16', 2)
ref = torch.relu(x[:20000].double() @ w.double().t() + b.double())
tail = torch.relu(x[-5000:].double() @ w.double().t() + b.double())
outs = {}
for k8 in (0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1):
	_lib.set_tuning('gemm8', k8)
	y = torch.full_like(x, float('nan'))
	def launch():
		_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y), ctypes.c_void_p(s.cuda_stream)), 'lin')
	for _ in range(60):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(100):
		launch()
	e1.record(); e1.synchronize()
	err = max(float((y[:20000].double() - ref).abs().max()), float((y[-5000:].double() - tail).abs().max()))
	print(f'gemm8={k8}: {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us   max |y - float64| {err:.3e}   nan {int(torch.isnan(y).sum())}')
	outs[k8] = y
print('max |gemm8 - gemm7|', float((outs[0] - outs[1]).abs().max()), [float((outs[0] - outs[k]).abs().max()) for k in range(2, 9)])
_lib.set_tuning('mlp_f16', 0)
