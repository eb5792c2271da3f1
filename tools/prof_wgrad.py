"""Time the weight-gradient kernels in isolation (find_linear_wgrad: dw2 + slab reduce; dw3 with FIND_TUNING=mlp_f16=1).
python tools/prof_wgrad.py [iters] [n_feet] [n_pts]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib  # noqa: E402
if os.environ.get('FIND_LIB'):   # a library variant under test
	_lib.LIB_PATH = os.environ['FIND_LIB']

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_feet = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 6890
L = _lib.lib()
rows = n_feet * n_pts
g = torch.Generator().manual_seed(0)
dz = (torch.randn(rows, 256, generator=g) * 0.1).cuda()
x = torch.relu(torch.randn(rows, 256, generator=g)).cuda()
dw = torch.empty(256, 256, device='cuda')
db = torch.empty(256, device='cuda')
nb = L.find_linear_wgrad_scratch_bytes(n_feet)
scratch = torch.empty(nb // 4, device='cuda')
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def launch():
	_lib.check(L.find_linear_wgrad(_lib.ctx(), _lib.ptr(dz), _lib.ptr(x), n_feet, n_pts, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch), nb, s), 'wgrad')


for _ in range(100):
	launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
	launch()
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * rows * 256 * 256
print(f'wgrad {n_feet}x{n_pts} (kernel + slab reduce): {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TFLOP/s  {2 * rows * 1024 / ms / 1e9:.2f} TB/s of operand reads')
