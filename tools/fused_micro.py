"""Where the time of fused_chain_kernel goes: forward chain at batch 1 (6890 rows, no grad) under the kernel's ablation bits
(find_ctx knob "ablate": 1 no W staging, 2 no MFMAs, 4 no epilogue, 8 no Fourier features).  python tools/fused_micro.py"""
import os, sys
os.environ.setdefault('FIND_DIAG', '1')   # laboratory build (include/find_hip_diag.h): this tool uses what the product library does not carry
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from find_amd import _lib, synthetic
dev = torch.device('cuda:0')
model = synthetic.make_model(6890, train_size=1, val_size=1, device=dev)
lat = synthetic.latents(1, seed=0, device=dev)
tex = len(sys.argv) > 1 and sys.argv[1] == 'tex'   # the texture pass's shape instead: 16 x 1000 per-foot points, colour head (64-row tiles)
if tex:
	model = synthetic.make_model(6890, train_size=16, val_size=1, device=dev)
	lat = synthetic.latents(16, seed=0, device=dev)
	pos = (torch.rand(16, 1000, 3, device=dev) * 0.2 - 0.1)
def fwd():
	with torch.no_grad():
		if tex:
			return model(pos, shapevec=lat['shapevec'], texvec=lat['texvec'], posevec=lat['posevec'], want=('col',))
		return model.get_meshes(shapevec=lat['shapevec'], reg=lat['reg'], texvec=lat['texvec'], posevec=lat['posevec'])
for ab in (0, 0, 1, 2, 3, 4, 8, 6, 14, 15, 0):
	_lib.set_tuning('ablate', ab)
	for _ in range(10): fwd()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(50): fwd()
	e1.record(); e1.synchronize()
	print(f'ablate={ab:2d}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per forward (whole call: repack + latent bias + chain + registration)', flush=True)
_lib.set_tuning('ablate', 0)
