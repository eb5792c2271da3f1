"""Forward silhouette render of the C3 end-to-end shapes under the rasteriser's ablation bits (find_debug_raster_ablate; _lib.set_tuning("raster_ablate", bits)):
where the time of raster_tile_kernel goes.  Usage: python tools/prof_raster_ablate.py [n_verts] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from find_amd import _lib, functional_render as FR, synthetic
from find_amd.cameras import look_at_view_transform
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 6890
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
v, f = synthetic.template(nv)
g = torch.Generator().manual_seed(0)
verts = (v[None] * (1 + 0.1 * torch.rand(16, 1, 3, generator=g))).cuda()
rng = np.random.RandomState(7)
R, T = look_at_view_transform(dist=np.full(4, 0.3), elev=rng.uniform(-90, 90, 4), azim=rng.uniform(-90, 90, 4), up=((1, 0, 0),))
R, T, fc = R.cuda(), T.cuda(), f.cuda()
params = FR.make_params(size)
L = _lib.lib()
for ab in [int(x) for x in sys.argv[3:]] or [0, 1, 2, 3, 4, 0]:
	_lib.set_tuning('raster_ablate', ab)
	for _ in range(3):
		FR.render(verts, None, fc, R, T, params, want_image=False)
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(10):
		FR.render(verts, None, fc, R, T, params, want_image=False)
	e1.record(); e1.synchronize()
	print(f'V={nv} {size}^2 ablate={ab}: {e0.elapsed_time(e1) / 10:.3f} ms / forward render (16 feet x 4 views)')
