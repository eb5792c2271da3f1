"""Render C3 (16 feet x 4 views @256^2, silhouette + Phong) forward+backward N times: target for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from find_amd import functional_render as FR, synthetic
from find_amd.cameras import look_at_view_transform
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
want_image = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
synthetic.MESH_KIND = os.environ.get('FIND_MESH_KIND', synthetic.MESH_KIND)   # 'uniform': pole-free triangulation
v, f = synthetic.template(6890)
g = torch.Generator().manual_seed(0)
verts = (v[None] * (1 + 0.1 * torch.rand(16, 1, 3, generator=g))).cuda()
cols = torch.rand(16, v.shape[0], 3, generator=g).cuda()
rng = np.random.RandomState(7)
R, T = look_at_view_transform(dist=np.full(4, 0.3), elev=rng.uniform(-90, 90, 4), azim=rng.uniform(-90, 90, 4), up=((1, 0, 0),))
R, T, fc = R.cuda(), T.cuda(), f.cuda()
params = FR.make_params(size)
if len(sys.argv) > 3:
	from find_amd import _lib
	_lib.set_tuning('raster_ablate', int(sys.argv[3]))
gt = torch.rand(16, 4, size, size, generator=g).cuda()
for _ in range(6):
	vg = verts.detach().requires_grad_(True)
	cg = cols.detach().requires_grad_(True) if want_image else None
	mask, image, _, _ = FR.render(vg, cg, fc, R, T, params, want_image=want_image)
	loss = ((mask - gt) ** 2).mean()
	if want_image:
		loss = loss + (image ** 2).mean()
	loss.backward()
torch.cuda.synchronize()
