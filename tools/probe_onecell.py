import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from find_amd import _lib, functional as FN
g = torch.Generator().manual_seed(77)
def sphere(n, p, r=0.1, noise=1e-3):
	v = torch.randn(n, p, 3, generator=g)
	return v / v.norm(dim=-1, keepdim=True) * r + noise * torch.randn(n, p, 3, generator=g)
sphere(3, 5000); sphere(3, 5000); torch.randn(2, 4000, 3, generator=g); torch.randn(2, 3000, 3, generator=g)
y = torch.randn(1, 2500, 3, generator=g) * 1e-6 + 0.3
x = torch.randn(1, 2500, 3, generator=g)
def run(mode):
	_lib.set_tuning('raster_ablate', mode)
	xg, yg = x.clone().cuda().requires_grad_(True), y.clone().cuda().requires_grad_(True)
	loss, _ = FN.chamfer_distance(xg, yg)
	loss.backward(); torch.cuda.synchronize()
	_lib.set_tuning('raster_ablate', 0)
	return loss.detach().clone(), xg.grad.clone(), yg.grad.clone()
ref = run(512)
worst = 0
for it in range(60):
	for mode in (1024, 512):
		l, gx, gy = run(mode)
		dx = (gx - ref[1]).abs().max().item() / ref[1].abs().max().item()
		dy = (gy - ref[2]).abs().max().item() / ref[2].abs().max().item()
		nrow = ((gy - ref[2]).abs().amax(-1) > 0).sum().item()
		if dx > 1e-7 or dy > 1e-7:
			print(it, mode, 'loss equal', torch.equal(l, ref[0]), 'dx %.2e dy %.2e rows differing in gy %d' % (dx, dy, nrow), 'max|gy| %.3e' % ref[2].abs().max().item())
print('distinct target rows', len(set(map(tuple, y[0].tolist()))))
