"""Diagnosis: per-pixel silhouette gradient of K-overflow pixels, HIP against autograd through the oracle's K = 100 fragments
(the scene of tests/test_gpu_render.py::test_silhouette_backward_with_k_overflow_vs_oracle_autograd).  Prints the pixels whose
gradient differs most, with the depth gaps around their K-th candidate."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import functional_render as FR, synthetic   # noqa: E402
from oracle import camera_ref, render_ref   # noqa: E402

size = 64
v, f = synthetic.template(6890)
g = torch.Generator().manual_seed(9)
verts = v[None] * (1 + 0.1 * torch.rand(1, 1, 3, generator=g))
rng = np.random.RandomState(4)
R, T = camera_ref.look_at_view_transform(dist=np.full(1, 0.3), elev=rng.uniform(-90, 90, 1), azim=rng.uniform(-90, 90, 1), up=((1, 0, 0),))
R, T = torch.from_numpy(R), torch.from_numpy(T)
rp = render_ref.default_params(size)
vproj = render_ref.project(rp, verts.numpy(), R.numpy(), T.numpy())
p2f, z, _, d = render_ref.rasterize(vproj, f.numpy(), 1, size, size, 104, rp.sil_blur_radius)
full = np.argwhere(p2f[0, :, :, 99] >= 0)
print('overflow pixels', len(full))
params = FR.make_params(size)
vg = verts.clone().cuda().requires_grad_(True)
mask, _, _, _ = FR.render(vg, None, f.cuda(), R.cuda(), T.cuda(), params, want_image=False)
vr = verts.clone().requires_grad_(True)
rm = render_ref.torch_mask(rp, vr, f, R, T, torch.from_numpy(np.ascontiguousarray(p2f[..., :100])).long(), 1)
rows = []
for (yi, xi) in full:
	gg, = torch.autograd.grad(mask[0, 0, yi, xi], vg, retain_graph=True)
	gr, = torch.autograd.grad(rm[0, 0, yi, xi], vr, retain_graph=True)
	s = gr.abs().max().item()
	e = (gg.cpu() - gr).abs().max().item()
	zz = z[0, yi, xi]
	n_c = int((p2f[0, yi, xi] >= 0).sum())
	rows.append((e / max(s, 1e-30), e, s, yi, xi, n_c, zz[98], zz[99], zz[100] if n_c > 100 else -1, float(mask[0, 0, yi, xi]), float(rm[0, 0, yi, xi])))
rows.sort(reverse=True)
print('rel err, abs err, scale, y, x, n(<=104), z98, z99, z100, mask gpu, mask ref')
for r in rows[:25]:
	print('%.2e %.2e %.2e  y=%d x=%d n=%d  z98=%.9g z99=%.9g z100=%.9g  gaps=%.2e %.2e  mask %.7f %.7f' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], (r[7] - r[6]) / r[7], (r[8] - r[7]) / r[7] if r[8] > 0 else -1, r[9], r[10]))
print('median rel err %.2e' % rows[len(rows) // 2][0])
# pixels below K for comparison
low = np.argwhere((p2f[0, :, :, 99] < 0) & (p2f[0, :, :, 0] >= 0))[::7]
w = 0
for (yi, xi) in low:
	gg, = torch.autograd.grad(mask[0, 0, yi, xi], vg, retain_graph=True)
	gr, = torch.autograd.grad(rm[0, 0, yi, xi], vr, retain_graph=True)
	s = gr.abs().max().item()
	if s > 0:
		w = max(w, (gg.cpu() - gr).abs().max().item() / s)
print('worst rel err over %d pixels with fewer than K candidates: %.2e' % (len(low), w))

# ---- one pixel in detail: which vertices differ, and where their faces sit in the oracle's depth order
for (yi, xi) in [(10, 34), (12, 35)]:
	gg, = torch.autograd.grad(mask[0, 0, yi, xi], vg, retain_graph=True)
	gr, = torch.autograd.grad(rm[0, 0, yi, xi], vr, retain_graph=True)
	diff = (gg.cpu() - gr)[0].abs().max(dim=1).values
	top = torch.argsort(diff, descending=True)[:12]
	faces_np = f.numpy()
	sel = p2f[0, yi, xi]          # packed == local (one image)
	zz, dd = z[0, yi, xi], d[0, yi, xi]
	print(f'pixel y={yi} x={xi}: candidates kept by the oracle {int((sel[:100] >= 0).sum())}, listed {int((sel >= 0).sum())}')
	for vtx in top.tolist():
		ranks = [k for k in range(sel.shape[0]) if sel[k] >= 0 and vtx in faces_np[sel[k]]]
		print('  vertex %5d  |gpu - ref| %.3e  gpu %s ref %s  ranks of its faces in the oracle list: %s' % (
			vtx, diff[vtx].item(), np.array2string(gg[0, vtx].cpu().numpy(), precision=4), np.array2string(gr[0, vtx].numpy(), precision=4),
			[(k, float('%.6g' % zz[k]), float('%.3g' % dd[k])) for k in ranks]))
