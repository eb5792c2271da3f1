"""Where does the Phong-image vertex gradient of the 6890-vertex template @256^2 differ from the oracle (tests/test_gpu_fullsize.py (b))?
GPU vs the oracle's autograd in fp32 and in fp64: is the difference the kernel's or fp32 rounding on both sides?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import render_ref
from find_amd import functional_render as FR
from test_gpu_fullsize import _template_scene

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
verts, f, cols, R, T = _template_scene(1, 1, seed_verts=5, seed_views=3)
params = FR.make_params(size)
vg = verts.clone().cuda().requires_grad_(True); cg = cols.clone().cuda().requires_grad_(True)
_, image, p2f, _ = FR.render(vg, cg, f.cuda(), R.cuda(), T.cuda(), params, want_mask=False, want_frags=True)
wi = torch.rand(image.shape, generator=torch.Generator().manual_seed(5))
gv, gc = torch.autograd.grad((image * wi.cuda()).sum(), (vg, cg))
rp = render_ref.default_params(size)
sel = p2f.cpu().long().reshape(-1, size, size, 1)
out = {}
for dt in (torch.float32, torch.float64):
	vr = verts.to(dt).requires_grad_(True); cr = cols.to(dt).requires_grad_(True)
	ri = render_ref.torch_phong_image(rp, vr, cr, f, R.to(dt), T.to(dt), sel, 1, compact=True)
	rv, rc = torch.autograd.grad((ri * wi.to(dt)).sum(), (vr, cr))
	out[dt] = (ri.detach(), rv, rc)
r32, r64 = out[torch.float32], out[torch.float64]
s = r64[1].abs().max().item()
print('scale', s)
print('image: gpu-f32 %.2e gpu-f64 %.2e f32-f64 %.2e' % ((image.detach().cpu() - r32[0]).abs().max(), (image.detach().cpu().double() - r64[0]).abs().max(), (r32[0].double() - r64[0]).abs().max()))
e32 = (gv.cpu() - r32[1]).abs(); e64 = (gv.cpu().double() - r64[1]).abs(); eo = (r32[1].double() - r64[1]).abs()
print('vertex grad rel max: gpu-f32 %.2e  gpu-f64 %.2e  f32oracle-f64oracle %.2e' % (e32.max() / s, e64.max() / s, eo.max() / s))
top = torch.topk(e64.max(dim=-1).values.reshape(-1), 8)
for val, idx in zip(top.values, top.indices):
	i = idx.item()
	print(i, 'err64 %.2e' % (val / s), 'gpu', gv.cpu()[0, i].tolist(), 'f64', r64[1][0, i].tolist(), 'f32', r32[1][0, i].tolist())
sc = r64[2].abs().max().item()
print('colour grad rel max: gpu-f64 %.2e f32-f64 %.2e' % ((gc.cpu().double() - r64[2]).abs().max() / sc, (r32[2].double() - r64[2]).abs().max() / sc))
