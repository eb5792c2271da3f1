"""Diagnosis: where does the Fourier layer's weight gradient of the texture pass differ from the reference (tests/golden/texture_loss.npz)?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import find_amd.losses as L
from test_gpu_pins import _reference_model, GOLD
from oracle import mlp_ref

z = np.load(os.path.join(GOLD, 'texture_loss.npz'))
m = _reference_model()
pts, cols = torch.from_numpy(z['points']).cuda(), torch.from_numpy(z['colours']).cuda()
lat = {k: torch.from_numpy(z[k]).cuda().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec')}
L.sample_points_from_meshes = lambda *a, **kw: (pts, cols)
loss = L.TextureLossGTSpace()(m, dict(mesh=None), shapevec=lat['shapevec'], texvec=lat['texvec'], posevec=lat['posevec'])
loss.backward()
g = m.base[0].weight.grad.cpu().double()
# float64 oracle
sd = {k: v.detach().cpu().double().requires_grad_(k.split('.')[0] in ('base', 'mlp_col')) for k, v in m.state_dict().items() if v.is_floating_point()}
B = m.encoder[0]._B.double()
l64 = {k: v.detach().cpu().double() for k, v in lat.items()}
col = mlp_ref.mlp_forward(sd, B, pts.cpu().double(), l64['shapevec'], l64['texvec'], l64['posevec'])['col']
c64 = cols.cpu().double()
mask = (c64 < 1).any(dim=-1, keepdim=True).expand(-1, -1, 3)
((torch.nn.functional.mse_loss(col, c64, reduction='none') * mask).mean()).backward()
r = sd['base.0.weight'].grad
e = (g - r).abs()
s = r.abs().max().item()
print('max err / max', e.max().item() / s, 'at', np.unravel_index(int(e.argmax()), e.shape))
for name, sl in (('sin', slice(0, 256)), ('cos', slice(256, 512)), ('xyz', slice(512, 515))):
	print(name, 'max err/scale %.2e' % (e[:, sl].max().item() / s), ' max |ref| in block / scale %.2e' % (r[:, sl].abs().max().item() / s))
col_err = e.max(dim=0).values / s
print('worst columns', torch.argsort(col_err, descending=True)[:10].tolist(), (torch.sort(col_err, descending=True).values[:10]).tolist())
row_err = e.max(dim=1).values / s
print('worst rows', torch.argsort(row_err, descending=True)[:10].tolist())
Bn = m.encoder[0]._B
print('|B| of worst columns', [float(Bn[:, c % 256].norm()) for c in torch.argsort(col_err, descending=True)[:10].tolist() if c < 512])
for k in ['base.2.weight', 'mlp_col.0.weight', 'base.0.bias']:
	gg, rr = dict(m.named_parameters())[k].grad.cpu().double(), sd[k].grad
	print(k, 'err/scale %.2e' % ((gg - rr).abs().max().item() / rr.abs().max().item()))
