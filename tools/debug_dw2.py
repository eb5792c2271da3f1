import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import _lib, synthetic
L = _lib.lib()
g = dict(np.load('tests/golden/mlp_main.npz'))
from find_amd.model import NeuralDisplacementField
m = NeuralDisplacementField(template_mesh_loc=None, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=4, val_size=2, shapevec_size=100, texvec_size=100, posevec_size=100)
m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd/')})
m = m.cuda()
for shape in [(2, 1000), (2, 16), (2, 32), (2, 48), (1, 64), (2, 17)]:
	N, V = shape
	gen = torch.Generator().manual_seed(1)
	pos = (torch.rand(N, V, 3, generator=gen) * 0.1).cuda()
	lat = [(torch.randn(N, 100, generator=gen) * 0.1).cuda() for _ in range(3)]
	res = {}
	for mode in (0, 1):
		_lib.set_tuning('dw2', mode)
		m.zero_grad()
		out = m(pos, shapevec=lat[0], texvec=lat[1], posevec=lat[2])
		((out['disp'] ** 2).sum() + (out['col'] ** 2).sum()).backward()
		res[mode] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
	bad = [(k, (res[0][k] - res[1][k]).abs().max().item(), res[0][k].abs().max().item()) for k in res[0] if (res[0][k] - res[1][k]).abs().max().item() > 1e-5 * max(1, res[0][k].abs().max().item())]
	print(shape, 'mismatches:', [(k, f'{e:.3g}', f'{s:.3g}') for k, e, s in bad][:6])
