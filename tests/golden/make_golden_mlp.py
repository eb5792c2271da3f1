#!/usr/bin/env python3
"""Generate golden vectors G1-G5 by importing the *reference* MLP path.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
It imports /root/reference/src/model/model.py with sys.modules stubs for the third-party
packages that are absent here (pytorch3d, trimesh, cv2, SUPR submodule ...).  Only pure-torch
reference code is executed for real:

  * src/utils/fourier_feature_transform.py:17-55   (FourierFeatureTransform)
  * src/model/model.py:206-391                      (NeuralDisplacementField.__init__)
  * src/model/model.py:393-453                      (NeuralDisplacementField.forward)
  * src/model/model.py:92-152                       (LatentVector)

Outputs (data only: inputs + expected outputs, no reference source):
  tests/golden/mlp_main.npz      G1 _B, G2 seeded state_dict, G3 forward cases, G4 autograd grads
  tests/golden/mlp_variants.npz  G3 flag / latent-size variants (weights reproduced from the seed)
  tests/golden/latent_vector.npz G5 LatentVector.__getitem__ behaviours

Usage:  python tests/golden/make_golden_mlp.py
"""
import os
import sys
from unittest.mock import MagicMock

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
GRAD_STRIDE = 17


def import_reference():
	sys.path.insert(0, REF)
	sys.argv = ['x']  # Opts() parses argv at import-default time (src/train/opts.py:170)
	stubs = ['pytorch3d', 'pytorch3d.structures', 'pytorch3d.transforms', 'pytorch3d.renderer', 'pytorch3d.io',
			 'pytorch3d.ops', 'pytorch3d.loss', 'pytorch3d.loss.chamfer', 'pytorch3d.ops.sample_points_from_meshes',
			 'pytorch3d.renderer.mesh', 'pytorch3d.renderer.mesh.rasterizer',
			 'pytorch3d.renderer.mesh.rasterize_meshes', 'pytorch3d.renderer.mesh.shader',
			 'pytorch3d.structures.utils', 'trimesh', 'cv2', 'ffmpeg', 'torchvision', 'torchvision.models',
			 'torchvision.models.vgg', 'torchvision.transforms', 'src.model.SUPR', 'src.model.SUPR.supr',
			 'src.model.SUPR.supr.pytorch', 'src.model.SUPR.supr.pytorch.supr', 'imageio', 'sklearn',
			 'sklearn.cluster', 'sklearn.decomposition', 'matplotlib', 'matplotlib.pyplot', 'matplotlib.colors',
			 'matplotlib.cm']
	for m in stubs:
		if m in sys.modules:
			continue
		try:
			__import__(m)
		except Exception:
			sys.modules[m] = MagicMock()
	from src.model.model import NeuralDisplacementField, LatentVector
	return NeuralDisplacementField, LatentVector


def synth_positions(gen, B, V):
	"""Template-like coordinates (ellipsoid bounding box)."""
	import torch
	ext = torch.tensor([0.12, 0.045, 0.04])
	return (torch.rand(B, V, 3, generator=gen) * 2 - 1) * ext


def main():
	import numpy as np
	import torch
	torch.set_num_threads(4)
	NDF, LatentVector = import_reference()

	def build(**kw):
		m = NDF(template_mesh_loc=None, device='cpu', **kw)
		# G2: reference zero-inits the last disp layer (model.py:516-518); re-init so the head carries signal
		g = torch.Generator().manual_seed(1234)
		with torch.no_grad():
			m.mlp_disp[-1].weight.copy_(torch.randn(m.mlp_disp[-1].weight.shape, generator=g) * 0.01)
			m.mlp_disp[-1].bias.copy_(torch.randn(m.mlp_disp[-1].bias.shape, generator=g) * 0.01)
		return m

	# ------------------------------------------------------------------ main model (FIND's settings, train.py:148-157)
	main_kw = dict(use_shapevec=True, use_texvec=True, use_posevec=True, train_size=4, val_size=2,
				   shapevec_size=100, texvec_size=100, posevec_size=100)
	model = build(**main_kw)
	out = {}
	out['B'] = model.encoder[0]._B.numpy().copy()  # G1
	for k, v in model.state_dict().items():  # G2
		out['sd/' + k] = v.detach().numpy().copy()

	gen = torch.Generator().manual_seed(0)

	def lat(n, L):
		return torch.randn(n, L, generator=gen) * 0.1

	cases = {
		# name: (B_pos, V, B_lat)   B_pos==1 & B_lat>1 exercises the batch-1 broadcast (model.py:404-406)
		'a': (2, 1000, 2),
		'b': (1, 1000, 16),
		'c': (1, 1, 1),
		'd': (3, 37, 3),
		'e': (16, 1, 16),
		'f': (1, 1000, 1),   # SURVEY G3 / BASELINE configs[0]: a single foot on a 1k-vertex template (appended last: the seeded stream of a-e is unchanged)
	}
	for name, (Bp, V, Bl) in cases.items():
		pos = synth_positions(gen, Bp, V)
		sv, tv, pv = lat(Bl, 100), lat(Bl, 100), lat(Bl, 100)
		with torch.no_grad():
			res = model(pos, shapevec=sv, texvec=tv, posevec=pv)
		for k, v in dict(pos=pos, shapevec=sv, texvec=tv, posevec=pv, disp=res['disp'], col=res['col']).items():
			out[f'fwd/{name}/{k}'] = v.numpy().copy()

	# G4: autograd gradients of sum(disp^2)+sum(col^2) wrt every parameter and latent, cases a (general) and b (broadcast)
	for name in ['a', 'b', 'd', 'f']:
		pos = torch.from_numpy(out[f'fwd/{name}/pos'])
		lats = {k: torch.from_numpy(out[f'fwd/{name}/{k}']).clone().requires_grad_(True)
				for k in ['shapevec', 'texvec', 'posevec']}
		model.zero_grad()
		res = model(pos, **lats)
		loss = (res['disp'] ** 2).sum() + (res['col'] ** 2).sum()
		loss.backward()
		out[f'grad/{name}/loss'] = np.float64(loss.item())
		for k, v in lats.items():
			out[f'grad/{name}/{k}'] = v.grad.numpy().copy()
		for k, p in model.named_parameters():
			if p.grad is not None:
				g = p.grad.numpy().copy()
				# full gradients for case a; for b/d keep small tensors whole and a strided subsample (every
				# GRAD_STRIDE-th element of the flattened tensor) of the big weight matrices to keep the fixture small
				if name != 'a' and g.size > 4096:
					g = g.reshape(-1)[::GRAD_STRIDE].copy()
				out[f'grad/{name}/sd/{k}'] = g
	np.savez(os.path.join(HERE, 'mlp_main.npz'), **out)
	print('mlp_main.npz:', len(out), 'arrays')

	# ------------------------------------------------------------------ variants (weights reproducible from the seed)
	var = {}
	variants = {
		'ttf': dict(use_shapevec=True, use_texvec=True, use_posevec=False, shapevec_size=100, texvec_size=100, posevec_size=100),
		'fff': dict(use_shapevec=False, use_texvec=False, use_posevec=False, shapevec_size=100, texvec_size=100, posevec_size=100),
		'sizes': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=64, texvec_size=32, posevec_size=64),
		'depth2': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=100, texvec_size=100, posevec_size=100,
					   depth=2, dispdepth=2, coldepth=1, sigma=5),
		'avgcol': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=100, texvec_size=100, posevec_size=100,
					   use_avg_colour=True),
	}
	for name, kw in variants.items():
		m = build(train_size=3, val_size=1, **kw)
		if kw.get('use_avg_colour'):
			with torch.no_grad():
				m.avg_col.copy_(torch.tensor([0.1, -0.2, 0.05]))
		g = torch.Generator().manual_seed(11)
		pos = synth_positions(g, 2, 77)
		kws = {}
		if kw['use_shapevec']:
			kws['shapevec'] = torch.randn(2, kw['shapevec_size'], generator=g) * 0.1
		if kw['use_texvec']:
			kws['texvec'] = torch.randn(2, kw['texvec_size'], generator=g) * 0.1
		if kw['use_posevec']:
			kws['posevec'] = torch.randn(2, kw['shapevec_size'], generator=g) * 0.1  # table sized by shapevec_size (model.py:320)
		with torch.no_grad():
			res = m(pos, **kws)
		var[f'{name}/pos'] = pos.numpy()
		for k, v in kws.items():
			var[f'{name}/{k}'] = v.numpy()
		var[f'{name}/disp'] = res['disp'].numpy()
		var[f'{name}/col'] = res['col'].numpy()
		# checksums of the seeded weights so the test can prove it rebuilt the same network
		var[f'{name}/wsum'] = np.array([float(p.double().sum()) for p in m.parameters()])
		var[f'{name}/shapes'] = np.array([str({k: tuple(v.shape) for k, v in m.state_dict().items()})])
	np.savez(os.path.join(HERE, 'mlp_variants.npz'), **var)
	print('mlp_variants.npz:', len(var), 'arrays')

	# ------------------------------------------------------------------ G5 LatentVector
	lv = {}
	labels = ['0003-A', '0005-A', '0005-B', '0007-A']
	L = LatentVector(None, vec_size=5, name='shapevec_train', device='cpu', key='shape', labels=labels)
	with torch.no_grad():
		L.data.copy_(torch.arange(20, dtype=torch.float32).reshape(4, 5))
	lv['labels'] = np.array(labels)
	lv['data'] = L.data.detach().numpy().copy()
	lv['int_2'] = L[2].detach().numpy()
	lv['tensor_3_0'] = L[torch.tensor([3, 0])].detach().numpy()
	lv['str_0005-B'] = L['0005-B'].detach().numpy()
	lv['list_int_1_1_2'] = L[[1, 1, 2]].detach().numpy()
	lv['list_str'] = L[['0007-A', '0003-A']].detach().numpy()
	R = LatentVector(3, vec_size=9, name='reg_train', device='cpu', key='reg', init_values=np.array([0] * 6 + [1] * 3))
	lv['reg_init'] = R.data.detach().numpy().copy()
	lv['len_unlabelled'] = np.array([len(R)])
	np.savez(os.path.join(HERE, 'latent_vector.npz'), **lv)
	print('latent_vector.npz:', len(lv), 'arrays')


if __name__ == '__main__':
	main()
