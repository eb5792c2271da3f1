"""Pin the MLP oracle (oracle/mlp_ref.py) against golden vectors captured from the reference import
(tests/golden/make_golden_mlp.py): rows a1-a4 of SURVEY.md §8.  CPU only."""
import numpy as np
import torch

from oracle import mlp_ref

TOL = 2e-6  # same torch build, same op sequence: expect ~1e-7; the north_star gate is 1e-4


def _sd(g):
	return {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd/')}


def test_fourier_B_matches_reference(golden_main):
	B = mlp_ref.fourier_B(3, 256, 10.0)
	assert B.shape == (3, 256)
	np.testing.assert_array_equal(B.numpy(), golden_main['B'])
	# headline value quoted in SURVEY §8c
	np.testing.assert_allclose(B[:, 0].numpy(), [-12.5601, 5.1899, -15.2560], atol=1e-4)
	# rows sorted by norm (fourier_feature_transform.py:25)
	n = B.norm(dim=1)
	assert bool((n[1:] >= n[:-1]).all())


def test_forward_cases(golden_main):
	sd = _sd(golden_main)
	B = torch.from_numpy(golden_main['B'])
	for name in 'abcdef':
		g = {k: torch.from_numpy(golden_main[f'fwd/{name}/{k}']) for k in ['pos', 'shapevec', 'texvec', 'posevec', 'disp', 'col']}
		with torch.no_grad():
			res = mlp_ref.mlp_forward(sd, B, g['pos'], g['shapevec'], g['texvec'], g['posevec'])
		assert res['disp'].shape == g['disp'].shape
		assert (res['disp'] - g['disp']).abs().max() < TOL, name
		assert (res['col'] - g['col']).abs().max() < TOL, name
		assert g['disp'].abs().max() > 1e-4  # fixture is not the trivial zero-init head


def test_backward_case_a(golden_main):
	sd = {k: v.clone().requires_grad_(v.dtype == torch.float32 and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in _sd(golden_main).items()}
	B = torch.from_numpy(golden_main['B'])
	lat = {k: torch.from_numpy(golden_main[f'fwd/a/{k}']).clone().requires_grad_(True) for k in ['shapevec', 'texvec', 'posevec']}
	res = mlp_ref.mlp_forward(sd, B, torch.from_numpy(golden_main['fwd/a/pos']), **lat)
	loss = (res['disp'] ** 2).sum() + (res['col'] ** 2).sum()
	loss.backward()
	assert abs(loss.item() - float(golden_main['grad/a/loss'])) < 1e-3 * abs(float(golden_main['grad/a/loss']))
	for k in lat:
		np.testing.assert_allclose(lat[k].grad.numpy(), golden_main[f'grad/a/{k}'], atol=1e-4, rtol=1e-4)
	for k, v in sd.items():
		if v.requires_grad:
			ref = golden_main[f'grad/a/sd/{k}']
			scale = max(1.0, float(np.abs(ref).max()))
			assert np.abs(v.grad.numpy() - ref).max() < 1e-4 * scale, k


def test_variants(golden_variants):
	"""Flag / latent-size / depth variants: weights are rebuilt from the reference's seeding recipe
	(FFT reseeds the global RNG to 1, then nn.Linear inits in construction order)."""
	from find_amd.model import NeuralDisplacementField
	kws = {
		'ttf': dict(use_shapevec=True, use_texvec=True, use_posevec=False, shapevec_size=100, texvec_size=100, posevec_size=100),
		'fff': dict(use_shapevec=False, use_texvec=False, use_posevec=False, shapevec_size=100, texvec_size=100, posevec_size=100),
		'sizes': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=64, texvec_size=32, posevec_size=64),
		'depth2': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=100, texvec_size=100, posevec_size=100,
					   depth=2, dispdepth=2, coldepth=1, sigma=5),
		'avgcol': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=100, texvec_size=100, posevec_size=100,
					   use_avg_colour=True),
	}
	for name, kw in kws.items():
		m = NeuralDisplacementField(template_mesh_loc=None, device='cpu', train_size=3, val_size=1, **kw)
		g = torch.Generator().manual_seed(1234)
		with torch.no_grad():
			m.mlp_disp[-1].weight.copy_(torch.randn(m.mlp_disp[-1].weight.shape, generator=g) * 0.01)
			m.mlp_disp[-1].bias.copy_(torch.randn(m.mlp_disp[-1].bias.shape, generator=g) * 0.01)
			if kw.get('use_avg_colour'):
				m.avg_col.copy_(torch.tensor([0.1, -0.2, 0.05]))
		assert str({k: tuple(v.shape) for k, v in m.state_dict().items()}) == str(golden_variants[f'{name}/shapes'][0]), name
		wsum = np.array([float(p.detach().double().sum()) for p in m.parameters()])
		np.testing.assert_allclose(wsum, golden_variants[f'{name}/wsum'], rtol=0, atol=1e-9, err_msg=name)
		lat = {k: torch.from_numpy(golden_variants[f'{name}/{k}']) for k in ['shapevec', 'texvec', 'posevec'] if f'{name}/{k}' in golden_variants}
		with torch.no_grad():
			res = mlp_ref.mlp_forward(m.state_dict(), m.encoder[0]._B, torch.from_numpy(golden_variants[f'{name}/pos']),
									  use_avg_colour=kw.get('use_avg_colour', False), **lat)
		assert (res['disp'] - torch.from_numpy(golden_variants[f'{name}/disp'])).abs().max() < TOL, name
		assert (res['col'] - torch.from_numpy(golden_variants[f'{name}/col'])).abs().max() < TOL, name


def test_registration_known_answers():
	"""a5 is [P3D-recall]: anchor on scipy's intrinsic XYZ convention and the row-vector composition."""
	from scipy.spatial.transform import Rotation
	e = torch.tensor([[0.3, -0.2, 0.5], [1.0, 0.1, -0.7]])
	R = mlp_ref.euler_angles_to_matrix_xyz(e)
	Rs = Rotation.from_euler('XYZ', e.numpy()).as_matrix()  # intrinsic XYZ == Rx@Ry@Rz
	np.testing.assert_allclose(R.numpy(), Rs, atol=1e-6)
	# identity registration (model.py:343-344) leaves v + disp unchanged
	v = torch.randn(2, 5, 3)
	d = torch.randn(2, 5, 3) * 0.01
	reg = torch.tensor([[0, 0, 0, 0, 0, 0, 1, 1, 1.]]).repeat(2, 1)
	np.testing.assert_allclose(mlp_ref.registration(v, d, reg).numpy(), (v + d).numpy(), atol=1e-7)
	# row-vector right-multiplication: +90deg about X maps (0,1,0) -> (0,1,0)@Rx = (0,0,-1)
	reg = torch.tensor([[0, 0, 0, np.pi / 2, 0, 0, 1, 1, 1.]], dtype=torch.float32)
	p = mlp_ref.registration(torch.tensor([[[0., 1., 0.]]]), torch.zeros(1, 1, 3), reg)
	np.testing.assert_allclose(p.numpy(), [[[0, 0, -1]]], atol=1e-6)
	# scale is applied before rotation, translation last
	reg = torch.tensor([[1., 2., 3., 0, 0, np.pi / 2, 2., 1., 1.]], dtype=torch.float32)
	p = mlp_ref.registration(torch.tensor([[[1., 0., 0.]]]), torch.zeros(1, 1, 3), reg)
	# (1,0,0)*S=(2,0,0); @Rz(90) with Rz=[[0,-1,0],[1,0,0],[0,0,1]] -> (0,-2,0); + t
	np.testing.assert_allclose(p.numpy(), [[[1., 0., 3.]]], atol=1e-6)
