"""The HIP path against fixtures produced by RUNNING reference code (tests/golden/make_golden_pins.py; the CPU half -- the oracle against
the same fixtures -- is tests/test_oracle_pins.py):

  * blend_scene.npz   the reference's softmax_blend (src/model/renderer.py:23-72) applied to the fragments of a small scene -> the image
                      and the soft silhouette find_render_fwd must produce for that scene;
  * texture_loss.npz  the reference's TextureLossGTSpace.forward (src/model/losses.py:22-57) -> find_amd.losses.TextureLossGTSpace with the
                      same recorded sampler output: loss, gradients, and WHICH parameters get none;
  * ref_checkpoint.*  a checkpoint written by the reference's save_model (model.py:156-161), loaded by find_amd on the GPU -> identical forward."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


def test_render_of_a_scene_equals_the_reference_blend_of_its_fragments():
	from find_amd import functional_render as FR
	z = np.load(os.path.join(GOLD, 'blend_scene.npz'))
	size = int(z['image_size'])
	params = FR.make_params(size)
	params.ambient, params.diffuse, params.specular = 1.0, 0.0, 0.0   # flat shading: the fragment colour is the face colour
	verts, cols = torch.from_numpy(z['verts']).cuda(), torch.from_numpy(z['vert_colours']).cuda()
	faces, R, T = torch.from_numpy(z['faces']).cuda(), torch.from_numpy(z['R']).cuda(), torch.from_numpy(z['T']).cuda()
	prev, FR.FLAG_POLICY = FR.FLAG_POLICY, 'sync'
	try:
		mask, image, p2f, zbuf = FR.render(verts, cols, faces, R, T, params, want_mask=True, want_image=True, want_frags=True)
	finally:
		FR.FLAG_POLICY = prev
	N, M = verts.shape[0], R.shape[0]
	want_img = z['image'].reshape(N, M, size, size, 3)
	want_alpha = z['alpha'].reshape(N, M, size, size)
	same = p2f.cpu().numpy() == z['pix_to_face'].reshape(N, M, size, size)
	assert same.mean() > 0.998, same.mean()   # (a pixel centre exactly on an edge may go to either face)
	assert (z['pix_to_face'] >= 0).mean() > 0.1
	assert np.abs(image.cpu().numpy() - want_img)[same].max() < TOL
	assert np.abs((1.0 - mask.cpu().numpy()) - want_alpha).max() < TOL
	assert ((want_alpha > 0.01) & (want_alpha < 0.99)).sum() > 50   # the soft rim is in the picture, not only 0 / 1 pixels


def _reference_model(device='cuda'):
	from find_amd.model import NeuralDisplacementField
	z = np.load(os.path.join(GOLD, 'mlp_main.npz'))
	m = NeuralDisplacementField(template_mesh_loc=None, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=4, val_size=2,
								shapevec_size=100, texvec_size=100, posevec_size=100)
	m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')})
	assert torch.equal(m.encoder[0]._B, torch.from_numpy(z['B']))
	return m.to(device)


def test_texture_loss_equals_the_reference_loss_class():
	import find_amd.losses as L
	z = np.load(os.path.join(GOLD, 'texture_loss.npz'))
	m = _reference_model()
	pts, cols = torch.from_numpy(z['points']).cuda(), torch.from_numpy(z['colours']).cuda()
	lat = {k: torch.from_numpy(z[k]).cuda().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec')}
	calls = []

	def recorded_sampler(meshes, num_samples=10000, return_textures=False, **kw):
		calls.append((meshes, num_samples, return_textures))
		return pts, cols

	orig, L.sample_points_from_meshes = L.sample_points_from_meshes, recorded_sampler
	try:
		loss = L.TextureLossGTSpace()(m, dict(mesh='the GT meshes'), shapevec=lat['shapevec'], texvec=lat['texvec'], posevec=lat['posevec'])
	finally:
		L.sample_points_from_meshes = orig
	assert calls == [('the GT meshes', 1000, True)]
	want = float(z['loss'])
	assert abs(loss.item() - want) < TOL * max(1.0, want), (loss.item(), want)
	loss.backward()
	# the reference's autograd gives the displacement head, and the latents only that head reads, NO gradient (not a zero one)
	no_grad = set(z['no_grad'].tolist())
	for k, p in m.named_parameters():
		if k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'):
			assert (p.grad is None) == (k in no_grad), k
	assert lat['shapevec'].grad is None and lat['posevec'].grad is None and z['grad/shapevec'].size == 0
	worst = 0.0
	g, w = lat['texvec'].grad.cpu().numpy(), z['grad/texvec']
	worst = max(worst, np.abs(g - w).max() / np.abs(w).max())
	for k in z.files:
		if not k.startswith('grad/sd/'):
			continue
		g = dict(m.named_parameters())[k[8:]].grad.cpu().numpy()
		g = g.reshape(-1)[::17] if g.size > 4096 else g
		scale = float(z['gradmax/sd/' + k[8:]])   # the whole tensor's largest entry (the fixture keeps every 17th element of the big ones)
		worst = max(worst, np.abs(g - z[k]).max() / scale)
		assert np.abs(g - z[k]).max() < TOL * scale, k
	assert worst < TOL
	print(f'texture loss vs the reference class: worst gradient error {worst:.2e} of the tensor maximum')


def test_reference_checkpoint_on_the_gpu():
	from find_amd.model import NeuralDisplacementField
	from find_amd.opts import Opts
	z = np.load(os.path.join(GOLD, 'ref_checkpoint.npz'))
	m = NeuralDisplacementField.load(os.path.join(GOLD, 'ref_checkpoint.pth'), device='cuda', opts=Opts()).to('cuda')   # (train.py:158: model.to(device) after construction)
	assert m.base[0].weight.is_cuda and m.template_verts.is_cuda
	with torch.no_grad():
		res = m(torch.from_numpy(z['pos']).cuda(), shapevec=m.shapevec[['0005', '0003']], texvec=m.texvec[['0003', '0003']], posevec=m.posevec[['0005-B', '0003-A']])
		tmpl = m(m.template_verts.data, shapevec=m.shapevec_val[['0011', '0011']], texvec=m.texvec_val[['0011', '0011']],
				 posevec=m.posevec_val[['0011-A', '0011-B']])
	assert np.abs(res['disp'].cpu().numpy() - z['disp']).max() < 1e-5 and np.abs(res['col'].cpu().numpy() - z['col']).max() < 1e-5
	assert np.abs(tmpl['disp'].cpu().numpy() - z['template_disp']).max() < 1e-5 and np.abs(tmpl['col'].cpu().numpy() - z['template_col']).max() < 1e-5
	# save_model writes what the reference's load reads back: same keys, same values
	import tempfile
	with tempfile.TemporaryDirectory() as d:
		m.save_model(out_dir=d, fname='again')
		a = torch.load(os.path.join(d, 'again.pth'), map_location='cpu', weights_only=False)
	b = torch.load(os.path.join(GOLD, 'ref_checkpoint.pth'), map_location='cpu', weights_only=False)
	assert list(a['state_dict']) == list(b['state_dict']) and a['params'] == b['params']
	for k in a['state_dict']:
		assert torch.equal(a['state_dict'][k], b['state_dict'][k]), k


def test_model_with_loss_equals_the_reference_model_with_loss():
	"""find_amd.ModelWithLoss.forward on the GPU against what the REFERENCE's ModelWithLoss.forward returned for the same weights, latent tables,
	scans, sampler draws and flags (tests/golden/make_golden_composition.py ran src/model/model.py:1001-1163 for real, its PyTorch3D calls backed
	by the oracle): the loss dict -- keys in the reference's order --, the total, gradients of every MLP weight and latent table, for the
	network-stage flags (Opts.net_train_kwargs), the registration stage with a GT cut-off, a validation step with the z cut-off on both
	clouds, a scan whose 3-D supervision is withheld (total is the python int 0) and the texture term alone."""
	import sys
	sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
	from test_gpu_train3d import FixedDraws
	from find_amd.model_with_loss import ModelWithLoss
	from find_amd.opts import Opts
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	z = np.load(os.path.join(GOLD, 'composition.npz'))
	dev = torch.device('cuda')
	lab = {k[len('labels/'):]: [str(s) for s in z[k]] for k in z.files if k.startswith('labels/')}
	opts = Opts(chamf_loss=True, smooth_loss=True, texture_loss=True, use_pose_code=True, use_latent_labels=True)
	mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=3, val_size=3, shapevec_size=100,
						texvec_size=100, posevec_size=100, template_mesh_loc=None, latent_labels=lab)
	m = mwl.model
	m.set_template(torch.from_numpy(z['sd/template_verts'])[0], torch.from_numpy(z['sd/template_faces'])[0])
	m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}, strict=True)
	assert torch.equal(m.encoder[0]._B, torch.from_numpy(z['B']))
	mwl = mwl.to(dev)
	m = mwl.model
	gv, gf, gc = (torch.from_numpy(z[f'gt/{k}']).to(dev) for k in ('verts', 'faces', 'colours'))
	feet, names = [str(s) for s in z['batch/feet']], [str(s) for s in z['batch/names']]
	feet_val, names_val = [str(s) for s in z['batch/feet_val']], [str(s) for s in z['batch/names_val']]
	params = dict(mwl.named_parameters())
	for name in z['cases']:
		flags = {}
		for f in z[f'case/{name}/flags']:
			k, v = str(f).split('=')
			flags[k] = {'True': True, 'False': False, 'None': None}.get(v, None if v == 'None' else v)
			if k == 'gt_z_cutoff' and v != 'None':
				flags[k] = float(v)
		val = flags.get('is_train', True) is False
		idx = [int(i) for i in z[f'case/{name}/idx']]
		ft, nm = (feet_val, names_val) if val else (feet, names)
		b = dict(mesh=Meshes(gv[idx].contiguous(), gf, TexturesVertex(gc[idx].contiguous())), idx=torch.tensor(idx, device=dev), name=[nm[i] for i in idx],
				 shape=[ft[i] for i in idx], tex=[ft[i] for i in idx], pose=[nm[i] for i in idx], reg=[nm[i] for i in idx])
		b.update(sample_latent_vectors(b, m.latent_vectors_val if val else m.latent_vectors_train))
		n_draws = int(z[f'case/{name}/n_draws'])
		dr = [(torch.from_numpy(z[f'case/{name}/draw/{i}/face_idx']).to(dev), torch.from_numpy(z[f'case/{name}/draw/{i}/uv']).to(dev)) for i in range(n_draws)]
		# FixedDraws hands out (GT / Chamfer, prediction / Chamfer, GT / texture); the reference drew in the order GT, prediction, texture
		if n_draws == 3:
			draws = dr
		elif n_draws == 2:
			draws = [dr[0], dr[1], None]
		elif n_draws == 1:
			draws = [None, None, dr[0]]
		else:
			draws = [None, None, None]
		for k, v in (dict(s.split('=') for s in z[f'case/{name}/opts'])).items():
			setattr(opts, k, int(v))
		for p in mwl.parameters():
			p.grad = None
		try:
			with FixedDraws(draws):
				loss, losses = mwl(b, 0, opts, **flags)
		finally:
			opts.restrict_3d_n_train = None
		assert list(losses) == [str(s) for s in z[f'case/{name}/loss_keys']], (name, list(losses))
		if f'case/{name}/loss_is_python_zero' in z.files:
			assert not torch.is_tensor(loss) and loss == 0
			continue
		for k, v in losses.items():
			want = float(z[f'case/{name}/losses/{k}'])
			assert abs(v.item() - want) < TOL * max(1.0, abs(want)), (name, k, v.item(), want)
		assert abs(loss.item() - float(z[f'case/{name}/loss'])) < TOL * max(1.0, abs(loss.item()))
		loss.backward()
		# gradients against the reference's code run in float64 on the same draws (the fixture's yardstick run); the bound is the north star's
		# 1e-4 of the tensor's largest entry, or twice the distance of the reference's OWN fp32 gradients from those float64 values where that
		# is larger (it is not, in this fixture: <= 1.5e-5).  The draws are free of nearest-neighbour near-ties (make_golden_composition.py).
		seen, worst = 0, (0.0, '')
		for key in z.files:
			if not key.startswith(f'case/{name}/grad64/'):
				continue
			k = key[len(f'case/{name}/grad64/'):]
			want = z[key]
			if np.abs(want).max() == 0:
				assert params[k].grad is None or params[k].grad.abs().max().item() == 0, (name, k)
				continue
			got = params[k].grad.detach().cpu().numpy().astype(np.float64)
			got = got if got.size <= 4096 else got.reshape(-1)[::17]
			err = np.abs(got - want).max() / max(1e-3, np.abs(want).max())
			assert err < max(TOL, 2.0 * float(z[f'case/{name}/ref_fp32_error/{k}'])), (name, k, err, float(z[f'case/{name}/ref_fp32_error/{k}']))
			worst = max(worst, (float(err), k))
			seen += 1
		assert seen >= 4, (name, seen)
		print(f'composition case {name}: worst gradient error {worst[0]:.1e} of the tensor maximum ({worst[1]}); nearest-neighbour gap of the draws >= {float(z[f"case/{name}/nn_min_relative_gap"]):.1e}')
		have = {k for k, p in params.items() if p.grad is not None and p.grad.abs().max().item() > 0}
		want_keys = {key[len(f'case/{name}/grad/'):] for key in z.files if key.startswith(f'case/{name}/grad/') and np.abs(z[key]).max() > 0}
		assert have == want_keys, (name, have ^ want_keys)
