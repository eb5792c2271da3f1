#!/usr/bin/env python3
"""Golden vectors for three more pieces of the path that the reference itself can still compute in this container.

Runs ONLY in the build container (needs /root/reference); same sys.modules stubs as make_golden_mlp.py.  Executed for real:

  * src/model/renderer.py:23-72       softmax_blend -- the in-repo statement of PyTorch3D's softmax blend: sigmoid face probabilities,
                                      alpha = prod(1 - p), depth-softmax colour weights, background weight delta.  Pure torch.
  * src/model/losses.py:22-57         TextureLossGTSpace.forward on the reference NeuralDisplacementField, with the one PyTorch3D call
                                      in it (sample_points_from_meshes, losses.py:39-41) replaced by a function that returns recorded
                                      points / colours -- the sampler is an input of the loss, the rest runs as written.
  * src/model/model.py:156-161        Model.save_model -> a checkpoint written by the reference, for find_amd's Model.load.

Outputs (data only):
  tests/golden/blend.npz          fragments (pix_to_face / dists / zbuf / colours, K in {1, 4, 100}, empty slots, empty pixels) ->
                                  blended colours, and the alpha line 54 computes on the way (captured at torch.prod)
  tests/golden/blend_scene.npz    a small mesh scene: the ORACLE's fragments (K = 1 sharp, K = 100 blurred) blended by the REFERENCE function ->
                                  the image / silhouette the HIP renderer must produce for that scene (flat shading: ambient 1, no light terms)
  tests/golden/texture_loss.npz   sampled points / colours (a saturated share), latents -> loss, gradients of latents and weights
  tests/golden/ref_checkpoint.pth the reference's own torch.save of {'state_dict', 'params'} (label-addressed tables, trained-looking values)
  tests/golden/ref_checkpoint.npz inputs -> outputs of the reference model that wrote it

Usage:  python tests/golden/make_golden_pins.py"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GRAD_STRIDE = 17


def main():
	import numpy as np
	import torch
	import make_golden_mlp as G
	torch.set_num_threads(4)
	NDF, LatentVector = G.import_reference()
	import src.model.losses as ref_losses
	import src.model.renderer as ref_renderer

	# ------------------------------------------------------------------ softmax_blend
	out = {}
	gen = torch.Generator().manual_seed(21)
	blend_params = types.SimpleNamespace(sigma=1e-4, gamma=1e-4, background_color=(1.0, 1.0, 1.0))   # BlendParams() defaults, renderer.py:136
	for K in (1, 4, 100):
		N, H, W, C = 2, 6, 5, 3
		p2f = torch.randint(0, 50, (N, H, W, K), generator=gen)
		# empty slots fill from the back as in a K-buffer; some pixels are empty altogether
		n_valid = torch.randint(0, K + 1, (N, H, W), generator=gen)
		n_valid[0, 0, 0] = 0
		n_valid[1, 2, 3] = K
		slot = torch.arange(K).view(1, 1, 1, K)
		p2f = torch.where(slot < n_valid[..., None], p2f, torch.full_like(p2f, -1))
		# signed squared NDC distances around the blur scale (sigma 1e-4): inside (< 0) and outside; view depths 0.15 .. 0.45 ascending
		dists = (torch.rand(N, H, W, K, generator=gen) * 2 - 1) * 4e-4
		zbuf = torch.sort(0.15 + 0.3 * torch.rand(N, H, W, K, generator=gen), dim=-1).values
		dists = torch.where(p2f >= 0, dists, torch.full_like(dists, -1.0))
		zbuf = torch.where(p2f >= 0, zbuf, torch.full_like(zbuf, -1.0))
		colors = torch.rand(N, H, W, K, C, generator=gen)
		frags = types.SimpleNamespace(pix_to_face=p2f, dists=dists, zbuf=zbuf)
		seen = []
		real_prod = torch.prod

		def spy(*a, **kw):
			r = real_prod(*a, **kw)
			seen.append(r)
			return r

		torch.prod = spy
		try:
			pix = ref_renderer.softmax_blend(colors, frags, blend_params, znear=0.02, zfar=100)   # FoVPerspectiveCameras(znear=0.02), zfar default
		finally:
			torch.prod = real_prod
		assert len(seen) == 1 and seen[0].shape == (N, H, W)
		for k, v in dict(pix_to_face=p2f.int(), dists=dists, zbuf=zbuf, colors=colors, pixel_colors=pix, alpha=seen[0]).items():
			out[f'K{K}/{k}'] = v.numpy().copy()
	out['sigma'], out['gamma'], out['znear'], out['zfar'] = np.float64(1e-4), np.float64(1e-4), np.float64(0.02), np.float64(100.0)
	out['background'] = np.array([1.0, 1.0, 1.0], np.float32)
	np.savez(os.path.join(HERE, 'blend.npz'), **out)
	print('blend.npz:', len(out), 'arrays')

	# ------------------------------------------------------------------ the same function on the fragments of a rendered scene
	# Fragments from the oracle's naive rasteriser (PyTorch3D's is absent), blended by the reference: pins the blend END TO END for the HIP
	# renderer, which never materialises fragments.  Flat shading (ambient 1, diffuse = specular = 0) and one colour per face (vertices are
	# not shared between faces) make the fragment colour the face colour, so no unpinned shading enters.
	sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
	from find_amd import synthetic
	from oracle import camera_ref, render_ref
	size, n_views = 32, 2
	v0, f0 = synthetic.ellipsoid_mesh(7, 9)
	rng = np.random.RandomState(5)
	meshes = []
	for n in range(2):
		vv = v0.numpy() * (1 + 0.15 * rng.rand(1, 3))
		meshes.append(vv[f0.numpy().reshape(-1)])                     # (3F, 3): every face owns its three vertices
	verts = np.stack(meshes).astype(np.float32)
	F = f0.shape[0]
	faces = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
	face_col = rng.rand(2, F, 3).astype(np.float32)
	vcol = np.repeat(face_col, 3, axis=1)                              # (2, 3F, 3)
	R, T = camera_ref.look_at_view_transform(dist=np.full(n_views, 0.3), elev=rng.uniform(-60, 60, n_views), azim=rng.uniform(-60, 60, n_views), up=((1, 0, 0),))
	rp = render_ref.default_params(size)
	vproj = render_ref.project(rp, verts, R, T)
	sc = dict(verts=verts, faces=faces, vert_colours=vcol, R=R, T=T, image_size=np.int64(size))
	p2f, zb, ba, di = render_ref.rasterize(vproj, faces, n_views, size, size, 1, 0.0)
	cols = np.zeros(p2f.shape + (3,), np.float32)
	img_of = np.arange(p2f.shape[0]) // n_views
	hit = p2f >= 0
	local = np.where(hit, p2f - (np.arange(p2f.shape[0]) * F)[:, None, None, None], 0)
	cols[hit] = face_col[np.broadcast_to(img_of[:, None, None, None], p2f.shape)[hit], local[hit]]
	frags = types.SimpleNamespace(pix_to_face=torch.from_numpy(p2f).long(), dists=torch.from_numpy(di), zbuf=torch.from_numpy(zb))
	sc['image'] = ref_renderer.softmax_blend(torch.from_numpy(cols), frags, blend_params, znear=0.02, zfar=100).numpy()
	sc['pix_to_face'] = p2f[..., 0]
	p2f, zb, ba, di = render_ref.rasterize(vproj, faces, n_views, size, size, 100, rp.sil_blur_radius)
	frags = types.SimpleNamespace(pix_to_face=torch.from_numpy(p2f).long(), dists=torch.from_numpy(di), zbuf=torch.from_numpy(zb))
	seen = []
	torch.prod = lambda *a, **kw: (seen.append(real_prod(*a, **kw)), seen[-1])[1]
	try:
		ref_renderer.softmax_blend(torch.zeros(p2f.shape + (3,)), frags, blend_params, znear=0.02, zfar=100)
	finally:
		torch.prod = real_prod
	sc['alpha'] = seen[0].numpy()
	sc['max_candidates'] = np.int64((p2f >= 0).sum(-1).max())
	np.savez(os.path.join(HERE, 'blend_scene.npz'), **sc)
	print('blend_scene.npz: covered', float(hit.mean()), 'soft', float((sc['alpha'] < 1).mean()), 'max candidates per pixel', int(sc['max_candidates']))

	# ------------------------------------------------------------------ TextureLossGTSpace
	main_kw = dict(use_shapevec=True, use_texvec=True, use_posevec=True, train_size=4, val_size=2, shapevec_size=100, texvec_size=100, posevec_size=100)
	model = NDF(template_mesh_loc=None, device='cpu', **main_kw)
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():
		model.mlp_disp[-1].weight.copy_(torch.randn(model.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		model.mlp_disp[-1].bias.copy_(torch.randn(model.mlp_disp[-1].bias.shape, generator=g) * 0.01)
	gen = torch.Generator().manual_seed(33)
	Nf, S = 3, 1000   # losses.py:27 num_samples
	pts = G.synth_positions(gen, Nf, S)
	cols = torch.rand(Nf, S, 3, generator=gen) * 0.9 + 0.05
	# saturated samples (masked out, losses.py:43): all three channels >= 1 on a quarter of them, exactly 1.0 and above; a sample with ONE
	# channel below 1 stays in
	sat = torch.rand(Nf, S, generator=gen) < 0.25
	cols[sat] = 1.0
	cols[0, :5] = torch.tensor([1.0, 1.0, 1.2])
	cols[1, :5] = torch.tensor([1.0, 0.999, 1.0])
	lats = {k: (torch.randn(Nf, 100, generator=gen) * 0.1).requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec')}
	# ReLU is not differentiable at 0: a pre-activation within rounding of zero makes its gradient mask a coin toss between two correct
	# implementations, and ONE such element is 1e-4 of a 3000-row weight gradient.  The fixture is made tie-free: sample points whose
	# rows have a hidden pre-activation (float64, trunk and colour head) closer to zero than 1e-5 are redrawn.
	m64h = NDF(template_mesh_loc=None, device='cpu', **main_kw)
	m64h.load_state_dict(model.state_dict())
	m64h = m64h.double()
	m64h.encoder[0]._B = m64h.encoder[0]._B.double()
	hidden = [l for seq in (m64h.base, list(m64h.mlp_col)[:-1]) for l in seq if isinstance(l, torch.nn.Linear)]
	for it in range(50):
		seen_pre = []
		hooks = [l.register_forward_hook(lambda mod, inp, out: seen_pre.append(out.detach().abs().amin(dim=-1))) for l in hidden]
		with torch.no_grad():
			m64h(pts.double(), **{k: v.detach().double() for k, v in lats.items()})
		for h in hooks:
			h.remove()
		near = torch.stack(seen_pre).amin(dim=0) < 1e-5      # (Nf, S)
		if not near.any():
			break
		pts[near] = G.synth_positions(gen, 1, int(near.sum()))[0]
	assert not near.any()
	print(f'tie-free sample points after {it} redraw rounds; closest hidden pre-activation to zero: {float(torch.stack(seen_pre).min()):.2e}')
	calls = []

	def recorded_sampler(meshes, num_samples=10000, return_textures=False, **kw):
		calls.append((meshes, num_samples, return_textures))
		return pts, cols

	ref_losses.sample_points_from_meshes = recorded_sampler
	model.zero_grad()
	loss = ref_losses.TextureLossGTSpace()(model, dict(mesh='the GT meshes'), shapevec=lats['shapevec'], texvec=lats['texvec'], posevec=lats['posevec'])
	assert calls == [('the GT meshes', 1000, True)]
	loss.backward()
	tex = dict(points=pts.numpy(), colours=cols.numpy(), loss=np.float64(loss.item()))
	for k, v in lats.items():
		tex[k] = v.detach().numpy().copy()
		tex['grad/' + k] = v.grad.numpy().copy() if v.grad is not None else np.zeros(0, np.float32)
	tex['no_grad'] = np.array([k for k, p in model.named_parameters() if p.grad is None])
	for k, p in model.named_parameters():
		if p.grad is not None:
			gr = p.grad.numpy().copy()
			tex['grad/sd/' + k] = gr.reshape(-1)[::GRAD_STRIDE].copy() if gr.size > 4096 else gr
			tex['gradmax/sd/' + k] = np.float64(np.abs(gr).max())
	# The same loss in float64 (reference code, double parameters and inputs): how far the reference's OWN fp32 gradients are from the
	# exact ones -- the Fourier layer's features sin(2 pi x.B) have arguments of +-20, where one fp32 ulp of the argument is 2e-6 of the
	# feature, and its weight gradient is a cancelling sum over 3000 samples.  A port cannot be asked to agree with the fp32 reference
	# more closely than the fp32 reference agrees with exact arithmetic.
	m64 = NDF(template_mesh_loc=None, device='cpu', **main_kw)
	m64.load_state_dict(model.state_dict())
	m64 = m64.double()
	m64.encoder[0]._B = m64.encoder[0]._B.double()
	lat64 = {k: v.detach().double().requires_grad_(True) for k, v in lats.items()}
	pts64, cols64 = pts.double(), cols.double()
	ref_losses.sample_points_from_meshes = lambda *a, **kw: (pts64, cols64)
	loss64 = ref_losses.TextureLossGTSpace()(m64, dict(mesh='the GT meshes'), shapevec=lat64['shapevec'], texvec=lat64['texvec'], posevec=lat64['posevec'])
	loss64.backward()
	tex['loss64'] = np.float64(loss64.item())
	p32 = dict(model.named_parameters())
	for k, p in m64.named_parameters():
		if p.grad is not None:
			e = (p32[k].grad.double() - p.grad).abs().max().item() / p.grad.abs().max().item()
			tex['ref_fp32_error/sd/' + k] = np.float64(e)
			g64 = p.grad.numpy().copy()
			tex['grad64/sd/' + k] = g64.reshape(-1)[::GRAD_STRIDE].copy() if g64.size > 4096 else g64
	tex['ref_fp32_error/texvec'] = np.float64((lats['texvec'].grad.double() - lat64['texvec'].grad).abs().max().item() / lat64['texvec'].grad.abs().max().item())
	print('reference fp32 vs float64, worst relative gradient error per tensor:', {k[18:]: float('%.2g' % tex[k]) for k in tex if k.startswith('ref_fp32_error/sd/')})
	tex['masked_fraction'] = np.float64(1.0 - float((cols < 1).any(dim=-1).float().mean()))
	np.savez(os.path.join(HERE, 'texture_loss.npz'), **tex)
	print('texture_loss.npz:', len(tex), 'arrays; loss', loss.item(), 'masked', tex['masked_fraction'], 'params without grad:', list(tex['no_grad']))

	# ------------------------------------------------------------------ a checkpoint written by the reference
	labels = dict(shape=['0003', '0005'], tex=['0003', '0005'], pose=['0003-A', '0005-A', '0005-B'], reg=['0003-A', '0005-A', '0005-B'],
				  shape_val=['0011'], tex_val=['0011'], pose_val=['0011-A', '0011-B'], reg_val=['0011-A', '0011-B'])
	ck = NDF(template_mesh_loc=None, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=3, val_size=2,
			 shapevec_size=100, texvec_size=100, posevec_size=100, latent_labels=labels)
	gen = torch.Generator().manual_seed(44)
	with torch.no_grad():
		for p in ck.parameters():   # trained-looking values everywhere (tables, zero-initialised last layer, ...)
			if p.dtype.is_floating_point and p.requires_grad:
				p.add_(torch.randn(p.shape, generator=gen) * 0.02)
		tv = G.synth_positions(gen, 1, 42)
		ck.template_verts = torch.nn.Parameter(tv, requires_grad=False)
		ck.template_faces = torch.nn.Parameter(torch.randint(0, 42, (1, 80, 3), generator=gen).int(), requires_grad=False)
		ck.avg_col.copy_(torch.tensor([0.4, 0.5, 0.6]))
	ck.save_model(out_dir=HERE, fname='ref_checkpoint')
	pos = G.synth_positions(gen, 2, 64)
	with torch.no_grad():
		res = ck(pos, shapevec=ck.shapevec[['0005', '0003']], texvec=ck.texvec[['0003', '0003']], posevec=ck.posevec[['0005-B', '0003-A']])
		res1 = ck(ck.template_verts.data, shapevec=ck.shapevec_val[['0011', '0011']], texvec=ck.texvec_val[['0011', '0011']],
				  posevec=ck.posevec_val[['0011-A', '0011-B']])
	np.savez(os.path.join(HERE, 'ref_checkpoint.npz'), pos=pos.numpy(), disp=res['disp'].numpy(), col=res['col'].numpy(),
			 template_disp=res1['disp'].numpy(), template_col=res1['col'].numpy(),
			 keys=np.array(list(ck.state_dict().keys())), shapes=np.array([str(tuple(v.shape)) for v in ck.state_dict().values()]))
	print('ref_checkpoint.pth:', os.path.getsize(os.path.join(HERE, 'ref_checkpoint.pth')), 'bytes')


if __name__ == '__main__':
	main()
