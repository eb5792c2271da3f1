"""End-to-end GPU parity: ModelWithLoss.forward (reference src/model/model.py:1001-1163) with the train_3d.yaml loss set
(chamf + smooth + texture) and with the render losses (sil + pix), against the oracle composed from the same inputs.
The sampler's random draws are recorded on the GPU run and replayed in the oracle (draws are inputs)."""
import numpy as np
import pytest
import torch

from oracle import geom_ref, mlp_ref, render_ref

pytestmark = pytest.mark.gpu


def _setup(n_feet=3, n_verts=1002, gt_verts=1002, seed=0):
	from find_amd import synthetic
	from find_amd.model_with_loss import ModelWithLoss
	from find_amd.opts import Opts
	from find_amd.structures import Meshes, TexturesVertex
	opts = Opts(chamf_loss=True, smooth_loss=True, texture_loss=True, num_views=2)
	mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=n_feet, val_size=1,
						shapevec_size=100, texvec_size=100, posevec_size=100, template_mesh_loc=None)
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():
		mwl.model.mlp_disp[-1].weight.copy_(torch.randn(mwl.model.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		mwl.model.mlp_disp[-1].bias.copy_(torch.randn(3, generator=g) * 0.01)
	mwl = mwl.to('cuda')
	v, f = synthetic.template(n_verts)
	mwl.model.set_template(v.cuda(), f.cuda())
	lat = synthetic.latents(n_feet, seed=seed, device='cuda')
	with torch.no_grad():
		mwl.model.shapevec.data.copy_(lat['shapevec']); mwl.model.texvec.data.copy_(lat['texvec'])
		mwl.model.posevec.data.copy_(lat['posevec']); mwl.model.reg.data.copy_(lat['reg'])
	gv, gf, gc = synthetic.gt_feet(n_feet, gt_verts, seed=seed, device='cuda')
	gc = gc.clamp(0.05, 0.95)
	# a saturated cap (whole faces): its samples are masked out of the texture loss (losses.py:43).  1.25 rather than exactly
	# 1.0 keeps the `< 1` test away from the w0+w1+w2 = 1 +/- 1 ulp rounding boundary, where CPU and GPU may legitimately differ.
	gc[:, :gt_verts // 5] = 1.25
	batch = dict(mesh=Meshes(gv, gf, TexturesVertex(gc)), idx=torch.arange(n_feet, device='cuda'), name=[f'{i:04d}' for i in range(n_feet)])
	from find_amd.train_utils import sample_latent_vectors
	batch.update(sample_latent_vectors(batch, mwl.model.latent_vectors_train))
	return mwl, opts, batch, (gv, gf, gc)


class DrawRecorder:
	"""Wraps find_amd.functional.sample_surface -- the sampler the product runs (uniform draws from torch's device generator, faces chosen
	on the device) -- and keeps what it drew: (face_idx, uv) per call, in call order."""

	def __init__(self):
		from find_amd import functional as FN
		self.FN = FN
		self.orig = FN.sample_surface
		self.draws = []

	def __enter__(self):
		def wrapped(verts, faces, rnd, attr=None):
			out = self.orig(verts, faces, rnd, attr)
			self.draws.append((out[2].long().cpu(), out[3].cpu()))
			return out

		self.FN.sample_surface = wrapped
		return self

	def __exit__(self, *a):
		self.FN.sample_surface = self.orig

	def chamfer_and_texture(self):
		"""(GT draws of the Chamfer term, prediction draws of the Chamfer term, GT draws of the texture term), whatever order the terms
		were issued in: the texture term draws 1000 samples where the Chamfer term draws 5000 (losses.py:27,61)."""
		tex = [d for d in self.draws if d[0].shape[1] == 1000]
		cham = [d for d in self.draws if d[0].shape[1] != 1000]
		assert len(tex) == 1 and len(cham) == 2, [tuple(d[0].shape) for d in self.draws]
		return cham[0], cham[1], tex[0]


def _oracle_model(mwl, batch):
	m = mwl.model
	sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	lat = {k: batch[f'{k}_train'].detach().cpu().clone().requires_grad_(True) for k in ['shapevec', 'texvec', 'posevec', 'reg']}
	return sd, lat, m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()


def test_train_3d_loss_set_matches_oracle():
	mwl, opts, batch, (gv, gf, gc) = _setup()
	with DrawRecorder() as rec:
		loss, losses = mwl(batch, 0, opts, chamf=True, smooth=True, texture=True)
	assert set(losses) == {'loss_chamf', 'loss_smooth', 'loss_tex'}
	loss.backward()
	# ---- oracle with the recorded draws
	(fi_gt, uv_gt), (fi_pr, uv_pr), (fi_tx, uv_tx) = rec.chamfer_and_texture()
	sd, lat, B, tv, tf = _oracle_model(mwl, batch)
	res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
	gvc, gfc, gcc = gv.cpu(), gf.cpu(), gc.cpu()
	gt_s = geom_ref.sample_points(gvc, gfc, fi_gt, uv_gt)
	pr_s = geom_ref.sample_points(res['verts'], tf, fi_pr, uv_pr)
	l_ch = geom_ref.chamfer_distance(pr_s, gt_s)
	l_sm = geom_ref.mesh_smoothness(res['verts'], tf)
	tx_p, tx_c = geom_ref.sample_points(gvc, gfc, fi_tx, uv_tx, attr=gcc)
	col = mlp_ref.mlp_forward(sd, B, tx_p, lat['shapevec'], lat['texvec'], lat['posevec'])['col']
	mask = (tx_c < 1).any(dim=-1, keepdim=True).expand(-1, -1, 3)
	l_tx = (torch.nn.functional.mse_loss(col, tx_c, reduction='none') * mask).mean()
	ref = {'loss_chamf': l_ch * 10000., 'loss_smooth': l_sm * 1000., 'loss_tex': l_tx * 1.}  # opts.py:97-99
	for k in ref:
		assert abs(losses[k].item() - ref[k].item()) < 1e-4 * max(1.0, abs(ref[k].item())), (k, losses[k].item(), ref[k].item())
	assert float((~mask).float().mean()) > 0.05  # the white-sample mask is exercised
	rl = sum(ref.values())
	assert abs(loss.item() - rl.item()) < 1e-4 * max(1.0, abs(rl.item()))
	rl.backward()
	for k, name in [('shapevec', 'shapevec'), ('texvec', 'texvec'), ('posevec', 'posevec'), ('reg', 'reg')]:
		got = getattr(mwl.model, name).data.grad.cpu()
		want = lat[k].grad
		s = max(1e-3, want.abs().max().item())
		assert (got - want).abs().max().item() < 1e-4 * s, (k, (got - want).abs().max().item(), s)   # measured 1e-5 (nearest-neighbour picks are identical)
	for k in ['base.0.weight', 'base.4.bias', 'mlp_disp.2.weight', 'mlp_disp.6.weight', 'mlp_col.0.weight', 'mlp_col.6.bias']:
		got = dict(mwl.model.named_parameters())[k].grad.cpu()
		want = sd[k].grad
		s = max(1e-3, want.abs().max().item())
		assert (got - want).abs().max().item() < 1e-4 * s, (k, (got - want).abs().max().item(), s)   # measured 1e-5 (nearest-neighbour picks are identical)


def test_z_cutoff_variants_match_oracle():
	mwl, opts, batch, (gv, gf, gc) = _setup(seed=2)
	for kw in [dict(use_z_cutoff=True), dict(gt_z_cutoff=0.01)]:
		with DrawRecorder() as rec:
			loss, losses = mwl(batch, 0, opts, chamf=True, **kw)
		(fi_gt, uv_gt), (fi_pr, uv_pr) = rec.draws
		sd, lat, B, tv, tf = _oracle_model(mwl, batch)
		with torch.no_grad():
			res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
			gt_s = geom_ref.sample_points(gv.cpu(), gf.cpu(), fi_gt, uv_gt)
			pr_s = geom_ref.sample_points(res['verts'], tf, fi_pr, uv_pr)
			tot = 0.0
			for b in range(gt_s.shape[0]):  # ragged clouds, one at a time, exactly as losses.py:69-85 builds them
				if 'use_z_cutoff' in kw:
					p, q = pr_s[b][pr_s[b, :, 2] <= 0.07], gt_s[b][gt_s[b, :, 2] <= 0.07]
				else:
					p, q = pr_s[b], gt_s[b][gt_s[b, :, 2] <= 0.01]
				tot = tot + geom_ref.chamfer_distance(p[None], q[None])
			ref = tot / gt_s.shape[0] * 10000.
		assert abs(losses['loss_chamf'].item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item())), kw


def test_render_losses_match_oracle():
	mwl, opts, batch, (gv, gf, gc) = _setup(n_feet=2, seed=3)
	mwl.rdr = type(mwl.rdr)(image_size=64, device='cuda')
	np.random.seed(11)
	R, T = mwl.rdr.sample_views(nviews=2, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	out = mwl(batch, 0, opts, sil=True, pix=True, render_foot=True, return_renders=True, views=(R, T))
	loss, losses, renders = out
	assert set(losses) == {'loss_pix', 'loss_sil'} and set(renders) == {'pred', 'gt'}
	assert renders['pred']['image'].shape == (2, 2, 64, 64, 3) and renders['gt']['mask'].shape == (2, 2, 64, 64)
	loss.backward()
	# oracle forward
	sd, lat, B, tv, tf = _oracle_model(mwl, batch)
	with torch.no_grad():
		res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
	gt = render_ref.render(gv.cpu().numpy(), gf.cpu().numpy(), gc.cpu().numpy(), R.numpy(), T.numpy(), image_size=64)
	pr = render_ref.render(res['verts'].numpy(), tf.numpy(), res['col'].numpy(), R.numpy(), T.numpy(), image_size=64)
	sil = float(((pr['mask'] - gt['mask']) ** 2).mean()) * 5.0
	pix = float(((pr['image'] * pr['mask'][..., None] - gt['image'] * gt['mask'][..., None]) ** 2).mean()) * 1.0
	assert abs(losses['loss_sil'].item() - sil) < 1e-4 * max(1.0, sil), (losses['loss_sil'].item(), sil)
	assert abs(losses['loss_pix'].item() - pix) < 1e-4 * max(1.0, pix), (losses['loss_pix'].item(), pix)
	# gradients through both render losses against autograd through the oracle: the MLP on torch-CPU, the differentiable restatement of
	# the fragment math (render_ref.torch_mask / torch_phong_image) on the discrete face selection of the oracle's own rasteriser
	for name in ['reg', 'shapevec', 'texvec', 'posevec']:
		g = getattr(mwl.model, name).data.grad
		assert g is not None and torch.isfinite(g).all() and g.abs().max().item() > 0, name
	rp = render_ref.default_params(64)
	ro = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
	vproj = render_ref.project(rp, ro['verts'].detach().numpy(), R.numpy(), T.numpy())
	p2f_k, _, _, _ = render_ref.rasterize(vproj, tf.numpy(), 2, 64, 64, 100, rp.sil_blur_radius)
	p2f_1, _, _, _ = render_ref.rasterize(vproj, tf.numpy(), 2, 64, 64, 1, 0.0)
	om = render_ref.torch_mask(rp, ro['verts'], tf, R, T, torch.from_numpy(p2f_k).long(), 2)
	oi = render_ref.torch_phong_image(rp, ro['verts'], ro['col'], tf, R, T, torch.from_numpy(p2f_1).long(), 2)
	gm, gi = torch.from_numpy(gt['mask']), torch.from_numpy(gt['image'])
	ol = ((om - gm) ** 2).mean() * 5.0 + ((oi * om.unsqueeze(-1) - gi * gm.unsqueeze(-1)) ** 2).mean() * 1.0
	assert abs(ol.item() - loss.item()) < 1e-4 * max(1.0, abs(ol.item()))
	ol.backward()
	worst = 0.0
	for k in ['shapevec', 'texvec', 'posevec', 'reg']:
		got, want = getattr(mwl.model, k).data.grad.cpu(), lat[k].grad
		err = (got - want).abs().max().item() / max(1e-6, want.abs().max().item())
		worst = max(worst, err)
		assert err < 1e-4, (k, err)
	for k in ['base.0.weight', 'base.4.bias', 'mlp_disp.2.weight', 'mlp_disp.6.weight', 'mlp_col.0.weight', 'mlp_col.6.bias']:
		got, want = dict(mwl.model.named_parameters())[k].grad.cpu(), sd[k].grad
		err = (got - want).abs().max().item() / max(1e-6, want.abs().max().item())
		worst = max(worst, err)
		assert err < 1e-4, (k, err)
	print(f'render losses: worst gradient error {worst:.2e} of the tensor maximum')


def test_gt_render_on_the_second_stream_changes_nothing():
	"""ModelWithLoss renders the GT scans on a second stream beside the predicted render (model_with_loss.OVERLAP_GT_RENDER): renders,
	losses and gradients are those of the one-stream order -- over several steps, so that a missing wait between the streams (or a buffer
	handed back to the allocator too early) would show."""
	from find_amd import model_with_loss as MWL
	from find_amd.train_utils import sample_latent_vectors
	mwl, opts, batch, _ = _setup(n_feet=3, seed=5)
	mwl.rdr = type(mwl.rdr)(image_size=128, device='cuda')
	np.random.seed(12)
	R, T = mwl.rdr.sample_views(nviews=3, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	names = ['reg', 'shapevec', 'texvec', 'posevec']
	out = {}
	prev = MWL.OVERLAP_GT_RENDER
	try:
		for mode in (True, False):
			MWL.OVERLAP_GT_RENDER = mode
			rec = []
			for _ in range(4):
				mwl.zero_grad(set_to_none=True)
				batch.update(sample_latent_vectors(batch, mwl.model.latent_vectors_train))
				loss, losses, renders = mwl(batch, 0, opts, sil=True, pix=True, render_foot=True, return_renders=True, views=(R, T))
				loss.backward()
				# (garbage in between: a buffer freed on one stream and reused on the other would be overwritten here)
				junk = [torch.full((3, 3, 128, 128, 3), 7.0, device='cuda') for _ in range(4)]
				del junk
				rec.append((loss.detach().clone(), renders['gt']['mask'].clone(), renders['gt']['image'].clone(), renders['pred']['mask'].detach().clone(),
							[getattr(mwl.model, n).data.grad.clone() for n in names], mwl.model.base[0].weight.grad.clone()))
			torch.cuda.synchronize()
			out[mode] = rec
	finally:
		MWL.OVERLAP_GT_RENDER = prev
	# every step has the same inputs (no optimiser step): all eight results agree -- the silhouettes bit for bit, what goes through the
	# vertex normals (accumulated with atomics: order-dependent rounding) to rounding
	ref = out[False][0]

	def close(x, y, tol):
		return (x - y).abs().max().item() <= tol * max(1e-6, y.abs().max().item())

	for rec in out[True] + out[False][1:]:
		assert torch.equal(rec[1], ref[1]) and torch.equal(rec[3], ref[3])
		assert close(rec[0], ref[0], 1e-6) and close(rec[2], ref[2], 1e-5)
		for x, y in zip(rec[4], ref[4]):
			assert close(x, y, 1e-4)
		assert close(rec[5], ref[5], 1e-4)


def test_silhouette_only_step_skips_the_images_and_keeps_its_numbers():
	"""With the silhouette loss alone nothing reads the rendered images: ModelWithLoss does not render them (no shading, no vertex normals,
	no RGB backward).  Loss and gradients must be those of the same step with the images rendered (return_renders=True)."""
	from find_amd.train_utils import sample_latent_vectors
	mwl, opts, batch, _ = _setup(n_feet=2, seed=6)
	mwl.rdr = type(mwl.rdr)(image_size=96, device='cuda')
	np.random.seed(13)
	R, T = mwl.rdr.sample_views(nviews=2, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	names = ['reg', 'shapevec', 'texvec', 'posevec']
	got = {}
	for with_images in (True, False):
		mwl.zero_grad(set_to_none=True)
		batch.update(sample_latent_vectors(batch, mwl.model.latent_vectors_train))
		out = mwl(batch, 0, opts, sil=True, render_foot=True, return_renders=with_images, views=(R, T))
		if with_images:
			assert 'image' in out[2]['pred'] and 'image' in out[2]['gt']
		out[0].backward()
		got[with_images] = (out[0].detach().clone(), out[1]['loss_sil'].detach().clone(),
							[None if getattr(mwl.model, n).data.grad is None else getattr(mwl.model, n).data.grad.clone() for n in names],
							mwl.model.base[0].weight.grad.clone())
	a, b = got[True], got[False]
	assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
	for x, y in zip(a[2], b[2]):
		assert (x is None) == (y is None)
		if x is not None:   # (texvec: no gradient either way -- the silhouette does not depend on colour)
			assert (x - y).abs().max().item() <= 1e-6 * max(1e-6, x.abs().max().item())
	assert (a[3] - b[3]).abs().max().item() <= 1e-6 * max(1e-6, a[3].abs().max().item())


def test_pixel_loss_step_without_returned_renders_keeps_its_numbers():
	"""With copy_mask_out the predicted image is whitened where the GT's slicing plane hides the foot (model.py:1091-1094) -- for a caller who
	looks at it.  The pixel loss reads the image through the mask, which is zero there, so a step that does not return its renders skips the
	whitening: loss and gradients must be those of the step that returns them."""
	from find_amd.train_utils import sample_latent_vectors
	mwl, opts, batch, _ = _setup(n_feet=2, seed=7)
	mwl.rdr = type(mwl.rdr)(image_size=96, device='cuda')
	np.random.seed(14)
	R, T = mwl.rdr.sample_views(nviews=2, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	# faces to mask out, so that `hidden` is not empty
	batch = dict(batch)
	batch['masked_faces'] = [torch.arange(0, 400), torch.arange(200, 900)]
	names = ['reg', 'shapevec', 'texvec', 'posevec']
	got = {}
	for ret in (True, False):
		mwl.zero_grad(set_to_none=True)
		batch.update(sample_latent_vectors(batch, mwl.model.latent_vectors_train))
		out = mwl(batch, 0, opts, sil=True, pix=True, render_foot=True, return_renders=ret, views=(R, T))
		if ret:
			assert bool(out[2]['gt']['mask_out_masks'].any()), 'the scene must hide some pixels'
		out[0].backward()
		got[ret] = (out[0].detach().clone(), [getattr(mwl.model, n).data.grad.clone() for n in names], mwl.model.base[0].weight.grad.clone())
	a, b = got[True], got[False]
	assert abs(a[0].item() - b[0].item()) <= 1e-6 * max(1e-6, abs(a[0].item()))
	for x, y in zip(a[1], b[1]):
		assert (x - y).abs().max().item() <= 1e-5 * max(1e-6, x.abs().max().item())
	assert (a[2] - b[2]).abs().max().item() <= 1e-5 * max(1e-6, a[2].abs().max().item())


def test_loss_weights_are_applied_and_flags_respected():
	mwl, opts, batch, _ = _setup(n_feet=2, seed=4)
	l0, d0 = mwl(batch, 0, opts)
	assert d0 == {} and l0 == 0  # no loss enabled -> sum of nothing (trainer.py:108 `if loss == 0: continue`)
	torch.manual_seed(0)
	l1, d1 = mwl(batch, 0, opts, smooth=True)
	opts.weight_smooth = 2000.
	l2, d2 = mwl(batch, 0, opts, smooth=True)
	assert abs(l2.item() - 2 * l1.item()) < 1e-5 * abs(l2.item())
	with pytest.raises(NotImplementedError):
		mwl(batch, 0, opts, vgg_perc=True)


def test_lazy_colour_head_changes_no_number_and_answers_a_reader():
	"""A step that renders nothing defers the colour head of the template pass (model.get_meshes(lazy_colours=True)): losses and every
	gradient are those of the eager evaluation (the reference's, model.py:455-504; the losses bit for bit), nothing is computed until somebody reads
	res['col'] / meshes.textures, and a reader gets the eager values."""
	import find_amd.model_with_loss as MWL
	from find_amd.structures import LazyTexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	from test_gpu_train3d import FixedDraws
	mwl, opts, batch, (gv, gf, gc) = _setup()
	F_gt, F_t = gf.shape[0], mwl.model.template_faces.shape[1]
	g = torch.Generator().manual_seed(3)
	n = gv.shape[0]
	draws = [(torch.randint(0, F_gt, (n, 5000), generator=g).cuda(), torch.rand(n, 5000, 2, generator=g).cuda()),
			 (torch.randint(0, F_t, (n, 5000), generator=g).cuda(), torch.rand(n, 5000, 2, generator=g).cuda()),
			 (torch.randint(0, F_gt, (n, 1000), generator=g).cuda(), torch.rand(n, 1000, 2, generator=g).cuda())]
	out = {}
	prev = MWL.LAZY_COLOURS
	try:
		for lazy in (False, True):
			MWL.LAZY_COLOURS = lazy
			mwl.zero_grad(set_to_none=True)
			batch.update(sample_latent_vectors(batch, mwl.model.latent_vectors_train))   # (a fresh gather node per backward pass)
			with FixedDraws(draws):
				loss, losses = mwl(batch, 0, opts, chamf=True, smooth=True, texture=True)
			loss.backward()
			torch.cuda.synchronize()
			out[lazy] = (loss.detach().clone(), {k: v.detach().clone() for k, v in losses.items()},
						 {k: p.grad.detach().clone() for k, p in mwl.model.named_parameters() if p.grad is not None})
	finally:
		MWL.LAZY_COLOURS = prev
	assert torch.equal(out[True][0], out[False][0])
	assert all(torch.equal(out[True][1][k], out[False][1][k]) for k in out[False][1])
	assert set(out[True][2]) == set(out[False][2])
	for k in out[False][2]:
		# (equal up to the order of the weight-gradient partial sums: a one-head call splits its rows differently from a two-head call)
		a, b = out[True][2][k], out[False][2][k]
		assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-12), k
	# ---- a reader of the deferred head
	lat = {k: batch[f'{k}_train'] for k in ('shapevec', 'reg', 'texvec', 'posevec')}
	with torch.no_grad():
		eager = mwl.model.get_meshes(**lat)
		lazy = mwl.model.get_meshes(**lat, lazy_colours=True)
	assert isinstance(lazy['meshes'].textures, LazyTexturesVertex) and not lazy['meshes'].textures.evaluated and 'col' not in lazy.keys()
	assert torch.equal(lazy['verts'], eager['verts']) and torch.equal(lazy['disp'], eager['disp'])
	assert torch.equal(lazy['meshes'].textures.verts_features_padded(), eager['meshes'].textures.verts_features_padded())
	assert lazy['meshes'].textures.evaluated and torch.equal(lazy['col'], eager['col'])
	# a rendering step asks for both heads up front: the pixel loss differentiates through the colours
	mwl.zero_grad(set_to_none=True)
	batch.update(sample_latent_vectors(batch, mwl.model.latent_vectors_train))
	loss, losses = mwl(batch, 0, opts, chamf=False, smooth=False, texture=False, pix=True, sil=True, render_foot=True)
	loss.backward()
	assert mwl.model.mlp_col[0].weight.grad is not None and mwl.model.mlp_col[0].weight.grad.abs().max().item() > 0


def test_a_step_leaves_nothing_to_the_cyclic_garbage_collector():
	"""With the garbage collector switched off, the device memory held after a step must not grow from step to step: nothing of a step --
	result dict, meshes, deferred colour head, autograd graph with its saved activations -- may sit in a reference cycle (the first
	version of the deferred colour head did: textures -> closure -> result dict -> meshes -> textures; the allocator grew by a step's
	memory per step and a long loop slowed down by half)."""
	import gc
	from find_amd.train_utils import sample_latent_vectors
	from find_amd import functional_render as FR
	mwl, opts, batch, _ = _setup()
	for flags in (dict(chamf=True, smooth=True, texture=True), dict(sil=True, render_foot=True), dict(sil=True, pix=True, chamf=True, render_foot=True)):
		held = []
		gc.collect()
		gc.disable()
		try:
			for _ in range(5):
				mwl.zero_grad(set_to_none=True)
				b = dict(batch)
				b.update(sample_latent_vectors(b, mwl.model.latent_vectors_train))
				loss, losses = mwl(b, 0, opts, **flags)
				loss.backward()
				del loss, losses, b
				torch.cuda.synchronize()
				FR.check_render_flags(wait=True)   # (the watchdog's pending entries are host-side)
				held.append(torch.cuda.memory_allocated())
		finally:
			gc.enable()
		# (a leak grows by a step's memory EVERY step -- 27 MB here with the old cycle; workspace buffers that two steps take turns with do not)
		assert held[4] - held[2] < 2 ** 20 and held[3] - held[1] < 2 ** 20, (flags, held)
