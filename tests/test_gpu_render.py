"""GPU parity of the HIP renderer (C-ABI find_render_fwd / find_render_bwd) against the render oracle
(oracle/raster_ref.c naive rasteriser + shaders; oracle/render_ref.py differentiable restatement for gradients).
Floats -- outputs AND gradients -- within the north_star's 1e-4 (gradients relative to the tensor's largest entry; the kernel's
v_rcp_f32 normalisations and its atomics' summation order stay far below that: measured 1e-6 .. 1.3e-5).  The nearest-face index
map is compared exactly: a differing pixel must sit on a face boundary to rounding (_assert_index_mismatches_are_edge_ties; none
differ at the test sizes).  The one looser bound is the K-overflow gradient test, which says why."""
import numpy as np
import pytest
import torch

from oracle import camera_ref, render_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['list', 'band'])
def rasteriser(request):
	"""Every test of this file runs on BOTH forward rasterisers: raster_kernel (candidate lists: what find_render_fwd picks below 384^2) and
	raster_band_kernel (csrc/render_band.h, the list-free K-nearest rule: from 384^2 on) -- the switch bits 4096 / 2048 of
	find_render_switches force one at every size."""
	from find_amd import _lib
	global RASTER_BASE
	RASTER_BASE = 2048 if request.param == 'band' else 4096
	_lib.set_tuning('raster_ablate', RASTER_BASE)
	yield request.param
	_lib.set_tuning('raster_ablate', 0)


RASTER_BASE = 0
TOL = 1e-4
GRAD_TOL = 1e-4   # gradients relative to the tensor's largest entry; measured on MI355X: silhouette 1e-5, Phong image 1e-6 .. 1.3e-5


def _scene(n_meshes=2, rings=14, segs=18, seed=0, n_views=2):
	from find_amd import synthetic
	v, f = synthetic.ellipsoid_mesh(rings, segs)
	g = torch.Generator().manual_seed(seed)
	verts = v[None] * (1 + 0.1 * torch.rand(n_meshes, 1, 3, generator=g)) + 0.002 * torch.randn(n_meshes, v.shape[0], 3, generator=g)
	cols = torch.rand(n_meshes, v.shape[0], 3, generator=g)
	rng = np.random.RandomState(seed + 7)
	R, T = camera_ref.look_at_view_transform(dist=np.full(n_views, 0.3), elev=rng.uniform(-90, 90, n_views), azim=rng.uniform(-90, 90, n_views), up=((1, 0, 0),))
	return verts, f, cols, torch.from_numpy(R), torch.from_numpy(T)


def _render_gpu(verts, faces, cols, R, T, size, **kw):
	from find_amd import functional_render as FR
	params = FR.make_params(size)
	return FR.render(verts.cuda(), cols.cuda() if cols is not None else None, faces.cuda(), R.cuda(), T.cuda(), params, **kw), params


def _assert_index_mismatches_are_edge_ties(p2f_gpu, p2f_ref, verts, faces, R, T, size, faces_per_mesh=None):
	"""Index work is compared exactly: wherever the two nearest-face maps differ, the pixel centre must lie ON the boundary of a face
	one of them picked -- to rounding: the smallest barycentric coordinate of that face, evaluated in float64 from the oracle's
	projected vertices, is below 1e-5 in magnitude (`inside` is w > 0 for all three; the two implementations round the edge functions
	differently, so only such pixels may legitimately flip).  Everything else is an error, however rare."""
	rp = render_ref.default_params(size)
	vproj = render_ref.project(rp, np.ascontiguousarray(verts), np.ascontiguousarray(R), np.ascontiguousarray(T)).astype(np.float64)   # (n_img, V, 3)
	faces = np.asarray(faces)
	a, b = np.asarray(p2f_gpu).reshape(-1, size, size), np.asarray(p2f_ref).reshape(-1, size, size)
	F = faces.shape[-2]
	n_views = R.shape[0]
	bad = np.argwhere(a != b)
	worst = 0.0
	for img, yi, xi in bad:
		px, py = 1.0 - (2.0 * xi + 1.0) / size, 1.0 - (2.0 * yi + 1.0) / size
		tie = False
		for packed in (a[img, yi, xi], b[img, yi, xi]):
			if packed < 0:
				continue
			f = packed - img * F
			assert 0 <= f < F, (packed, img, F)
			fv = faces[f] if faces.ndim == 2 else faces[img // n_views][f]
			(x0, y0, _), (x1, y1, _), (x2, y2, _) = vproj[img][fv]
			area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0)
			w = [((px - x1) * (y2 - y1) - (py - y1) * (x2 - x1)) / area, ((px - x2) * (y0 - y2) - (py - y2) * (x0 - x2)) / area,
				 ((px - x0) * (y1 - y0) - (py - y0) * (x1 - x0)) / area]
			m = min(abs(x) for x in w)
			if m < 1e-5:
				tie = True
				worst = max(worst, m)
		assert tie, f'pix_to_face differs at image {img} pixel ({yi},{xi}) [{a[img, yi, xi]} vs {b[img, yi, xi]}] away from any face boundary'
	return len(bad), worst


@pytest.mark.parametrize('size', [64, 128])
def test_forward_mask_image_vs_oracle(size):
	verts, faces, cols, R, T = _scene()
	(mask, image, p2f, zbuf), _ = _render_gpu(verts, faces, cols, R, T, size, want_frags=True)
	ref = render_ref.render(verts.numpy(), faces.numpy(), cols.numpy(), R.numpy(), T.numpy(), image_size=size)
	em = np.abs(mask.cpu().numpy() - ref['mask']).max()
	assert em < TOL, em
	assert ref['mask'].max() > 0.99 and ref['mask'].min() == 0.0
	same = (p2f.cpu().numpy() == ref['pix_to_face'])
	n_bad, worst = _assert_index_mismatches_are_edge_ties(p2f.cpu().numpy(), ref['pix_to_face'], verts.numpy(), faces.numpy(), R.numpy(), T.numpy(), size)
	print(f'pix_to_face @{size}: {n_bad} of {same.size} pixels differ, all on a face boundary (|w_min| <= {worst:.1e})')
	assert same.mean() > 0.999, same.mean()
	ei = np.abs(image.cpu().numpy() - ref['image'])[same].max()
	assert ei < TOL, ei
	ez = np.abs(zbuf.cpu().numpy() - ref['zbuf'])[same].max()
	assert ez < 1e-5, ez
	# where the picked face differs the images still agree closely (the pixel centre sits on a shared edge)
	assert np.abs(image.cpu().numpy() - ref['image']).max() < 5e-2


def test_forward_only_mask_or_only_image():
	verts, faces, cols, R, T = _scene(n_meshes=1, n_views=3, seed=3)
	(m1, i1, _, _), _ = _render_gpu(verts, faces, cols, R, T, 64)
	(m2, i2, _, _), _ = _render_gpu(verts, faces, None, R, T, 64, want_image=False)
	(m3, i3, _, _), _ = _render_gpu(verts, faces, cols, R, T, 64, want_mask=False)
	assert i2 is None and m3 is None
	assert torch.equal(m1, m2)
	assert (i1 - i3).abs().max().item() < 1e-5  # vertex normals are accumulated with float atomics (order varies)


def test_silhouette_backward_vs_oracle_autograd():
	size = 48
	verts, faces, cols, R, T = _scene(n_meshes=2, rings=8, segs=10, seed=1)
	vg = verts.clone().cuda().requires_grad_(True)
	(mask, _, _, _), params = _render_gpu(vg, faces, None, R, T, size, want_image=False)
	gt = torch.rand(mask.shape, generator=torch.Generator().manual_seed(2))
	loss = ((mask - gt.cuda()) ** 2).mean()
	loss.backward()
	# oracle: fragments chosen by the C rasteriser, gradients by autograd through the torch restatement
	rp = render_ref.default_params(size)
	vproj = render_ref.project(rp, verts.numpy(), R.numpy(), T.numpy())
	p2f, _, _, _ = render_ref.rasterize(vproj, faces.numpy(), R.shape[0], size, size, 100, rp.sil_blur_radius)
	vr = verts.clone().requires_grad_(True)
	rm = render_ref.torch_mask(rp, vr, faces, R, T, torch.from_numpy(p2f).long(), R.shape[0])
	assert (mask.detach().cpu() - rm.detach()).abs().max().item() < TOL
	rl = ((rm - gt) ** 2).mean()
	rl.backward()
	assert abs(loss.item() - rl.item()) < 1e-6
	scale = vr.grad.abs().max().item()
	assert scale > 0
	err = (vg.grad.cpu() - vr.grad).abs().max().item()
	print(f'silhouette backward: max err {err:.3e} of scale {scale:.3e} = {err / scale:.2e}')
	assert err < GRAD_TOL * scale, (err, scale)


@pytest.mark.parametrize('rings,segs,size', [(8, 10, 48), (40, 50, 24)])
def test_image_backward_vs_oracle_autograd(rings, segs, size):
	"""(8, 10) @48: the face-centric RGB backward; (40, 50) @24: faces outnumber pixels 7:1 -> the pixel-centric one."""
	verts, faces, cols, R, T = _scene(n_meshes=2, rings=rings, segs=segs, seed=4)
	vg = verts.clone().cuda().requires_grad_(True)
	cg = cols.clone().cuda().requires_grad_(True)
	(_, image, p2f, _), params = _render_gpu(vg, faces, cg, R, T, size, want_mask=False, want_frags=True)
	w = torch.rand(image.shape, generator=torch.Generator().manual_seed(5))
	(image * w.cuda()).sum().backward()
	rp = render_ref.default_params(size)
	vr = verts.clone().requires_grad_(True)
	cr = cols.clone().requires_grad_(True)
	ri = render_ref.torch_phong_image(rp, vr, cr, faces, R, T, p2f.cpu().long().reshape(-1, size, size, 1), R.shape[0])
	assert (image.detach().cpu() - ri.detach()).abs().max().item() < TOL
	(ri * w).sum().backward()
	sc = cr.grad.abs().max().item()
	assert (cg.grad.cpu() - cr.grad).abs().max().item() < 1e-4 * sc
	sv = vr.grad.abs().max().item()
	err = (vg.grad.cpu() - vr.grad).abs().max().item()
	print(f'image backward ({rings}x{segs} @{size}): max err {err:.3e} of scale {sv:.3e} = {err / sv:.2e}')
	assert err < GRAD_TOL * sv, (err, sv)


def test_c3_size_properties():
	"""C3-size render (16 feet x 4 views @256^2 of the 6890-vertex template): masks lie in [0,1], are 0 far from and ~1
	deep inside the silhouette, rendering is invariant to the order of the batch, and no face straddles the clip plane."""
	from find_amd import functional_render as FR
	from find_amd import synthetic
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(0)
	verts = (v[None] * (1 + 0.1 * torch.rand(16, 1, 3, generator=g))).cuda()
	cols = torch.rand(16, v.shape[0], 3, generator=g).cuda()
	rng = np.random.RandomState(7)
	R, T = camera_ref.look_at_view_transform(dist=np.full(4, 0.3), elev=rng.uniform(-90, 90, 4), azim=rng.uniform(-90, 90, 4), up=((1, 0, 0),))
	R, T = torch.from_numpy(R).cuda(), torch.from_numpy(T).cuda()
	params = FR.make_params(256)
	mask, image, _, _ = FR.render(verts, cols, f.cuda(), R, T, params)
	assert mask.shape == (16, 4, 256, 256) and image.shape == (16, 4, 256, 256, 3)
	assert mask.min().item() >= 0.0 and mask.max().item() <= 1.0
	assert mask[:, :, 0, 0].abs().max().item() == 0.0
	frac = (mask > 0.5).float().mean().item()
	assert 0.02 < frac < 0.6, frac
	assert (image[mask == 0] == 1).all()  # background is white where nothing is near
	perm = torch.randperm(16, generator=torch.Generator().manual_seed(1))
	m2, i2, _, _ = FR.render(verts[perm.cuda()], cols[perm.cuda()], f.cuda(), R, T, params)
	assert torch.equal(m2, mask[perm.cuda()])
	assert (i2 - image[perm.cuda()]).abs().max().item() < 1e-5  # normals are accumulated with float atomics


def test_c4_rank_share_properties():
	"""C4 per-rank share (16 of the 128 feet x 4 views @512^2, 6890-vertex template): resolution consistency -- the 512^2
	silhouette area fraction agrees with the 256^2 one, masks stay in [0,1], the render is batch-order invariant, and the
	silhouette loss gradient is finite and non-zero only on vertices."""
	from find_amd import functional_render as FR
	from find_amd import synthetic
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(2)
	verts = (v[None] * (1 + 0.1 * torch.rand(16, 1, 3, generator=g))).cuda().requires_grad_(True)
	cols = torch.rand(16, v.shape[0], 3, generator=g).cuda()
	rng = np.random.RandomState(11)
	R, T = camera_ref.look_at_view_transform(dist=np.full(4, 0.3), elev=rng.uniform(-90, 90, 4), azim=rng.uniform(-90, 90, 4), up=((1, 0, 0),))
	R, T = torch.from_numpy(R).cuda(), torch.from_numpy(T).cuda()
	m512, i512, _, _ = FR.render(verts, cols, f.cuda(), R, T, FR.make_params(512))
	m256, _, _, _ = FR.render(verts.detach(), cols, f.cuda(), R, T, FR.make_params(256))
	assert m512.shape == (16, 4, 512, 512) and i512.shape == (16, 4, 512, 512, 3)
	assert m512.min().item() >= 0.0 and m512.max().item() <= 1.0
	a512 = (m512 > 0.5).float().mean(dim=(2, 3))
	a256 = (m256 > 0.5).float().mean(dim=(2, 3))
	assert (a512 - a256).abs().max().item() < 0.01  # same silhouettes, finer grid
	perm = torch.randperm(16, generator=torch.Generator().manual_seed(3)).cuda()
	m2, _, _, _ = FR.render(verts.detach()[perm], cols[perm], f.cuda(), R, T, FR.make_params(512))
	assert torch.equal(m2, m512.detach()[perm])
	target = (m256.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) > 0.5).float()
	loss = ((m512 - target) ** 2).mean()
	loss.backward()
	assert torch.isfinite(verts.grad).all() and verts.grad.abs().max().item() > 0


def test_full_template_mask_vs_oracle_with_k_overflow():
	"""6890-vertex template at 256^2 (C3 geometry, 1 foot x 2 views): here ~6% of the pixels see more than
	faces_per_pixel = 100 silhouette candidates, where PyTorch3D keeps the 100 nearest in depth and the HIP kernel blends
	all of them.  The oracle implements the K-nearest rule: the masks must still agree to the north_star tolerance."""
	from find_amd import functional_render as FR
	from find_amd import synthetic
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(5)
	verts = v[None] * (1 + 0.1 * torch.rand(1, 1, 3, generator=g))
	rng = np.random.RandomState(3)
	R, T = camera_ref.look_at_view_transform(dist=np.full(2, 0.3), elev=rng.uniform(-90, 90, 2), azim=rng.uniform(-90, 90, 2), up=((1, 0, 0),))
	R, T = torch.from_numpy(R), torch.from_numpy(T)
	mask, _, _, _ = FR.render(verts.cuda(), None, f.cuda(), R.cuda(), T.cuda(), FR.make_params(256), want_image=False)
	ref = render_ref.render(verts.numpy(), f.numpy(), None, R.numpy(), T.numpy(), image_size=256, want_image=False)
	err = np.abs(mask.cpu().numpy() - ref['mask'])
	assert err.max() < TOL, (err.max(), int((err > TOL).sum()))


def test_lists_longer_than_the_binning_buffer_and_the_pole_of_a_scan_vs_oracle():
	"""Tiles whose face list outgrows the binning wave's LDS buffer (BIN_CAP = 2048 entries: the list then goes out in face order, no
	depth-slab sort, no early exit) come about naturally with a dense mesh on a small image: the 50 002-vertex template at 64^2 puts
	several thousand faces into every covered tile and hundreds of candidates on every covered pixel; and the pole of a lat-long GT scan
	turned towards the camera is the fan the tie fix-up exists for (every face of the fan shares the pole's depth).  Masks against the
	oracle's K-nearest blend at the north_star tolerance, outside provable depth ties at the K-th place (as in the gradient test below)."""
	from find_amd import synthetic
	size = 64
	rp = render_ref.default_params(size)
	for n_verts, elev, azim in ((50002, 25.0, -40.0), (10002, 0.0, 0.0)):
		v, f = synthetic.template(n_verts)
		verts = v[None].clone()
		# (elev 0, azim 0): the camera sits on the +z axis and looks straight at the pole of the lat-long grid
		R, T = camera_ref.look_at_view_transform(dist=np.full(1, 0.3), elev=np.array([elev]), azim=np.array([azim]), up=((1, 0, 0),))
		R, T = torch.from_numpy(R), torch.from_numpy(T)
		(mask, _, _, _), _ = _render_gpu(verts, f, None, R, T, size, want_image=False)
		vproj = render_ref.project(rp, verts.numpy(), R.numpy(), T.numpy())
		p2f101, z101, _, d101 = render_ref.rasterize(vproj, f.numpy(), 1, size, size, 101, rp.sil_blur_radius)
		full = p2f101[..., 99] >= 0
		assert full.mean() > 0.05, full.mean()
		z99, z100 = z101[..., 99].astype(np.float64), z101[..., 100].astype(np.float64)
		tie = ((p2f101[..., 100] >= 0) & (z100 - z99 <= 4e-6 * z99)).reshape(mask.shape)
		# the other discontinuity of the K-buffer: a fragment whose squared distance equals the blur radius to rounding (the same 4e-6 = ~30 ulp)
		# is a candidate on one side of the rounding and none on the other -- a fragment of p = 1e-4, nothing by itself, but where the pixel is
		# full it decides which face is the K-th: a provable tie like the depth ties (round 5: one pixel of the 50 002-vertex scene, 4 ulp)
		edge = (full & ((p2f101 >= 0) & (np.abs(d101.astype(np.float64) - rp.sil_blur_radius) <= 4e-6 * rp.sil_blur_radius)).any(-1)).reshape(mask.shape)
		assert edge.sum() <= 4, int(edge.sum())
		tie = tie | edge
		ref = render_ref.render(verts.numpy(), f.numpy(), None, R.numpy(), T.numpy(), image_size=size, want_image=False)
		err = np.abs(mask.cpu().numpy() - ref['mask'])
		print(f'{n_verts} vertices @{size}: {int(full.sum())} pixels with a full K-buffer, {int(tie.sum())} depth ties at the K-th place; '
			  f'max |mask - oracle| outside them {err[~tie].max():.2e}, inside {err[tie].max() if tie.any() else 0.0:.2e}')
		assert err[~tie].max() < TOL
		assert tie.mean() < 0.15   # (the denser mesh: 7 % of the pixels; 6890 vertices at this size: 1-2 %)


def test_silhouette_backward_with_k_overflow_vs_oracle_autograd():
	"""Dense mesh on a small image (6890-vertex template @64^2: most silhouette pixels see far more than 100 candidates):
	the gradient must flow only through the K nearest candidates of such pixels, as autograd through the oracle's K = 100
	fragments does."""
	from find_amd import synthetic
	size = 64
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(9)
	verts = v[None] * (1 + 0.1 * torch.rand(1, 1, 3, generator=g))
	rng = np.random.RandomState(4)
	R, T = camera_ref.look_at_view_transform(dist=np.full(1, 0.3), elev=rng.uniform(-90, 90, 1), azim=rng.uniform(-90, 90, 1), up=((1, 0, 0),))
	R, T = torch.from_numpy(R), torch.from_numpy(T)
	vg = verts.clone().cuda().requires_grad_(True)
	(mask, _, _, _), params = _render_gpu(vg, f, None, R, T, size, want_image=False)
	rp = render_ref.default_params(size)
	vproj = render_ref.project(rp, verts.numpy(), R.numpy(), T.numpy())
	# Which face is the 100th nearest?  Where the 100th and the 101st candidate of a pixel have depths that agree to rounding, the answer
	# depends on the last bits of the interpolated depth (thousands of quarter-pixel faces per pixel here) and two correct implementations
	# may keep different faces.  Those pixels are found in the oracle's own K = 101 fragments and taken out of the loss on BOTH sides;
	# everywhere else mask and gradient must agree to the north_star tolerance.
	p2f101, z101, _, _ = render_ref.rasterize(vproj, f.numpy(), 1, size, size, 101, rp.sil_blur_radius)
	p2f = np.ascontiguousarray(p2f101[..., :100])
	assert (p2f[..., 99] >= 0).mean() > 0.05  # the K-buffer really is full on a good share of the pixels
	z99, z100 = z101[..., 99].astype(np.float64), z101[..., 100].astype(np.float64)
	DEPTH_TIE = 4e-6   # relative; ~30 ulp of the fp32 depth: the interpolation is a dozen roundings on either side
	tie = (p2f101[..., 100] >= 0) & (z100 - z99 <= DEPTH_TIE * z99)
	tie = torch.from_numpy(tie.reshape(mask.shape))
	print('depth-tie pixels: %d of %d (%d with a full K-buffer)' % (int(tie.sum()), tie.numel(), int((p2f[..., 99] >= 0).sum())))
	assert tie.float().mean().item() < 0.02
	w = (~tie).float()
	gt = torch.rand(mask.shape, generator=torch.Generator().manual_seed(2))
	loss = (((mask - gt.cuda()) ** 2) * w.cuda()).mean()
	loss.backward()
	vr = verts.clone().requires_grad_(True)
	rm = render_ref.torch_mask(rp, vr, f, R, T, torch.from_numpy(p2f).long(), 1)
	dm = (mask.detach().cpu() - rm.detach()).abs()
	print('mask: max diff outside the ties %.2e, inside %.2e' % ((dm * w).max().item(), (dm * (1 - w)).max().item()))
	assert (dm * w).max().item() < TOL
	rl = (((rm - gt) ** 2) * w).mean()
	rl.backward()
	scale = vr.grad.abs().max().item()
	assert scale > 0
	err = (vg.grad.cpu() - vr.grad).abs()
	print('grad: max err %.2e of scale %.2e' % (err.max().item(), scale))
	assert err.max().item() < TOL * scale, (err.max().item(), scale)


def test_forward_edge_cases_vs_oracle():
	"""What PyTorch3D's rasteriser handles at the edges (rasterize_meshes.cu / FootRenderer, renderer.py:208-245): per-mesh ragged
	face lists (-1 padding), a zero-area face, a mesh wholly behind the camera, a view that sees nothing, an image size that is
	not a multiple of the 16-pixel tile.  Same checks as the regular forward test."""
	from find_amd import functional_render as FR
	size = 40
	verts, faces, cols, R, T = _scene(n_meshes=3, rings=7, segs=9, seed=11, n_views=3)
	# (the mesh pushed behind one camera crosses the z-clip plane of another: faces that straddle it are rasterised unclipped by both
	# the oracle and the kernel -- the documented deviation from PyTorch3D -- so the render watchdog is told to let this scene through)
	monkey = FR.FLAG_POLICY
	FR.FLAG_POLICY = 'ignore'
	try:
		_edge_cases_body(size, verts, faces, cols, R, T)
	finally:
		FR.FLAG_POLICY = monkey


def _edge_cases_body(size, verts, faces, cols, R, T):
	F = faces.shape[0]
	fr = faces[None].expand(3, -1, -1).clone()
	fr[1, F - 9:] = -1                      # mesh 1 has 9 faces fewer (ragged batch)
	fr[0, 5] = torch.tensor([3, 3, 7])      # degenerate (zero-area) face
	verts = verts.clone()
	R, T = R.clone(), T.clone()
	# view 2 looks the other way: flip its view-space y and z (row-vector convention X_view = X R + T  ->  R' = R D, T' = T D)
	D = torch.tensor([1.0, -1.0, -1.0])
	R[2] = R[2] * D[None, :]
	T[2] = T[2] * D
	# mesh 2 sits 0.7 m behind the camera of view 0: move it 1 m against the world direction that maps to view 0's z axis
	verts[2] = verts[2] - 1.0 * R[0][:, 2]
	(mask, image, p2f, zbuf), _ = _render_gpu(verts, fr, cols, R, T, size, want_frags=True)
	ref = render_ref.render(verts.numpy(), fr.numpy(), cols.numpy(), R.numpy(), T.numpy(), image_size=size)
	m = mask.cpu().numpy()
	assert np.abs(m - ref['mask']).max() < TOL
	same = (p2f.cpu().numpy() == ref['pix_to_face'])
	_assert_index_mismatches_are_edge_ties(p2f.cpu().numpy(), ref['pix_to_face'], verts.numpy(), fr.numpy(), R.numpy(), T.numpy(), size)
	assert same.mean() > 0.999, same.mean()
	assert np.abs(image.cpu().numpy() - ref['image'])[same].max() < TOL
	assert np.abs(zbuf.cpu().numpy() - ref['zbuf'])[same].max() < 1e-5
	# the scene really contains the cases: an empty view, a mesh behind a camera, covered pixels elsewhere
	assert ref['mask'][0, 2].max() == 0.0 and m[0, 2].max() == 0.0
	assert (p2f.cpu().numpy()[0, 2] == -1).all()
	assert ref['mask'][2, 0].max() == 0.0 and m[2, 0].max() == 0.0
	assert ref['mask'][0, 0].max() > 0.99 and ref['mask'][1, 1].max() > 0.99
	# the padded / degenerate faces are never picked
	ids = p2f.cpu().numpy()
	F3 = fr.shape[1]
	local = np.where(ids >= 0, ids % F3, -1)
	assert not (local[1] >= F - 9).any()
	assert not (local[0] == 5).any()
	# gradients flow through the ragged batch without touching the padding
	vg = verts.clone().cuda().requires_grad_(True)
	(mk, _, _, _), _ = _render_gpu(vg, fr, None, R, T, size, want_image=False)
	mk.sum().backward()
	assert torch.isfinite(vg.grad).all() and vg.grad[0].abs().max() > 0 and vg.grad[2].abs().max() >= 0


def test_toes_view_matches_oracle_and_straddling_faces_fail_loudly():
	"""view_from('toes') (renderer.py:192), the closest camera the reference defines, against the oracle; then a camera pushed INTO the
	mesh: faces straddle the z-clip plane, PyTorch3D would clip them, this rasteriser does not -- the render must fail (checked
	synchronously here; by default the counters are looked at one call later and a bad render WARNS, functional_render.FLAG_POLICY)."""
	from find_amd import functional_render as FR
	from find_amd import synthetic
	from find_amd.renderer import FootRenderer
	v, f = synthetic.template(1002)
	verts = v[None].clone()
	rdr = FootRenderer(image_size=64, device='cuda')
	R, T = rdr.view_from('toes')
	Rn, Tn = camera_ref.look_at_view_transform(dist=0.1, elev=0.0, azim=0.0, at=((0.1, 0, 0),), up=((1, 0, 0),))
	assert np.abs(R.numpy() - Rn).max() < 1e-6 and np.abs(T.numpy() - Tn).max() < 1e-6
	prev, FR.FLAG_POLICY = FR.FLAG_POLICY, 'sync'
	try:
		(mask, _, p2f, zbuf), _ = _render_gpu(verts, f, None, R, T, 64, want_image=False, want_frags=True)
		ref = render_ref.render(verts.numpy(), f.numpy(), None, Rn, Tn, image_size=64, want_image=False)
		assert np.abs(mask.cpu().numpy() - ref['mask']).max() < TOL
		assert zbuf[p2f >= 0].min().item() > 0.059
		# camera 0.02 m from the origin, inside the ellipsoid: its far wall crosses z = 0.01
		Rb, Tb = camera_ref.look_at_view_transform(dist=0.02, elev=0.0, azim=0.0, up=((1, 0, 0),))
		with pytest.raises(RuntimeError, match='straddle the z-clip plane'):
			_render_gpu(verts, f, None, torch.from_numpy(Rb), torch.from_numpy(Tb), 64, want_image=False)
		FR.FLAG_POLICY = 'strict'   # the bad render returns, the NEXT call (or an explicit check) raises
		_render_gpu(verts, f, None, torch.from_numpy(Rb), torch.from_numpy(Tb), 64, want_image=False)
		with pytest.raises(RuntimeError, match='straddle the z-clip plane'):
			FR.check_render_flags(wait=True)
		FR.FLAG_POLICY = 'warn'     # the default: the same look-up warns, with the description of the render, and training goes on
		_render_gpu(verts, f, None, torch.from_numpy(Rb), torch.from_numpy(Tb), 64, want_image=False)
		with pytest.warns(RuntimeWarning, match='1 meshes x 1 views @64x64.*straddle the z-clip plane'):
			FR.check_render_flags(wait=True)
		# the pinned slots are a ring of 64, reused: 200 renders nobody looks at in between neither grow it nor lose a report
		with pytest.warns(RuntimeWarning, match='straddle the z-clip plane'):
			for _ in range(200):
				_render_gpu(verts, f, None, torch.from_numpy(Rb), torch.from_numpy(Tb), 64, want_image=False)
			FR.check_render_flags(wait=True)
		assert FR._slots.shape[0] == 64 and len(FR._free) == 64 and not FR._pending
	finally:
		FR.FLAG_POLICY = prev
		try:
			FR.check_render_flags(wait=True)
		except RuntimeError:
			pass


def test_early_exit_and_list_order_change_nothing():
	"""The rasteriser walks a tile's faces front to back (lists sorted by depth slab) and leaves a pixel -- or the whole tile -- alone once it
	holds its K nearest candidates in front of everything still to come.  With the exit switched off (find_debug_raster_ablate bit 8) and
	with the lists left in face order (bit 16: no slab sort, hence no exit either) the SAME pixels must come out: the same face in front,
	the same K-nearest set behind every mask value (products of the same factors in another order: equal to rounding), the same
	bound for the backward; and likewise when a tile finds no room in the list pool and scans the faces itself (bit 256: a pool of 512 entries) -- on a dense mesh at a small size, where every covered pixel has more than K candidates and ties at the K-th
	depth are common (tie_fix_kernel)."""
	from find_amd import _lib, functional_render as FR, synthetic
	from find_amd.cameras import look_at_view_transform
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(5)
	verts = (v[None] * (1 + 0.1 * torch.rand(2, 1, 3, generator=g))).cuda()
	cols = torch.rand(2, v.shape[0], 3, generator=g).cuda()
	R, T = look_at_view_transform(dist=np.full(3, 0.3), elev=np.array([10.0, -60.0, 85.0]), azim=np.array([20.0, -80.0, 0.0]), up=((1, 0, 0),))
	params = FR.make_params(96)
	out = {}
	for bits in (0, 8, 16, 256):
		_lib.set_tuning('raster_ablate', bits | RASTER_BASE)
		try:
			vg = verts.clone().requires_grad_(True)
			mask, image, p2f, zbuf = FR.render(vg, cols, f.cuda(), R.cuda(), T.cuda(), params, want_image=True, want_frags=True)
			gt = torch.rand(mask.shape, generator=torch.Generator().manual_seed(3)).cuda()
			((mask - gt) ** 2).mean().backward()
			out[bits] = (mask.detach().clone(), image.detach().clone(), p2f.clone(), zbuf.clone(), vg.grad.clone())
		finally:
			_lib.set_tuning('raster_ablate', RASTER_BASE)
	m0, i0, p0, z0, g0 = out[0]
	assert float((m0 > 0.5).float().mean()) > 0.05
	for bits in (8, 16, 256):
		m1, i1, p1, z1, g1 = out[bits]
		assert torch.equal(p0, p1) and torch.equal(z0, z1), bits   # the same face, the same depth (the same rounding) in front of every pixel
		assert (m0 - m1).abs().max().item() < 2e-6, bits
		assert (i0 - i1).abs().max().item() < 1e-5, bits          # (vertex normals are float-atomic sums: two runs of one variant differ as much)
		assert (g0 - g1).abs().max().item() < 1e-4 * g0.abs().max().item(), bits   # the same K nearest reach the backward (measured 2.6e-5: float atomics, product order)


def test_largest_image_size_renders_and_the_next_one_is_refused():
	"""Image sizes go up to 2048 (tile coordinates are packed in 8 bits, render.hip): one mesh, one view at 2048^2 must give the silhouette
	of the 512^2 render -- same covered fraction of the image, soft mask in [0, 1], every covered pixel's face a real face -- with
	gradients flowing; 2049 must be refused before anything is launched."""
	from find_amd import functional_render as FR
	verts, faces, cols, R, T = _scene(n_meshes=1, rings=10, segs=12, seed=21, n_views=1)
	cover = {}
	for size in (512, 2048):
		vg = verts.clone().cuda().requires_grad_(True)
		(mask, image, p2f, zbuf), _ = _render_gpu(vg, faces, cols, R, T, size, want_frags=True)
		assert mask.shape == (1, 1, size, size) and image.shape == (1, 1, size, size, 3)
		assert float(mask.detach().min()) >= 0.0 and float(mask.detach().max()) <= 1.0 and torch.isfinite(image).all()
		ids = p2f.cpu()
		assert int(ids.max()) < faces.shape[0] and int(ids.min()) >= -1
		cover[size] = float((ids >= 0).float().mean())
		mask.sum().backward()
		assert torch.isfinite(vg.grad).all() and float(vg.grad.abs().max()) > 0
	assert abs(cover[2048] - cover[512]) < 0.01 * cover[512], cover
	with pytest.raises(RuntimeError, match='image size out of range'):
		_render_gpu(verts, faces, cols, R, T, 2049)
