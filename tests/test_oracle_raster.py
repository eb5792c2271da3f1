"""Known-answer tests anchoring the render oracle (oracle/raster_ref.c, render_ref.py, camera_ref.py): SURVEY.md A.2-A.4.
PyTorch3D is absent and the reference has no render fixtures (parity unpinned), so the conventions are pinned
analytically: pixel<->NDC flip, inside/outside signed distances, depth ordering, blur cut-off, silhouette and blend
formulas (the latter also against the in-repo statement of the math, reference src/model/renderer.py:23-72).  CPU only."""
import math

import numpy as np
import torch

from oracle import camera_ref, render_ref

S = camera_ref.fov_scale(60.0)
I3 = np.eye(3, dtype=np.float32)[None]
T0 = np.zeros((1, 3), np.float32)


def ndc_tri(pts_ndc, z=1.0):
	"""World vertices (camera at origin looking down +z, R=I) that project to the given NDC xy at depth z."""
	p = np.asarray(pts_ndc, np.float32)
	zz = np.broadcast_to(np.asarray(z, np.float32), (p.shape[0],))
	return np.concatenate([p * zz[:, None] / S, zz[:, None]], 1)[None]


def test_look_at_view_transform_conventions():
	R, T = camera_ref.look_at_view_transform(dist=0.3, elev=0, azim=0, up=((0, 1, 0),))
	# camera on +z looking at the origin: world origin lands at depth 0.3 in front of the camera
	np.testing.assert_allclose(T[0], [0, 0, 0.3], atol=1e-7)
	np.testing.assert_allclose(R[0], np.diag([-1., 1., -1.]), atol=1e-7)  # x_axis = up x z = -x_world
	p = np.array([[0.1, 0.05, 0.0]], np.float32) @ R[0] + T[0]
	np.testing.assert_allclose(p, [[-0.1, 0.05, 0.3]], atol=1e-7)
	# FIND's up=(1,0,0) with elev=90 puts the camera on +y; degenerate up-vectors take the replacement branch
	R2, T2 = camera_ref.look_at_view_transform(dist=0.35, elev=90, azim=0, up=((1, 0, 0),))
	C = camera_ref.camera_center(R2, T2)
	np.testing.assert_allclose(C[0], [0, 0.35, 0], atol=1e-6)
	np.testing.assert_allclose(R2[0] @ R2[0].T, np.eye(3), atol=1e-6)
	np.testing.assert_allclose(np.linalg.det(R2[0]), 1.0, atol=1e-6)
	# batched sampling as in FootRenderer.sample_views (renderer.py:149-152)
	R3, T3 = camera_ref.look_at_view_transform(dist=np.full(4, 0.3), elev=np.array([-90, -30, 10, 80.]), azim=np.array([-80, 0, 45, 90.]), up=((1, 0, 0),))
	assert R3.shape == (4, 3, 3) and T3.shape == (4, 3)
	for m in range(4):
		np.testing.assert_allclose(R3[m] @ R3[m].T, np.eye(3), atol=1e-5)
		np.testing.assert_allclose(np.linalg.norm(camera_ref.camera_center(R3, T3)[m]), 0.3, atol=1e-6)
		# the object centre always projects to the image centre
		np.testing.assert_allclose((np.zeros(3) @ R3[m] + T3[m])[:2], [0, 0], atol=1e-6)


def test_projection_and_pixel_ndc_flip():
	rp = render_ref.default_params(8)
	verts = ndc_tri([[0.5, -0.25], [0.0, 0.0], [-1.0, 1.0]], z=[1.0, 2.0, 0.5])
	vp = render_ref.project(rp, verts, I3, T0)
	np.testing.assert_allclose(vp[0, :, :2], [[0.5, -0.25], [0, 0], [-1, 1]], atol=1e-6)
	np.testing.assert_allclose(vp[0, :, 2], [1.0, 2.0, 0.5], atol=1e-7)
	# +x is LEFT, +y is UP: a small triangle around NDC (1-1/W, 1-1/H) covers exactly the top-left pixel (0,0)
	W = 8
	c = 1 - 1 / W
	tri = ndc_tri([[c + 0.05, c - 0.05], [c - 0.05, c - 0.05], [c, c + 0.07]])
	p2f, zb, ba, di = render_ref.rasterize(render_ref.project(rp, tri, I3, T0), np.array([[0, 1, 2]]), 1, W, W, 1, 0.0)
	hit = np.argwhere(p2f[0, :, :, 0] >= 0)
	assert hit.tolist() == [[0, 0]]


def test_single_triangle_inside_outside_and_blur():
	W = 16
	rp = render_ref.default_params(W)
	tri = ndc_tri([[0.5, -0.5], [-0.5, -0.5], [0.0, 0.5]], z=[1.0, 1.0, 3.0])
	vp = render_ref.project(rp, tri, I3, T0)
	faces = np.array([[0, 1, 2]])
	blur = 0.01
	p2f, zb, ba, di = render_ref.rasterize(vp, faces, 1, W, W, 4, blur)
	px = lambda i: 1 - (2 * i + 1) / W
	# a pixel strictly inside: negative distance, barycentrics sum to 1, depth from perspective-correct interpolation
	yi, xi = 9, 8
	assert p2f[0, yi, xi, 0] == 0 and p2f[0, yi, xi, 1] == -1
	assert di[0, yi, xi, 0] < 0
	np.testing.assert_allclose(ba[0, yi, xi, 0].sum(), 1.0, atol=1e-6)
	b = ba[0, yi, xi, 0]
	np.testing.assert_allclose(zb[0, yi, xi, 0], b[0] * 1 + b[1] * 1 + b[2] * 3, rtol=1e-6)
	# screen-space (affine) barycentrics recovered from the perspective-correct ones: b_i ~ w_i / z_i  =>  w_i ~ b_i * z_i
	wa = b * np.array([1, 1, 3.]); wa /= wa.sum()
	x_rec = wa @ np.array([0.5, -0.5, 0.0]); y_rec = wa @ np.array([-0.5, -0.5, 0.5])
	np.testing.assert_allclose([x_rec, y_rec], [px(xi), px(yi)], atol=1e-5)
	# inside distance = squared distance to the nearest of the three edges
	def seg_d2(p, a, b):
		a, b, p = np.asarray(a, float), np.asarray(b, float), np.asarray(p, float)
		t = np.clip((b - a) @ (p - a) / ((b - a) @ (b - a)), 0, 1)
		return float(((a + t * (b - a) - p) ** 2).sum())
	P = [px(xi), px(yi)]
	V3 = [[0.5, -0.5], [-0.5, -0.5], [0.0, 0.5]]
	expect = min(seg_d2(P, V3[0], V3[1]), seg_d2(P, V3[0], V3[2]), seg_d2(P, V3[1], V3[2]))
	np.testing.assert_allclose(-di[0, yi, xi, 0], expect, rtol=1e-4)
	# a pixel just below the bottom edge: positive squared distance, kept only while < blur
	rows = [r for r in range(W) if px(r) < -0.5]
	r0 = rows[0]  # first row below the edge: distance |px(r0)+0.5| = 1/16 -> d = 0.0039 < 0.01
	assert p2f[0, r0, 8, 0] == 0
	np.testing.assert_allclose(di[0, r0, 8, 0], (px(r0) + 0.5) ** 2, rtol=1e-4)
	assert p2f[0, rows[1], 8, 0] == -1  # next row: d = (3/16)^2 = 0.035 >= blur
	# clipped barycentrics outside the face are non-negative and renormalised
	assert (ba[0, r0, 8, 0] >= 0).all() and abs(ba[0, r0, 8, 0].sum() - 1) < 1e-6
	# with blur 0 nothing outside survives and K=1 inside pixels agree
	p2f0, zb0, ba0, di0 = render_ref.rasterize(vp, faces, 1, W, W, 1, 0.0)
	assert p2f0[0, r0, 8, 0] == -1 and p2f0[0, yi, xi, 0] == 0


def test_depth_order_topk_and_culling():
	W = 8
	rp = render_ref.default_params(W)
	big = [[0.9, -0.9], [-0.9, -0.9], [0.0, 0.9]]
	v = np.concatenate([ndc_tri(big, z=2.0)[0], ndc_tri(big, z=1.0)[0], ndc_tri(big, z=3.0)[0], ndc_tri(big, z=-1.0)[0]])[None]
	faces = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]])
	vp = render_ref.project(rp, v, I3, T0)
	p2f, zb, ba, di = render_ref.rasterize(vp, faces, 1, W, W, 2, 0.0)
	assert p2f[0, 4, 4].tolist() == [1, 0]      # nearest two, ascending z; face 3 (behind the camera) never appears
	np.testing.assert_allclose(zb[0, 4, 4], [1.0, 2.0], rtol=1e-6)
	p2f3, zb3, _, _ = render_ref.rasterize(vp, faces, 1, W, W, 5, 0.0)
	assert p2f3[0, 4, 4].tolist() == [1, 0, 2, -1, -1]
	# back-facing triangles are kept (cull_backfaces=False in FIND) and a degenerate one is dropped
	flip = np.array([[0, 2, 1], [3, 3, 4]])
	p2ff, _, _, _ = render_ref.rasterize(vp, flip, 1, W, W, 2, 0.0)
	assert p2ff[0, 4, 4].tolist() == [0, -1]


def test_silhouette_formula_and_blend_against_in_repo_math():
	# one fragment at signed distance d: mask = sigmoid(-d/1e-4)
	for d in [-3e-4, -1e-5, 2e-5, 9e-4]:
		m = render_ref.silhouette(np.array([[0, -1, -1]], np.int32), np.array([[d, -1, -1]], np.float32))
		np.testing.assert_allclose(m[0], 1 / (1 + math.exp(d / 1e-4)), rtol=1e-5, atol=1e-7)  # 1-(1-p) in fp32
	# several fragments: 1 - prod(1 - p_k), empty slots ignored (renderer.py:50-54)
	d = np.array([[-2e-4, 5e-5, 3e-4, -1.0]], np.float32)
	f = np.array([[4, 9, 2, -1]], np.int32)
	p = 1 / (1 + np.exp(d[0, :3].astype(np.float64) / 1e-4))
	np.testing.assert_allclose(render_ref.silhouette(f, d)[0], 1 - np.prod(1 - p), rtol=1e-5, atol=1e-7)
	# blur radius used by FootRenderer: log(1/1e-4 - 1) * 1e-4, where the face probability has dropped to 1e-4
	rp = render_ref.default_params(8)
	np.testing.assert_allclose(1 / (1 + math.exp(rp.sil_blur_radius / 1e-4)), 1e-4, rtol=1e-4)
	np.testing.assert_allclose(math.sqrt(rp.sil_blur_radius) * 128, 3.88, atol=0.01)  # ~3.9 px at 256^2


def test_c_rasteriser_agrees_with_torch_restatement():
	"""Fragments recomputed differentiably from pix_to_face equal what the C rasteriser stored."""
	from find_amd import synthetic
	v, f = synthetic.ellipsoid_mesh(10, 14)
	rp = render_ref.default_params(32, faces_per_pixel=100)
	R, T = camera_ref.look_at_view_transform(dist=0.3, elev=[20., -60.], azim=[30., 70.], up=((1, 0, 0),))
	verts = v[None].numpy()
	vp = render_ref.project(rp, verts, R, T)
	for K, blur, clip in [(100, rp.sil_blur_radius, True), (1, 0.0, False)]:
		p2f, zb, ba, di = render_ref.rasterize(vp, f.numpy(), 2, 32, 32, K, blur)
		tp = render_ref.torch_project(torch.from_numpy(verts), torch.from_numpy(R), torch.from_numpy(T))
		np.testing.assert_allclose(tp.numpy(), vp, rtol=1e-5, atol=1e-6)
		pz, bary, dist, valid, _ = render_ref.torch_fragments(tp, f, torch.from_numpy(p2f).long(), 2, 32, 32, clip_bary=clip)
		vm = valid.numpy()
		assert vm.sum() > 100
		np.testing.assert_allclose(pz.numpy()[vm], zb[vm], rtol=1e-4, atol=1e-6)
		np.testing.assert_allclose(dist.numpy()[vm], di[vm], rtol=2e-3, atol=1e-7)
		np.testing.assert_allclose(bary.numpy()[vm], ba[vm], atol=2e-4)
	# the closed surface is covered front and back: interior pixels see >= 2 faces, mask ~ 1 inside, 0 far outside
	p2f, zb, ba, di = render_ref.rasterize(vp, f.numpy(), 2, 32, 32, 100, rp.sil_blur_radius)
	mask = render_ref.silhouette(p2f, di)
	# (interior values are < 1 wherever the pixel centre is within ~0.01 NDC of the covering faces' edges)
	assert mask[0, 16, 16] > 0.9 and mask[0, 0, 0] == 0.0
	assert mask[0, 12:20, 12:20].min() > 0.5
	assert ((p2f[0, 16, 16] >= 0).sum()) >= 2


def test_phong_known_colour_and_background():
	"""A triangle facing the light head-on at the image centre: colour = (0.5 + 0.3*cos) * texel + 0.2*spec^64."""
	W = 8
	rp = render_ref.default_params(W)
	rp.light_pos[:] = [0., 0., -5.]   # light behind the camera, which sits at the origin looking down +z
	tri = ndc_tri([[0.9, -0.9], [-0.9, -0.9], [0.0, 0.9]], z=1.0)
	faces = np.array([[0, 1, 2]])  # orientation such that the area-weighted normal points to -z (towards camera/light)
	col = np.array([[[0.2, 0.4, 0.6]] * 3], np.float32)
	out = render_ref.render(tri, faces, col, I3, T0, image_size=W, rp=rp)
	n = render_ref.vertex_normals(tri, faces)[0, 0]
	np.testing.assert_allclose(n, [0, 0, -1], atol=1e-6)
	img = out['image'][0, 0]
	assert out['pix_to_face'][0, 0, 4, 4] == 0
	# at the centre pixel the surface point is ~ (x,y,1): light dir ~ (0,0,-1) up to the small xy offset
	b = 1 - 9 / W  # ndc of pixel 4
	P = np.array([b / S, b / S, 1.0])
	l = np.array([0, 0, -5.]) - P; l /= np.linalg.norm(l)
	cosang = -l[2]
	vdir = -P / np.linalg.norm(P)
	r = -l + 2 * cosang * np.array([0, 0, -1.])
	spec = 0.2 * max(vdir @ r, 0) ** 64
	expect = (0.5 + 0.3 * cosang) * np.array([0.2, 0.4, 0.6]) + spec
	np.testing.assert_allclose(img[4, 4], expect, rtol=1e-4)
	np.testing.assert_allclose(img[0, 0], [1, 1, 1], atol=1e-7)  # background where no face (renderer.py:118)
	# torch restatement reproduces the C image
	p2f1 = torch.from_numpy(np.where(out['pix_to_face'][0] >= 0, out['pix_to_face'][0], -1)).long().reshape(1, W, W, 1)
	timg = render_ref.torch_phong_image(rp, torch.from_numpy(tri), torch.from_numpy(col), torch.from_numpy(faces), torch.from_numpy(I3), torch.from_numpy(T0), p2f1, 1)
	np.testing.assert_allclose(timg[0, 0].numpy(), img, atol=2e-5)


def test_two_triangle_fragment_table_computed_by_hand():
	"""K = 2 fragment table of one pixel under two stacked triangles, every number derived by hand from SURVEY A.3 (not from the
	oracle's code): a 1x1 image has its pixel centre at NDC (0, 0).
	A (face 0), constant depth 1:  a0 (-.5,-.5)  a1 (.5,-.5)  a2 (0,.5)
	   unnormalised edge functions at the origin: (0.25, 0.25, 0.5) of area 1 -> bary (1/4, 1/4, 1/2); z = 1;
	   distances to the edges: bottom edge y = -.5 -> 0.5^2 = 0.25; side edges 2x + y - .5 = 0 -> (.5/sqrt 5)^2 = 0.05 -> dist = -0.05 (inside).
	B (face 1), depths (1.5, 2.5, 2.0):  b0 (-.6,-.4)  b1 (.6,-.4)  b2 (0,.8): the origin is its centroid -> screen bary (1/3, 1/3, 1/3);
	   perspective correction t = (w0 z1 z2, z0 w1 z2, z0 z1 w2) = (5/3, 1, 5/4), sum 47/12 -> bary (20/47, 12/47, 15/47);
	   z = sum bary_i z_i = (30 + 30 + 30)/47 = 90/47 (the harmonic mean 3 / (1/1.5 + 1/2.5 + 1/2));
	   edges: y = -.4 -> 0.16; side edges 1.2x + .6y - .48 = 0 -> .48^2 / 1.8 = 0.128 -> dist = -0.128.
	Output order: ascending depth -> [A, B]."""
	vp = np.array([[[-.5, -.5, 1.0], [.5, -.5, 1.0], [0.0, .5, 1.0], [-.6, -.4, 1.5], [.6, -.4, 2.5], [0.0, .8, 2.0]]], np.float32)
	faces = np.array([[0, 1, 2], [3, 4, 5]])
	p2f, zb, ba, di = render_ref.rasterize(vp, faces, 1, 1, 1, 2, 0.0)
	assert p2f[0, 0, 0].tolist() == [0, 1]
	np.testing.assert_allclose(zb[0, 0, 0], [1.0, 90.0 / 47.0], rtol=2e-7)
	np.testing.assert_allclose(ba[0, 0, 0, 0], [0.25, 0.25, 0.5], atol=1e-7)
	np.testing.assert_allclose(ba[0, 0, 0, 1], [20.0 / 47.0, 12.0 / 47.0, 15.0 / 47.0], atol=1e-7)
	np.testing.assert_allclose(di[0, 0, 0], [-0.05, -0.128], rtol=1e-6)
	# the same two faces listed the other way round: the table is ordered by depth, not by face index
	p2f_r, zb_r, _, _ = render_ref.rasterize(vp, faces[::-1].copy(), 1, 1, 1, 2, 0.0)
	assert p2f_r[0, 0, 0].tolist() == [1, 0]
	np.testing.assert_allclose(zb_r[0, 0, 0], [1.0, 90.0 / 47.0], rtol=2e-7)
	# with the silhouette pass's clipped barycentrics (blur > 0) an inside pixel keeps the same numbers, and the soft mask of the two
	# fragments is 1 - (1 - sigmoid(.05 / 1e-4)) (1 - sigmoid(.128 / 1e-4)) = 1 to fp32
	p2f_c, zb_c, ba_c, di_c = render_ref.rasterize(vp, faces, 1, 1, 1, 2, 9.21e-4)
	np.testing.assert_allclose(ba_c[0, 0, 0, 1], [20.0 / 47.0, 12.0 / 47.0, 15.0 / 47.0], atol=1e-7)
	assert render_ref.silhouette(p2f_c, di_c)[0, 0, 0] == 1.0


def test_sphere_silhouette_area_is_analytic():
	"""A sphere of radius r seen from distance d through FoVPerspectiveCameras(fov 60) projects to a disc of NDC radius
	rho = s r / sqrt(d^2 - r^2), s = 1 / tan(30 deg) (the tangent cone of the sphere): the pixels covered by the K = 1 / blur 0 pass,
	times the NDC area of a pixel, must be pi rho^2.  Pins the projection scale, the pixel <-> NDC mapping and the coverage test in one
	number.  The SOFT silhouette is wider by the blur margin (every face within sqrt(blur_radius) = 0.03 NDC of a pixel adds to it)."""
	from find_amd import synthetic
	r, d, W = 0.1, 0.3, 128
	v, f = synthetic.ellipsoid_mesh(40, 80, axes=(r, r, r))
	R, T = camera_ref.look_at_view_transform(dist=d, elev=20.0, azim=-35.0, up=((1, 0, 0),))
	rp = render_ref.default_params(W)
	vp = render_ref.project(rp, v[None].numpy(), R, T)
	p2f, _, _, _ = render_ref.rasterize(vp, f.numpy(), 1, W, W, 1, 0.0)
	rho = S * r / math.sqrt(d * d - r * r)
	hard = (p2f[0, :, :, 0] >= 0).astype(np.float64)
	area = float(hard.sum()) * (2.0 / W) ** 2
	assert abs(area / (math.pi * rho * rho) - 1.0) < 0.005, (area, math.pi * rho * rho)
	# the disc is centred: its centroid is the image centre to a fraction of a pixel
	ys, xs = np.mgrid[0:W, 0:W]
	assert abs((hard * xs).sum() / hard.sum() - (W - 1) / 2) < 0.3 and abs((hard * ys).sum() / hard.sum() - (W - 1) / 2) < 0.3
	# soft silhouette: contains the disc, and exceeds it by a ring no wider than the blur margin
	soft = render_ref.render(v[None].numpy(), f.numpy(), None, R, T, image_size=W, want_image=False)['mask'][0, 0]
	assert (soft[hard > 0] > 0.5).all()
	soft_area = float(soft.sum()) * (2.0 / W) ** 2
	ring = 2 * math.pi * rho * math.sqrt(rp.sil_blur_radius)
	assert math.pi * rho * rho < soft_area < math.pi * rho * rho + ring


def test_toes_view_stays_in_front_of_the_clip_plane():
	"""FootRenderer.view_from('toes') (renderer.py:192: dist 0.1, looking at (0.1, 0, 0)) is the closest camera the reference defines:
	a foot-sized template stays 0.06 m in front of it, six times the z-clip distance znear / 2 = 0.01, so no face straddles the plane
	and the cull-only treatment of the clip plane is exact for every FIND camera."""
	from find_amd import synthetic
	v, f = synthetic.template(1002)
	R, T = camera_ref.look_at_view_transform(dist=0.1, elev=0.0, azim=0.0, at=((0.1, 0, 0),), up=((1, 0, 0),))
	rp = render_ref.default_params(32)
	vp = render_ref.project(rp, v[None].numpy(), R, T)
	assert vp[..., 2].min() > 0.059
	p2f, zb, _, _ = render_ref.rasterize(vp, f.numpy(), 1, 32, 32, 1, 0.0)
	assert (p2f >= 0).mean() > 0.3 and zb[p2f >= 0].min() > 0.059


def test_compact_fragment_form_equals_the_dense_form():
	"""The full-size GPU parity tests (tests/test_gpu_fullsize.py) use only the oracle's COMPACT form -- covered pixels only -- of torch_mask /
	torch_phong_image (VERDICT r4 weak 1 iv: nothing asserted that it IS the dense form).  A 32^2 scene of two feet x two views: masks,
	images and every gradient (vertices, colours) must be bit-equal; per-pixel arithmetic is elementwise, so leaving empty pixels out may not
	change one bit of the others, and an empty pixel is exactly background / zero with zero gradient."""
	from find_amd import synthetic
	rp = render_ref.default_params(32)
	v, f = synthetic.ellipsoid_mesh(9, 12)
	g = torch.Generator().manual_seed(2)
	verts = torch.stack([v * (1 + 0.1 * torch.rand(1, 3, generator=g)) for _ in range(2)]).float()
	cols = torch.rand(2, v.shape[0], 3, generator=g)
	R, T = camera_ref.look_at_view_transform(dist=np.full(2, 0.3), elev=np.array([20., -50.]), azim=np.array([30., -70.]), up=((1, 0, 0),))
	Rt, Tt = torch.from_numpy(R), torch.from_numpy(T)
	vp = render_ref.project(rp, verts.numpy(), R, T)
	res = {}
	for K, blur in ((100, rp.sil_blur_radius), (1, 0.0)):
		p2f = torch.from_numpy(render_ref.rasterize(vp, f.numpy(), 2, 32, 32, K, blur)[0]).long()
		assert 0.05 < float((p2f[..., 0] >= 0).float().mean()) < 0.9     # some pixels covered, some empty: both branches of the compaction run
		for compact in (False, True):
			vv, cc = verts.clone().requires_grad_(True), cols.clone().requires_grad_(True)
			if K == 100:
				out = render_ref.torch_mask(rp, vv, f, Rt, Tt, p2f, 2, compact=compact)
				w = torch.rand(out.shape, generator=torch.Generator().manual_seed(5))
				(out * w).sum().backward()
				res[K, compact] = (out.detach(), vv.grad)
			else:
				out = render_ref.torch_phong_image(rp, vv, cc, f, Rt, Tt, p2f, 2, compact=compact)
				w = torch.rand(out.shape, generator=torch.Generator().manual_seed(6))
				(out * w).sum().backward()
				res[K, compact] = (out.detach(), vv.grad, cc.grad)
		# forward: bit-equal (elementwise arithmetic per pixel; an empty pixel is background / zero exactly)
		assert torch.equal(res[K, False][0], res[K, True][0]), K
		# gradients: the same per-fragment terms, scattered to the vertices by index_add over another row set -- torch's CPU scatter sums in
		# an order that depends on the tensor's shape, so fp32 sums agree to rounding (measured 6e-6 of the largest entry) ...
		for a, b in zip(res[K, False][1:], res[K, True][1:]):
			assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (K, float((a - b).abs().max()), float(a.abs().max()))
	# ... and in float64 to 1e-13: the two forms are the same function, not two approximations of it
	p2f = torch.from_numpy(render_ref.rasterize(vp, f.numpy(), 2, 32, 32, 100, rp.sil_blur_radius)[0]).long()
	p2f1 = torch.from_numpy(render_ref.rasterize(vp, f.numpy(), 2, 32, 32, 1, 0.0)[0]).long()
	g64 = {}
	for compact in (False, True):
		vv, cc = verts.double().requires_grad_(True), cols.double().requires_grad_(True)
		m = render_ref.torch_mask(rp, vv, f, Rt.double(), Tt.double(), p2f, 2, compact=compact)
		im = render_ref.torch_phong_image(rp, vv, cc, f, Rt.double(), Tt.double(), p2f1, 2, compact=compact)
		((m ** 2).sum() + (im ** 3).sum()).backward()
		g64[compact] = (m.detach(), im.detach(), vv.grad, cc.grad)
	for a, b in zip(g64[False], g64[True]):
		assert float((a - b).abs().max()) <= 1e-13 * max(1.0, float(a.abs().max()))
	# and the fragments themselves: the compact rows are the dense rows of the covered pixels
	p2f = torch.from_numpy(render_ref.rasterize(vp, f.numpy(), 2, 32, 32, 100, rp.sil_blur_radius)[0]).long()
	vpt = render_ref.torch_project(verts, Rt, Tt, rp.fov_deg)
	pix = render_ref.covered_pixels(p2f)
	dense = render_ref.torch_fragments(vpt, f, p2f, 2, 32, 32, clip_bary=True)
	comp = render_ref.torch_fragments(vpt, f, p2f[pix], 2, 32, 32, clip_bary=True, pixels=pix)
	for a, b in zip(dense[:4], comp[:4]):
		assert torch.equal(a[pix], b)
