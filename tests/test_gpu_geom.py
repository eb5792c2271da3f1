"""GPU parity of the geometry kernels (C-ABI: find_face_areas, find_sample_points_*, find_nn_*, find_smooth_*) against
oracle/geom_ref.py on the same seeded inputs.  Index outputs bit-exact; floats within 1e-4 (observed ~1e-7)."""
import numpy as np
import pytest
import torch

from oracle import geom_ref as G

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _mesh(n_verts=1002, n=3, seed=0):
	from find_amd import synthetic
	v, f = synthetic.template(n_verts)
	g = torch.Generator().manual_seed(seed)
	verts = v[None] + 0.003 * torch.randn(n, v.shape[0], 3, generator=g)
	return verts, f


def test_face_areas_and_sampling_vs_oracle():
	from find_amd import functional as FN
	from find_amd import synthetic
	verts, faces = _mesh()
	col = torch.rand(verts.shape, generator=torch.Generator().manual_seed(1))
	areas = FN.face_areas(verts.cuda(), faces.cuda())
	ref = G.face_areas(verts, faces)
	assert (areas.cpu() - ref).abs().max().item() < 1e-8
	fi, uv = synthetic.surface_draws(3, 5000, faces.shape[0], seed=2, device='cpu', areas=ref)
	vg = verts.clone().cuda().requires_grad_(True)
	pts, cs = FN.sample_points(vg, faces.cuda(), fi.cuda(), uv.cuda(), col.cuda())
	vr = verts.clone().requires_grad_(True)
	rp, rc = G.sample_points(vr, faces, fi, uv, attr=col)
	assert (pts.detach().cpu() - rp.detach()).abs().max().item() < 1e-6
	assert (cs.cpu() - rc).abs().max().item() < 1e-6
	w = torch.randn(pts.shape, generator=torch.Generator().manual_seed(3))
	(pts * w.cuda()).sum().backward()
	(rp * w).sum().backward()
	assert (vg.grad.cpu() - vr.grad).abs().max().item() < 1e-4 * max(1.0, vr.grad.abs().max().item())
	# padded (ragged) faces get zero area
	fpad = torch.cat([faces, torch.full((7, 3), -1, dtype=faces.dtype)])[None].expand(3, -1, -1).contiguous()
	a2 = FN.face_areas(verts.cuda(), fpad.cuda())
	assert float(a2[:, -7:].abs().max()) == 0.0 and (a2[:, :-7].cpu() - ref).abs().max().item() < 1e-8


@pytest.mark.parametrize('shape', [(3, 700, 513), (1, 1, 1), (2, 5000, 5000), (2, 1025, 2049)])
def test_nn_vs_oracle(shape):
	from find_amd import functional as FN
	n, p1, p2 = shape
	g = torch.Generator().manual_seed(p1 + p2)
	x = torch.randn(n, p1, 3, generator=g) * 0.05
	y = torch.randn(n, p2, 3, generator=g) * 0.05
	d, i = FN.knn1(x.cuda(), y.cuda())
	rd, ri = G.knn1(x, y)
	assert torch.equal(i.cpu().long(), ri), 'nearest-neighbour indices must be bit-exact'
	assert (d.cpu() - rd).abs().max().item() < 1e-7


def test_nn_ragged_and_ties():
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(11)
	x = torch.randn(3, 300, 3, generator=g)
	y = torch.randn(3, 200, 3, generator=g)
	xl = torch.tensor([300, 17, 1])
	yl = torch.tensor([5, 200, 64])
	d, i = FN.knn1(x.cuda(), y.cuda(), xl.cuda(), yl.cuda())
	rd, ri = G.knn1(x, y, xl, yl)
	assert torch.equal(i.cpu().long(), ri)
	assert (d.cpu() - rd).abs().max().item() < 1e-6
	# duplicated targets: lowest index wins
	y2 = y.clone()
	y2[:, 100] = y2[:, 3]
	_, i2 = FN.knn1(y2[:, 3:4].contiguous().cuda(), y2.cuda())
	assert int(i2[0, 0]) == 3


def test_chamfer_with_an_empty_cloud():
	"""A cloud that the z-cutoff empties on one or both sides (losses.py:69-85 builds ragged point sets): no query, no target -- its terms are
	0 (lengths clamp to 1 in the means), the other clouds' terms and gradients are unaffected, nothing is NaN."""
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(21)
	x = torch.randn(3, 64, 3, generator=g) * 0.05
	y = torch.randn(3, 80, 3, generator=g) * 0.05
	for xl, yl in [(torch.tensor([64, 0, 10]), torch.tensor([80, 33, 0])), (torch.tensor([0, 0, 0]), torch.tensor([80, 0, 5]))]:
		xg = x.clone().cuda().requires_grad_(True)
		yg = y.clone().cuda().requires_grad_(True)
		loss, _ = FN.chamfer_distance(xg, yg, xl.cuda(), yl.cuda())
		loss.backward()
		xr = x.clone().requires_grad_(True)
		yr = y.clone().requires_grad_(True)
		ref = G.chamfer_distance(xr, yr, xl, yl)
		assert torch.isfinite(ref) and abs(loss.item() - ref.item()) < 1e-6 * max(1e-3, abs(ref.item()))
		if ref.requires_grad and ref.grad_fn is not None:
			ref.backward()
		for got, want in ((xg.grad, xr.grad), (yg.grad, yr.grad)):
			want = torch.zeros_like(x if got.shape == x.shape else y) if want is None else want
			assert torch.isfinite(got).all()
			assert (got.cpu() - want).abs().max().item() < 1e-7
		d, i = FN.knn1(x.cuda(), y.cuda(), xl.cuda(), yl.cuda())
		rd, ri = G.knn1(x, y, xl, yl)
		assert torch.equal(i.cpu().long(), ri) and (d.cpu() - rd).abs().max().item() < 1e-6


def test_chamfer_forward_backward_vs_oracle():
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(5)
	for xl, yl in [(None, None), (torch.tensor([900, 31]), torch.tensor([1000, 500]))]:
		x = (torch.randn(2, 1000, 3, generator=g) * 0.05)
		y = (torch.randn(2, 1200, 3, generator=g) * 0.05)
		xg = x.clone().cuda().requires_grad_(True)
		yg = y.clone().cuda().requires_grad_(True)
		loss, none = FN.chamfer_distance(xg, yg, None if xl is None else xl.cuda(), None if yl is None else yl.cuda())
		assert none is None
		loss.backward()
		xr = x.clone().requires_grad_(True)
		yr = y.clone().requires_grad_(True)
		ref = G.chamfer_distance(xr, yr, xl, yl)
		ref.backward()
		assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
		assert (xg.grad.cpu() - xr.grad).abs().max().item() < 1e-7
		assert (yg.grad.cpu() - yr.grad).abs().max().item() < 1e-7


def test_chamfer_properties_eval_size():
	"""eval_3d-size clouds (10 000 samples): self-distance is exactly 0 and a small rigid shift gives 2|t|^2."""
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(6)
	x = (torch.rand(4, 10000, 3, generator=g)).cuda()
	assert FN.chamfer_distance(x, x)[0].item() == 0.0
	d, i = FN.knn1(x, x)
	assert torch.equal(i.long().cpu(), torch.arange(10000)[None].expand(4, -1))
	t = torch.tensor([1e-4, -2e-4, 5e-5]).cuda()
	c = FN.chamfer_distance(x, x + t)[0].item()
	assert abs(c - 2 * float((t ** 2).sum())) < 1e-3 * 2 * float((t ** 2).sum())


def test_smoothness_forward_backward_vs_oracle():
	from find_amd import functional as FN
	verts, faces = _mesh(1002, n=3, seed=4)
	topo = FN.MeshTopology.get(faces.cuda(), verts.shape[1])
	edges = G.unique_edges(faces)
	assert topo.n_edges == edges.shape[0] and torch.equal(topo.edges.cpu().long(), edges)
	vg = verts.clone().cuda().requires_grad_(True)
	le, ll = FN.mesh_edge_and_laplacian(vg, topo)
	vr = verts.clone().requires_grad_(True)
	re, rl = G.mesh_edge_loss(vr, edges), G.mesh_laplacian_smoothing_cot(vr, faces)
	assert abs(le.item() - re.item()) < 1e-6 * max(1e-3, abs(re.item())) + 1e-9
	assert abs(ll.item() - rl.item()) < 2e-5 * max(1e-3, abs(rl.item()))
	(0.1 * ll + 10 * le).backward()
	(0.1 * rl + 10 * re).backward()
	scale = max(1e-6, vr.grad.abs().max().item())
	assert (vg.grad.cpu() - vr.grad).abs().max().item() < 2e-4 * scale
	# gradient of each term separately
	for we, wl in [(1.0, 0.0), (0.0, 1.0)]:
		vg2 = verts.clone().cuda().requires_grad_(True)
		a, b = FN.mesh_edge_and_laplacian(vg2, topo)
		(we * a + wl * b).backward()
		vr2 = verts.clone().requires_grad_(True)
		(we * G.mesh_edge_loss(vr2, edges) + wl * G.mesh_laplacian_smoothing_cot(vr2, faces)).backward()
		s = max(1e-6, vr2.grad.abs().max().item())
		assert (vg2.grad.cpu() - vr2.grad).abs().max().item() < 2e-4 * s


def test_smoothness_full_template_properties():
	"""6890-vertex template, 16 meshes: a uniformly scaled copy scales the edge loss by s^2 and the Laplacian term by s;
	translation leaves both unchanged."""
	from find_amd import functional as FN
	verts, faces = _mesh(6890, n=16, seed=8)
	topo = FN.MeshTopology.get(faces.cuda(), verts.shape[1])
	v = verts.cuda()
	e1, l1 = FN.mesh_edge_and_laplacian(v, topo)
	e2, l2 = FN.mesh_edge_and_laplacian(v * 1.5, topo)
	e3, l3 = FN.mesh_edge_and_laplacian(v + torch.tensor([0.1, -0.2, 0.3], device='cuda'), topo)
	assert abs(e2.item() / e1.item() - 2.25) < 1e-4
	assert abs(l2.item() / l1.item() - 1.5) < 5e-3  # not exact: the Heron-area clamp (1e-12) is active on the thin pole triangles
	assert abs(e3.item() - e1.item()) < 1e-4 * e1.item()
	assert abs(l3.item() - l1.item()) < 5e-3 * l1.item()


# ------------------------------------------------------------------------------------------------ fused loss-side entry points
def test_nn_packed_four_queries_per_lane():
	"""Launches with >= 4096 blocks take four queries per lane (two packed pairs): same indices, bit for bit."""
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(21)
	x = torch.randn(16, 33000, 3, generator=g) * 0.05
	y = torch.randn(16, 40, 3, generator=g) * 0.05
	d, i = FN.knn1(x.cuda(), y.cuda())
	rd, ri = G.knn1(x, y)
	assert torch.equal(i.cpu().long(), ri)
	assert (d.cpu() - rd).abs().max().item() < 1e-7


def test_sample_surface_keeps_the_area_sums_of_meshes_without_gradient():
	"""A mesh that carries no gradient (a GT scan) is sampled from its KEPT running area sum on later calls (find_sample_surface_again;
	functional._AREA_SUMS): same samples as a fresh call for the same draws, other draws work, an in-place change of the vertices drops the
	entry (the samples follow the new vertices), a mesh with gradient never enters the cache."""
	from find_amd import functional as FN
	verts, faces = _mesh(1002, n=3, seed=21)
	v, f = verts.cuda(), faces.cuda()
	rnd = torch.rand(3, 4000, 3, generator=torch.Generator().manual_seed(22)).cuda()
	rnd2 = torch.rand(3, 700, 3, generator=torch.Generator().manual_seed(23)).cuda()
	FN._AREA_SUMS.clear()
	a = FN.sample_surface(v, f, rnd)
	assert len(FN._AREA_SUMS) == 1
	b = FN.sample_surface(v, f, rnd)            # from the kept sums
	assert len(FN._AREA_SUMS) == 1
	assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
	c = FN.sample_surface(v, f, rnd2)           # other draws, another sample count, on another stream
	side = torch.cuda.Stream()
	side.wait_stream(torch.cuda.current_stream())
	with torch.cuda.stream(side):
		c2 = FN.sample_surface(v, f, rnd2)
	torch.cuda.current_stream().wait_stream(side)
	assert torch.equal(c[0], c2[0]) and torch.equal(c[2], c2[2])
	prev = FN.CACHE_AREA_SUMS
	try:
		FN.CACHE_AREA_SUMS = False
		c3 = FN.sample_surface(v, f, rnd2)
	finally:
		FN.CACHE_AREA_SUMS = prev
	assert torch.equal(c[0], c3[0]) and torch.equal(c[2], c3[2])
	v[0, :, 0] *= 3.0                           # in place: the version counter moves, the entry no longer matches
	d = FN.sample_surface(v, f, rnd)
	try:
		FN.CACHE_AREA_SUMS = False
		d2 = FN.sample_surface(v, f, rnd)
	finally:
		FN.CACHE_AREA_SUMS = prev
	assert torch.equal(d[0], d2[0]) and torch.equal(d[2], d2[2]) and not torch.equal(d[2][0], a[2][0])
	n = len(FN._AREA_SUMS)
	vg = v.clone().requires_grad_(True)
	FN.sample_surface(vg, f, rnd)
	assert len(FN._AREA_SUMS) == n


def test_sample_surface_faces_follow_the_area_distribution():
	"""find_sample_surface_fwd: the face of every sample is the one the float64 running sum of the oracle's areas assigns to the
	sample's draw (mismatches only where the draw sits within rounding of a boundary), the points are the oracle's for those faces,
	padded / zero-area faces are never chosen, and face frequencies follow the areas."""
	from find_amd import functional as FN
	verts, faces = _mesh(1002, n=3, seed=12)
	col = torch.rand(verts.shape, generator=torch.Generator().manual_seed(13))
	F = faces.shape[0]
	fpad = torch.cat([faces, torch.full((5, 3), -1, dtype=faces.dtype)])[None].expand(3, -1, -1).contiguous()
	rnd = torch.rand(3, 5000, 3, generator=torch.Generator().manual_seed(14))
	rnd[0, :4, 0] = torch.tensor([0.0, 1.0 - 2 ** -24, 0.5, 1e-9])   # the ends of the range
	areas = G.face_areas(verts.double(), faces)
	cdf = areas.cumsum(1)
	for fc in (faces, fpad):
		vg = verts.clone().cuda().requires_grad_(True)
		pts, cs, fi, uv = FN.sample_surface(vg, fc.cuda(), rnd.cuda(), col.cuda())
		assert fi.dtype == torch.int32 and int(fi.min()) >= 0 and int(fi.max()) < F, 'padded faces must never be chosen'
		assert torch.equal(uv.cpu(), rnd[..., 1:])
		r = rnd[..., 0].double() * cdf[:, -1:]
		want = torch.searchsorted(cdf, r, right=True).clamp(max=F - 1)
		got = fi.cpu().long()
		bad = got != want
		assert bad.float().mean().item() < 5e-3
		if bad.any():   # a mismatch must be a draw within fp32 rounding of the boundary between the two faces
			b = torch.minimum(got, want)[bad]
			m = bad.nonzero()[:, 0]
			assert ((got - want)[bad].abs() == 1).all()
			assert ((r[bad] - cdf[m, b]).abs() < 2e-6 * cdf[m, -1]).all()
		vr = verts.clone().requires_grad_(True)
		rp, rc = G.sample_points(vr, faces, got, uv.cpu(), attr=col)
		assert (pts.detach().cpu() - rp.detach()).abs().max().item() < 1e-6
		assert (cs.cpu() - rc).abs().max().item() < 1e-6
		w = torch.randn(pts.shape, generator=torch.Generator().manual_seed(15))
		(pts * w.cuda()).sum().backward()
		(rp * w).sum().backward()
		assert (vg.grad.cpu() - vr.grad).abs().max().item() < 1e-4 * max(1.0, vr.grad.abs().max().item())
	# frequencies: 400 000 samples on one mesh, every face within 6 sigma of its binomial expectation
	S = 400000
	big = torch.rand(1, S, 3, generator=torch.Generator().manual_seed(16)).cuda()
	_, _, fi, _ = FN.sample_surface(verts[:1].cuda(), faces.cuda(), big)
	cnt = torch.bincount(fi[0].long().cpu(), minlength=F).double()
	p = (areas[0] / areas[0].sum())
	z = (cnt - S * p) / (S * p * (1 - p)).sqrt()
	assert z.abs().max().item() < 6.0, z.abs().max().item()
	# a face of zero area between others is skipped; a mesh of zero total area falls back to face 0
	vz = verts[:1].clone()
	vz[0, faces[10]] = vz[0, faces[10, 0]].clone()
	az = G.face_areas(vz.double(), faces)[0]
	dead = (az == 0).nonzero()[:, 0]
	_, _, fi, _ = FN.sample_surface(vz.cuda(), faces.cuda(), big[:, :50000].contiguous())
	assert not torch.isin(fi[0].long().cpu(), dead).any()
	_, _, fi, _ = FN.sample_surface(torch.zeros(1, 1002, 3).cuda(), faces.cuda(), big[:, :100].contiguous())
	assert int(fi.abs().max()) == 0


@pytest.mark.parametrize('shape', [(1, 5000, 5000), (1, 777, 5000), (16, 5000, 5000), (2, 300, 9)])
def test_chamfer_fused_vs_oracle(shape):
	"""find_chamfer_fwd / find_chamfer_bwd at the training sizes: batch 1 splits every cloud's targets over eight blocks (merged
	with 64-bit atomic minima), batch 16 does not; loss and both gradients against the oracle."""
	from find_amd import functional as FN
	n, p1, p2 = shape
	g = torch.Generator().manual_seed(p1 * 7 + p2 + n)
	x = torch.randn(n, p1, 3, generator=g) * 0.05
	y = torch.randn(n, p2, 3, generator=g) * 0.05
	chk = min(n, 2)   # the oracle materialises (n, p1, p2)
	xg = x.clone().cuda().requires_grad_(True)
	yg = y.clone().cuda().requires_grad_(True)
	loss, _ = FN.chamfer_distance(xg, yg)
	(loss * 3.0).backward()
	xr = x[:chk].clone().requires_grad_(True)
	yr = y[:chk].clone().requires_grad_(True)
	ref = G.chamfer_distance(xr, yr)
	if chk == n:
		assert abs(loss.item() - ref.item()) < 1e-6 * max(1e-3, abs(ref.item()))
	(ref * 3.0 * chk / n).backward()
	assert (xg.grad[:chk].cpu() - xr.grad).abs().max().item() < 1e-6 * max(1e-3, xr.grad.abs().max().item()) + 1e-9
	assert (yg.grad[:chk].cpu() - yr.grad).abs().max().item() < 1e-6 * max(1e-3, yr.grad.abs().max().item()) + 1e-9
	# only one side needs a gradient (the GT samples never do)
	xg2 = x.clone().cuda().requires_grad_(True)
	FN.chamfer_distance(xg2, y.cuda())[0].backward()
	assert (xg2.grad * 3.0 - xg.grad).abs().max().item() < 1e-6 * max(1e-3, xg.grad.abs().max().item()) + 1e-9


def test_chamfer_fused_ties_go_to_the_lowest_index():
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(31)
	x = torch.randn(1, 400, 3, generator=g)
	y = torch.randn(1, 4000, 3, generator=g)
	y[0, 3001] = y[0, 17]    # duplicates in different target splits and different groups of four
	y[0, 18] = y[0, 17]
	x[0, 0] = y[0, 17] + 1e-4
	yg = y.clone().cuda().requires_grad_(True)
	FN.chamfer_distance(x.cuda(), yg)[0].backward()
	yr = y.clone().requires_grad_(True)
	G.chamfer_distance(x, yr).backward()
	assert (yg.grad.cpu() - yr.grad).abs().max().item() < 1e-7
	# x_0's pull lands on target 17 alone (the direction y -> x touches every y row, so compare with the oracle above and check
	# that the three duplicates did not receive the same gradient)
	assert (yg.grad[0, 17] - yg.grad[0, 18]).abs().max().item() > 1e-9


def test_masked_mse_vs_torch():
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(41)
	pred = torch.rand(4, 1000, 3, generator=g)
	gt = torch.rand(4, 1000, 3, generator=g) * 1.3
	gt[0, :100] = 1.25          # saturated in every channel: masked out
	gt[1, :50, 0] = 1.0          # exactly 1 in one channel only: still inside
	pg = pred.clone().cuda().requires_grad_(True)
	loss = FN.masked_mse(pg, gt.cuda())
	(loss * 2.5).backward()
	pr = pred.clone().requires_grad_(True)
	mask = (gt < 1).any(dim=-1).unsqueeze(-1).expand(-1, -1, 3)
	ref = (torch.nn.functional.mse_loss(pr, gt, reduction='none') * mask).mean()
	(ref * 2.5).backward()
	assert abs(loss.item() - ref.item()) < 1e-6 * abs(ref.item())
	assert (pg.grad.cpu() - pr.grad).abs().max().item() < 1e-6 * pr.grad.abs().max().item()
	assert float(pg.grad[0, :100].abs().max()) == 0.0
	# sizes that leave points past the last group of four, and tensors that do not start on a 16-byte boundary (the scalar path)
	for n_pts, shift in ((1003, 0), (7, 0), (1, 0), (1001, 1), (16000, 0)):
		pred = torch.rand(n_pts + shift, 3, generator=g)
		gt = torch.rand(n_pts + shift, 3, generator=g) * 1.3
		gt[::7] = 1.25
		loss = FN.masked_mse(pred.cuda()[shift:][None], gt.cuda()[shift:][None])
		mask = (gt[shift:] < 1).any(dim=-1, keepdim=True).expand(-1, 3)
		ref = (torch.nn.functional.mse_loss(pred[shift:], gt[shift:], reduction='none') * mask).double().mean()
		assert abs(loss.item() - ref.item()) < 2e-6 * abs(ref.item()) + 1e-12, (n_pts, shift, loss.item(), ref.item())


def test_smoothness_loss_scalar_vs_oracle():
	from find_amd import functional as FN
	verts, faces = _mesh(1002, n=3, seed=44)
	topo = FN.MeshTopology.get(faces.cuda(), verts.shape[1])
	edges = G.unique_edges(faces)
	vg = verts.clone().cuda().requires_grad_(True)
	loss = FN.mesh_smoothness_loss(vg, topo, w_edge=10.0, w_lap=0.1)
	(loss * 1000.0).backward()
	vr = verts.clone().requires_grad_(True)
	ref = 0.1 * G.mesh_laplacian_smoothing_cot(vr, faces) + 10 * G.mesh_edge_loss(vr, edges)
	(ref * 1000.0).backward()
	assert abs(loss.item() - ref.item()) < 2e-5 * abs(ref.item())
	assert (vg.grad.cpu() - vr.grad).abs().max().item() < 2e-4 * vr.grad.abs().max().item()
	e, l = FN.mesh_edge_and_laplacian(verts.cuda(), topo)
	assert abs(loss.item() - (10.0 * e.item() + 0.1 * l.item())) < 1e-6 * abs(loss.item())


@pytest.mark.parametrize('shape', [(2, 3, 40, 40), (1, 1, 7, 5), (3, 2, 129, 65)])
def test_image_mse_vs_torch(shape):
	"""find_image_mse_*: MSE(image * mask, gt image * gt mask) (pixel loss, model.py:1101-1105) and MSE(mask, gt mask) (silhouette loss),
	values and both gradients against the torch expression on the host."""
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(sum(shape))
	img, gimg = torch.rand(*shape, 3, generator=g), torch.rand(*shape, 3, generator=g)
	m, gm = torch.rand(*shape, generator=g), (torch.rand(*shape, generator=g) > 0.4).float()
	a, am = img.clone().cuda().requires_grad_(True), m.clone().cuda().requires_grad_(True)
	loss = FN.image_mse(a, gimg.cuda(), am, gm.cuda())
	(loss * 3.0).backward()
	ar, amr = img.clone().requires_grad_(True), m.clone().requires_grad_(True)
	ref = torch.nn.functional.mse_loss(ar * amr.unsqueeze(-1), gimg * gm.unsqueeze(-1))
	(ref * 3.0).backward()
	assert abs(loss.item() - ref.item()) < 1e-6 * max(1e-3, abs(ref.item()))
	assert (a.grad.cpu() - ar.grad).abs().max().item() < 1e-6 * max(1e-6, ar.grad.abs().max().item()) + 1e-12
	assert (am.grad.cpu() - amr.grad).abs().max().item() < 1e-5 * max(1e-6, amr.grad.abs().max().item()) + 1e-12
	# silhouettes: no masks, one channel
	s = m.clone().cuda().requires_grad_(True)
	ls = FN.image_mse(s, gm.cuda())
	ls.backward()
	sr = m.clone().requires_grad_(True)
	rs = torch.nn.functional.mse_loss(sr, gm)
	rs.backward()
	assert abs(ls.item() - rs.item()) < 1e-6 * max(1e-3, abs(rs.item()))
	assert (s.grad.cpu() - sr.grad).abs().max().item() < 1e-6 * max(1e-6, sr.grad.abs().max().item()) + 1e-12
	# only the mask needs a gradient; a shape mismatch raises
	am2 = m.clone().cuda().requires_grad_(True)
	FN.image_mse(img.cuda(), gimg.cuda(), am2, gm.cuda()).backward()
	assert (am2.grad * 3.0 - am.grad).abs().max().item() < 1e-5 * max(1e-6, am.grad.abs().max().item()) + 1e-12
	with pytest.raises(RuntimeError):
		FN.image_mse(img.cuda(), gimg.cuda()[:, :, :-1])


def test_chamfer_through_the_grid_equals_brute_force_bit_for_bit():
	"""Large clouds (the evaluation's 10 000 samples) go through a uniform grid over the targets (nn_grid_*_kernel) instead of all pairs;
	here the grid is switched on for every size (bit 1024 of the profiling switch) and off (bit 512).  The answer must be
	the brute-force kernel's to the bit -- loss and both gradients -- on inputs chosen to stress the search: surface-like clouds (the FIND
	case), a Gaussian blob, everything in one cell, duplicates (ties: the lowest index wins), queries far outside the targets' bounding box
	(the cubes grow to the whole grid), ragged lengths incl. an empty cloud, and the eval size."""
	from find_amd import _lib
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(77)

	def sphere(n, p, r=0.1, noise=1e-3):
		v = torch.randn(n, p, 3, generator=g)
		return v / v.norm(dim=-1, keepdim=True) * r + noise * torch.randn(n, p, 3, generator=g)

	cases = []
	cases.append(('surface', sphere(3, 5000), sphere(3, 5000), None, None))
	cases.append(('blob', torch.randn(2, 4000, 3, generator=g) * 0.05, torch.randn(2, 3000, 3, generator=g) * 0.05, None, None))
	y1 = torch.randn(1, 2500, 3, generator=g) * 1e-6 + 0.3   # all targets in one cell of their own tiny box; queries elsewhere
	cases.append(('one cell', torch.randn(1, 2500, 3, generator=g), y1, None, None))
	xd, yd = sphere(1, 3000), sphere(1, 3000)
	yd[0, 100:200] = yd[0, 0:100]          # duplicated targets
	xd[0, :50] = yd[0, 100:150]            # queries exactly on duplicated targets (distance 0 to two indices)
	cases.append(('duplicates', xd, yd, None, None))
	xo = sphere(1, 2100)
	xo[0, :64] = xo[0, :64] * 50.0 + 3.0   # queries far outside the targets' box
	cases.append(('outliers', xo, sphere(1, 2100), None, None))
	cases.append(('ragged', sphere(3, 4096), sphere(3, 2048), torch.tensor([4096, 0, 2500], dtype=torch.int64), torch.tensor([2048, 1000, 0], dtype=torch.int64)))
	cases.append(('eval size', sphere(1, 10000), sphere(1, 10000, noise=2e-3), None, None))
	# clouds far from the origin against their extent (un-normalised scans placed a long way off): one ulp of the box origin is
	# a good share of a cell there, and the gap test must not let a cube stop before a nearer target outside it
	off = torch.tensor([50.0, -120.0, 7.0])
	cases.append(('far from the origin', sphere(2, 9000) + off, sphere(2, 9000, noise=2e-3) + off, None, None))
	cases.append(('far from the origin, flat', sphere(1, 6000, r=0.02) * torch.tensor([1.0, 1.0, 0.01]) + 300.0, sphere(1, 6000, r=0.02) * torch.tensor([1.0, 1.0, 0.01]) + 300.0, None, None))
	for name, x, y, xl, yl in cases:
		out = []
		for brute in (False, True):
			_lib.set_tuning('raster_ablate', 512 if brute else 1024)
			try:
				xg, yg = x.clone().cuda().requires_grad_(True), y.clone().cuda().requires_grad_(True)
				loss, _ = FN.chamfer_distance(xg, yg, None if xl is None else xl.cuda(), None if yl is None else yl.cuda())
				loss.backward()
				torch.cuda.synchronize()
				out.append((loss.detach().clone(), xg.grad.clone(), yg.grad.clone()))
			finally:
				_lib.set_tuning('raster_ablate', 0)
		(l0, gx0, gy0), (l1, gx1, gy1) = out
		assert torch.equal(l0, l1), (name, l0.item(), l1.item())
		# (a cloud's gradient holds a float-atomic scatter -- its role as the other direction's target --: sums of the same addends in
		#  another order.  A different neighbour anywhere, also between duplicates, moves a whole addend to another row.
		#  The bound is that of a float sum in another order -- up to a few hundred addends with mixed signs land on one row in the
		#  'one cell' and 'outliers' cases (2 500 queries scattered onto near-coincident targets): 3e-5 of the largest entry; at 1e-6 the
		#  'one cell' case failed one run in five, between two runs of the SAME kernel -- while one misplaced addend is 1e-2 or more.)
		assert (gx0 - gx1).abs().max().item() <= 3e-5 * max(gx1.abs().max().item(), 1e-12), name
		assert (gy0 - gy1).abs().max().item() <= 3e-5 * max(gy1.abs().max().item(), 1e-12), name
		assert torch.isfinite(l0).all()
