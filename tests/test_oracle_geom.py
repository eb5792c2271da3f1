"""Known-answer tests anchoring the PyTorch3D-recall oracles (oracle/geom_ref.py) -- SURVEY.md §8c.
The reference ships no tests or fixtures for these rows (parity unpinned), so each oracle function is checked against
an analytic answer or an independent brute-force formulation.  CPU only."""
import math

import numpy as np
import torch

from oracle import geom_ref as G


def _tetra():
	v = torch.tensor([[1., 1., 1.], [1., -1., -1.], [-1., 1., -1.], [-1., -1., 1.]])
	f = torch.tensor([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]])
	return v, f


def test_face_areas_unit_triangle_and_tetra():
	v = torch.tensor([[[0., 0., 0.], [1., 0., 0.], [0., 2., 0.]]])
	f = torch.tensor([[0, 1, 2]])
	assert abs(G.face_areas(v, f).item() - 1.0) < 1e-7
	tv, tf = _tetra()
	a = G.face_areas(tv[None], tf)
	np.testing.assert_allclose(a.numpy(), np.full((1, 4), math.sqrt(3) / 4 * 8), rtol=1e-6)  # edge 2*sqrt(2)


def test_sample_points_barycentric_formula():
	v = torch.tensor([[[0., 0., 0.], [1., 0., 0.], [0., 1., 0.], [0., 0., 1.]]])
	f = torch.tensor([[0, 1, 2], [1, 2, 3]])
	fi = torch.tensor([[0, 1, 1]])
	uv = torch.tensor([[[0.25, 0.5], [1.0, 0.0], [0.0, 0.7]]])
	p = G.sample_points(v, f, fi, uv)
	# u=.25 -> sqrt=.5: w=(.5,.25,.25) on face 0 -> (.25,.25,0)
	np.testing.assert_allclose(p[0, 0].numpy(), [0.25, 0.25, 0.0], atol=1e-7)
	# u=1,v=0 -> w=(0,1,0): second vertex of face 1 = vertex 2
	np.testing.assert_allclose(p[0, 1].numpy(), [0., 1., 0.], atol=1e-7)
	# u=0 -> w=(1,0,0): first vertex of face 1 = vertex 1
	np.testing.assert_allclose(p[0, 2].numpy(), [1., 0., 0.], atol=1e-7)
	# attributes use the same weights
	col = torch.tensor([[[1., 0, 0], [0, 1., 0], [0, 0, 1.], [1., 1., 1.]]])
	_, c = G.sample_points(v, f, fi, uv, attr=col)
	np.testing.assert_allclose(c[0, 0].numpy(), [0.5, 0.25, 0.25], atol=1e-7)
	# samples lie in the face plane and inside the triangle
	g = torch.Generator().manual_seed(0)
	uv = torch.rand(1, 200, 2, generator=g)
	pts = G.sample_points(v, f, torch.zeros(1, 200, dtype=torch.long), uv)
	assert float(pts[..., 2].abs().max()) == 0.0
	assert bool((pts[..., 0] >= 0).all() and (pts[..., 1] >= 0).all() and (pts[..., 0] + pts[..., 1] <= 1 + 1e-6).all())


def test_uniform_area_sampling_statistics():
	"""sqrt(u) barycentric sampling is uniform over the triangle: the mean tends to the centroid."""
	v = torch.tensor([[[0., 0., 0.], [3., 0., 0.], [0., 3., 0.]]])
	g = torch.Generator().manual_seed(1)
	uv = torch.rand(1, 200000, 2, generator=g)
	pts = G.sample_points(v, torch.tensor([[0, 1, 2]]), torch.zeros(1, 200000, dtype=torch.long), uv)
	np.testing.assert_allclose(pts.mean(1)[0].numpy(), [1.0, 1.0, 0.0], atol=1e-2)


def test_knn_matches_cdist_and_tie_rule():
	g = torch.Generator().manual_seed(2)
	x = torch.randn(3, 70, 3, generator=g)
	y = torch.randn(3, 50, 3, generator=g)
	d, i = G.knn1(x, y)
	cd = torch.cdist(x.double(), y.double()) ** 2
	dm, im = cd.min(dim=2)
	np.testing.assert_allclose(d.numpy(), dm.numpy(), rtol=1e-5, atol=1e-6)
	assert torch.equal(i, im)
	# ties: duplicated target -> lowest index
	y2 = torch.cat([y[:, :1], y[:, :1], y[:, 1:]], dim=1)
	_, i2 = G.knn1(y[:, :1], y2)
	assert int(i2[0, 0]) == 0


def test_chamfer_known_answers():
	g = torch.Generator().manual_seed(3)
	x = torch.randn(2, 100, 3, generator=g)
	assert abs(float(G.chamfer_distance(x, x))) < 1e-12
	# rigid shift smaller than half the point spacing: every NN is the shifted copy -> 2*|t|^2
	grid = torch.stack(torch.meshgrid(torch.arange(5.), torch.arange(5.), torch.arange(4.), indexing='ij'), -1).reshape(1, -1, 3)
	t = torch.tensor([0.1, -0.2, 0.05])
	np.testing.assert_allclose(float(G.chamfer_distance(grid, grid + t)), 2 * float((t ** 2).sum()), rtol=1e-5)
	# batch mean of per-cloud means, ragged lengths ignore padding
	x = torch.tensor([[[0., 0, 0], [1., 0, 0], [9., 9, 9]]])
	y = torch.tensor([[[0., 0, 0.5], [7., 7, 7]]])
	c = G.chamfer_distance(x, y, x_len=torch.tensor([2]), y_len=torch.tensor([1]))
	# x->y: (0.25 + 1.25)/2 ; y->x: 0.25/1
	np.testing.assert_allclose(float(c), 0.75 + 0.25, rtol=1e-6)


def test_chamfer_gradient_formula():
	"""d/dx_i = 2 (x_i - y_nn(i)) / P1 / N  plus the symmetric term from being someone's nearest neighbour."""
	x = torch.tensor([[[0., 0., 0.], [2., 0., 0.]]], requires_grad=True)
	y = torch.tensor([[[0.5, 0., 0.], [2., 1., 0.], [2.2, 3., 0.]]])
	G.chamfer_distance(x, y).backward()
	# x0 -> y0 ; x1 -> y1 ; y0 -> x0, y1 -> x1, y2 -> x1
	gx0 = 2 * (0 - 0.5) / 2 + 2 * (0 - 0.5) / 3
	gx1y = 2 * (0 - 1) / 2 + 2 * (0 - 1) / 3 + 2 * (0 - 3) / 3
	np.testing.assert_allclose(x.grad[0, 0].numpy(), [gx0, 0, 0], rtol=1e-6)
	np.testing.assert_allclose(x.grad[0, 1, 1].item(), gx1y, rtol=1e-6)


def test_edge_loss_unit_cube():
	v = torch.tensor([[x, y, z] for x in (0., 1.) for y in (0., 1.) for z in (0., 1.)])
	f = torch.tensor([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]])
	e = G.unique_edges(f)
	assert e.shape[0] == 18  # 12 cube edges + 6 face diagonals
	loss = G.mesh_edge_loss(v[None], e)
	np.testing.assert_allclose(float(loss), (12 * 1.0 + 6 * 2.0) / 18, rtol=1e-6)
	# batch mean
	loss2 = G.mesh_edge_loss(torch.stack([v, 2 * v]), e)
	np.testing.assert_allclose(float(loss2), (1 + 4) / 2 * (12 + 12) / 18, rtol=1e-6)


def test_cot_laplacian_flat_grid_and_tetra():
	# interior vertices of a planar regular grid: cot-Laplacian residual vanishes (vertex = weighted mean of neighbours)
	n = 5
	xs, ys = torch.meshgrid(torch.arange(n, dtype=torch.float32), torch.arange(n, dtype=torch.float32), indexing='ij')
	v = torch.stack([xs, ys, torch.zeros_like(xs)], -1).reshape(-1, 3)
	f = []
	for i in range(n - 1):
		for j in range(n - 1):
			a, b, c, d = i * n + j, (i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1
			f += [[a, b, c], [a, c, d]]
	f = torch.tensor(f)
	L, rowsum = G.cot_laplacian_apply(v, f)
	lap = L.mm(v) / rowsum - v
	interior = [(i * n + j) for i in range(1, n - 1) for j in range(1, n - 1)]
	assert float(lap[interior].abs().max()) < 1e-5
	# symmetric, zero diagonal; on a regular tetrahedron every off-diagonal is the SUM of the two opposite cotangents
	# (PyTorch3D omits the conventional 1/2; it cancels in the row normalisation): 2 * cot(60 deg)
	tv, tf = _tetra()
	L, rs = G.cot_laplacian_apply(tv, tf)
	assert torch.allclose(L, L.t())
	assert float(L.diag().abs().max()) == 0.0
	np.testing.assert_allclose(L[0, 1].item(), 2 * (1 / math.sqrt(3)), rtol=1e-5)
	# regular tetrahedron centred at the origin: (LV)/rowsum = mean of the other three = -v/3 -> |lap| = 4/3 |v|
	loss = G.mesh_laplacian_smoothing_cot(tv[None], tf)
	np.testing.assert_allclose(float(loss), 4 / 3 * math.sqrt(3), rtol=1e-5)


def test_laplacian_gradient_ignores_weights():
	"""Gradient flows through the V operands only (L and norm_w are built under no_grad)."""
	tv, tf = _tetra()
	v = (tv + 0.1 * torch.randn(4, 3, generator=torch.Generator().manual_seed(0)))[None].requires_grad_(True)
	loss = G.mesh_laplacian_smoothing_cot(v, tf)
	loss.backward()
	L, rs = G.cot_laplacian_apply(v[0].detach(), tf)
	nw = 1.0 / rs
	lap = L.mm(v[0].detach()) * nw - v[0].detach()
	u = lap / lap.norm(dim=1, keepdim=True) / 4
	expect = L.mm(u * nw) - u
	np.testing.assert_allclose(v.grad[0].numpy(), expect.numpy(), rtol=1e-4, atol=1e-6)


def test_keypoint_error_units():
	pv = torch.zeros(2, 5, 3)
	gk = torch.zeros(2, 2, 3)
	gk[..., 0] = 0.004
	assert abs(float(G.keypoint_error_mm(pv, [1, 3], gk)) - 4.0) < 1e-6
