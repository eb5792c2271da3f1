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
