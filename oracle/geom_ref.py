"""Oracle (TEST INFRASTRUCTURE ONLY) for rows a11-a14, a16: surface sampling, Chamfer / KNN, mesh smoothness,
eval metrics.  torch-CPU (differentiable, so autograd supplies the reference gradients).

PARITY UNPINNED: the reference delegates these to PyTorch3D @ 1706eb8216248e54f68cad86f7ea4125c79a3ca4
(requirements_mac_linux.txt:31), which is not vendored / installable here.  Each function restates PyTorch3D's
published algorithm (pytorch3d/ops/sample_points_from_meshes.py, loss/chamfer.py, ops/knn.py,
loss/mesh_edge_loss.py, loss/mesh_laplacian_smoothing.py, ops/laplacian_matrices.py) and is anchored on the
reference's call sites and on known-answer tests (tests/test_oracle_geom.py)."""
import torch


# --------------------------------------------------------------------------------------------- sampling (a12)
def face_areas(verts, faces):
	"""0.5*|(v1-v0)x(v2-v0)|; verts (N,V,3), faces (F,3) or (N,F,3) long -> (N,F)."""
	if faces.dim() == 2:
		faces = faces.unsqueeze(0).expand(verts.shape[0], -1, -1)
	idx = faces.long()
	v0 = torch.gather(verts, 1, idx[..., 0:1].expand(-1, -1, 3))
	v1 = torch.gather(verts, 1, idx[..., 1:2].expand(-1, -1, 3))
	v2 = torch.gather(verts, 1, idx[..., 2:3].expand(-1, -1, 3))
	return 0.5 * torch.linalg.cross(v1 - v0, v2 - v0, dim=-1).norm(dim=-1)


def sample_points(verts, faces, face_idx, uv, attr=None):
	"""sample_points_from_meshes with the random draws given (call sites losses.py:39-41,63,67; eval_3d.py:149-150):
	(w0,w1,w2) = (1-sqrt(u), sqrt(u)(1-v), sqrt(u) v);  p = w0 v0 + w1 v1 + w2 v2.   face_idx (N,S), uv (N,S,2)."""
	N = verts.shape[0]
	if faces.dim() == 2:
		faces = faces.unsqueeze(0).expand(N, -1, -1)
	f = torch.gather(faces.long(), 1, face_idx.long().unsqueeze(-1).expand(-1, -1, 3))  # (N,S,3) vertex ids
	us = uv[..., 0].sqrt()
	w = torch.stack([1.0 - us, us * (1.0 - uv[..., 1]), us * uv[..., 1]], dim=-1)  # (N,S,3)

	def interp(a):
		out = 0
		for c in range(3):
			out = out + w[..., c:c + 1] * torch.gather(a, 1, f[..., c:c + 1].expand(-1, -1, a.shape[-1]))
		return out

	pts = interp(verts)
	return (pts, interp(attr)) if attr is not None else pts


# --------------------------------------------------------------------------------------------- Chamfer (a11, a16)
def knn1(x, y, x_len=None, y_len=None):
	"""knn_points(K=1): squared-L2 distance and index of the nearest y for every x (brute force).  Padding of y is
	ignored; padded x rows get dist 0 / idx -1.  Lowest index wins ties."""
	N, P1, _ = x.shape
	P2 = y.shape[1]
	d = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)  # (N,P1,P2)
	if y_len is not None:
		bad = torch.arange(P2)[None, None, :] >= y_len.view(N, 1, 1)
		d = d.masked_fill(bad, float('inf'))
	dist, idx = d.min(dim=2)
	if y_len is not None:
		# a cloud without targets: knn_points pads what it cannot find with distance 0 / index -1 [P3D-recall]
		none = (y_len.view(N, 1) <= 0).expand(N, P1)
		dist = dist.masked_fill(none, 0.0)
		idx = idx.masked_fill(none, -1)
	if x_len is not None:
		padx = torch.arange(P1)[None, :] >= x_len.view(N, 1)
		dist = dist.masked_fill(padx, 0.0)
		idx = idx.masked_fill(padx, -1)
	return dist, idx


def chamfer_distance(x, y, x_len=None, y_len=None):
	"""pytorch3d.loss.chamfer_distance defaults (batch 'mean', point 'mean', L2, bidirectional) -- call sites
	losses.py:77,85,88 and eval_3d.py:151,159:
	    (sum_n sum_i min_j|x_i-y_j|^2 / P1_n  +  sum_n sum_j min_i|x_i-y_j|^2 / P2_n) / N"""
	N = x.shape[0]
	dx, _ = knn1(x, y, x_len, y_len)
	dy, _ = knn1(y, x, y_len, x_len)
	lx = x_len.clamp(min=1).to(x.dtype) if x_len is not None else torch.full((N,), float(x.shape[1]), dtype=x.dtype)
	ly = y_len.clamp(min=1).to(x.dtype) if y_len is not None else torch.full((N,), float(y.shape[1]), dtype=x.dtype)
	cham_x = (dx.sum(1) / lx).sum() / max(N, 1)
	cham_y = (dy.sum(1) / ly).sum() / max(N, 1)
	return cham_x + cham_y


def keypoint_error_mm(pred_verts, kp_idx, gt_kps):
	"""eval_3d.py:142,220-221: mean Euclidean distance of template keypoint vertices to GT keypoints, in mm."""
	return (pred_verts[:, kp_idx] - gt_kps).norm(dim=-1).mean() * 1e3


# --------------------------------------------------------------------------------------------- smoothness (a14)
def unique_edges(faces):
	"""Unique undirected edges (E,2) of a (F,3) face list, sorted (as Meshes.edges_packed)."""
	f = faces.long()
	e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], dim=0)
	e, _ = e.sort(dim=1)
	return torch.unique(e, dim=0)


def mesh_edge_loss(verts, edges):
	"""mesh_edge_loss(target_length=0): mean over meshes of mean_e (|va-vb|)^2.  verts (N,V,3), edges (E,2) shared."""
	d = verts[:, edges[:, 0]] - verts[:, edges[:, 1]]
	per_edge = d.norm(dim=-1, p=2) ** 2.0
	return (per_edge.sum(1) / edges.shape[0]).sum() / verts.shape[0]


def cot_laplacian_apply(verts1, faces, eps=1e-12):
	"""One mesh: returns (L @ V, rowsum(L)) with L the symmetric cotangent matrix of ops/laplacian_matrices.py
	(L[v1,v2] += cot_a/4 ... with Heron areas clamped at eps), built WITHOUT gradient like the reference."""
	V = verts1.shape[0]
	with torch.no_grad():
		f = faces.long()
		v0, v1, v2 = verts1[f[:, 0]], verts1[f[:, 1]], verts1[f[:, 2]]
		A = (v1 - v2).norm(dim=1)
		B = (v0 - v2).norm(dim=1)
		C = (v0 - v1).norm(dim=1)
		s = 0.5 * (A + B + C)
		area = (s * (s - A) * (s - B) * (s - C)).clamp(min=eps).sqrt()
		A2, B2, C2 = A * A, B * B, C * C
		cot = torch.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], dim=1) / 4.0
		ii = f[:, [1, 2, 0]].reshape(-1)
		jj = f[:, [2, 0, 1]].reshape(-1)
		w = cot.reshape(-1)
		L = torch.zeros(V, V, dtype=verts1.dtype)
		L.index_put_((ii, jj), w, accumulate=True)
		L = L + L.t()
		rowsum = L.sum(dim=1, keepdim=True)
	return L, rowsum


def mesh_laplacian_smoothing_cot(verts, faces):
	"""mesh_laplacian_smoothing(method='cot'): mean over meshes of mean_v | (L V)_v * norm_w_v - V_v |_2,
	norm_w = 1/rowsum where rowsum > 0 (values <= 0 are left as they are, as PyTorch3D does)."""
	N, V, _ = verts.shape
	total = 0
	for n in range(N):
		L, rowsum = cot_laplacian_apply(verts[n], faces)
		norm_w = rowsum.clone()
		pos = norm_w > 0
		norm_w[pos] = 1.0 / norm_w[pos]
		lap = L.mm(verts[n]) * norm_w - verts[n]
		total = total + lap.norm(dim=1).sum() / V
	return total / N


def mesh_smoothness(verts, faces, edges=None):
	"""MeshSmoothnessLoss (losses.py:93-99): 0.1 * laplacian(cot) + 10 * edge."""
	if edges is None:
		edges = unique_edges(faces)
	return 0.1 * mesh_laplacian_smoothing_cot(verts, faces) + 10 * mesh_edge_loss(verts, edges)
