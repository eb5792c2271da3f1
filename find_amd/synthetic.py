"""Seeded synthetic inputs of the FIND hot path: there is no Foot3D dataset or checkpoint in the
build / GPU containers, so benchmarks, smoke and parity tests use these generators.  Host-side numpy/torch only."""
import math

import numpy as np
import torch

TEMPLATE_GRIDS = {1002: (25, 40), 6890: (84, 82), 10002: (100, 100), 50002: (200, 250)}


def ellipsoid_mesh(rings, segs, axes=(0.12, 0.045, 0.04)):
	"""Closed lat-long ellipsoid centred at the origin: V = rings*segs + 2, F = 2*(V - 2).
	Returns verts (V,3) float32, faces (F,3) int64 (outward orientation)."""
	th = (np.arange(1, rings + 1) / (rings + 1)) * math.pi          # polar angle, poles excluded
	ph = (np.arange(segs) / segs) * 2 * math.pi
	T, P = np.meshgrid(th, ph, indexing='ij')
	x = np.sin(T) * np.cos(P)
	y = np.sin(T) * np.sin(P)
	z = np.cos(T)
	grid = np.stack([x, y, z], -1).reshape(-1, 3)
	verts = np.concatenate([[[0, 0, 1.0]], grid, [[0, 0, -1.0]]], 0) * np.asarray(axes)[None]
	top, bot = 0, rings * segs + 1

	def vid(r, s):
		return 1 + r * segs + (s % segs)

	faces = []
	for s in range(segs):
		faces.append([top, vid(0, s), vid(0, s + 1)])
		faces.append([bot, vid(rings - 1, s + 1), vid(rings - 1, s)])
	for r in range(rings - 1):
		for s in range(segs):
			a, b, c, d = vid(r, s), vid(r + 1, s), vid(r + 1, s + 1), vid(r, s + 1)
			faces.append([a, b, c])
			faces.append([a, c, d])
	return torch.from_numpy(verts.astype(np.float32)), torch.tensor(faces, dtype=torch.int64)


# Which triangulation the generators below use.  'latlong' (default; what every round's numbers and every test were taken on): a
# latitude-longitude grid -- its two poles are vertices of valence `segs` (82 on the template, 100 on the GT scans), and on a 256^2 render
# the hundred-odd slivers around a pole put 2 - 3 thousand faces into single 8 x 8 tiles.  'uniform': a Fibonacci sphere triangulated by its
# convex hull (valence 5 - 7 everywhere, the same V and F = 2 V - 4): what a decimated scan or FIND's own template looks like to a rasteriser.
# bench.py reports the render configurations on both: 'uniform' is the record c3 / c4_rank_share, 'latlong' the records *_latlong_stress.
MESH_KIND = 'latlong'
# Vertex / face order of the 'uniform' meshes: 'morton' (default: spatially coherent, below) or 'lattice' (the Fibonacci lattice's own
# spiral order with faces sorted by their first vertex, as round 5 and the first half of round 6 generated them; FIND_UNIFORM_ORDER).
import os as _os
UNIFORM_ORDER = _os.environ.get('FIND_UNIFORM_ORDER', 'morton')


def uniform_sphere_mesh(n_verts):
	"""Unit sphere with n_verts near-uniformly spread vertices (Fibonacci lattice), triangulated by its convex hull: F = 2 V - 4 faces,
	outward orientation.  Returns verts (V,3) float32, faces (F,3) int64."""
	from scipy.spatial import ConvexHull
	i = np.arange(n_verts) + 0.5
	z = 1.0 - 2.0 * i / n_verts
	r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
	ph = i * (math.pi * (3.0 - math.sqrt(5.0)))
	v = np.stack([r * np.cos(ph), r * np.sin(ph), z], -1)
	# Vertices in a spatially coherent order (Morton code of the position, 10 bits per axis), faces by their lowest vertex: neighbours in
	# memory are neighbours in space, as in the meshes modelling tools and scan pipelines write (FIND's template, decimated scans) -- and
	# as in the latitude-longitude grids.  The lattice's own order runs along a spiral: 64 consecutive faces then span a whole ring of the
	# sphere, and a rasteriser's per-run bounding boxes (bin_kernel: runs of 64 faces) cull nothing -- round 6 measured bin_kernel at
	# 1.8 x its latitude-longitude time on lattice-ordered meshes (profiles/r06_raster_pmc_uniform_*.txt, taken before this ordering).
	q = np.clip(((v + 1.0) * 0.5 * 1023.0).astype(np.int64), 0, 1023)
	code = np.zeros(n_verts, np.int64)
	for bit in range(10):
		for ax in range(3):
			code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
	if UNIFORM_ORDER == 'morton':
		v = v[np.argsort(code, kind='stable')]
	f = ConvexHull(v).simplices.astype(np.int64)
	n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
	flip = (n * v[f[:, 0]]).sum(-1) < 0
	f[flip] = f[flip][:, [0, 2, 1]]
	if UNIFORM_ORDER == 'morton':
		k = f.argmin(1)   # rotate every face so that its lowest vertex comes first (orientation unchanged)
		f = np.stack([f[np.arange(len(f)), k], f[np.arange(len(f)), (k + 1) % 3], f[np.arange(len(f)), (k + 2) % 3]], -1)
	f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
	assert f.shape[0] == 2 * n_verts - 4
	return torch.from_numpy(v.astype(np.float32)), torch.from_numpy(f)


def sphere_mesh(n_verts):
	"""Unit sphere of MESH_KIND with n_verts vertices."""
	if MESH_KIND == 'uniform':
		return uniform_sphere_mesh(n_verts)
	rings, segs = TEMPLATE_GRIDS[n_verts]
	return ellipsoid_mesh(rings, segs, axes=(1.0, 1.0, 1.0))


def template(n_verts=6890):
	if MESH_KIND == 'uniform':
		v, f = uniform_sphere_mesh(n_verts)
		return v * torch.tensor([0.12, 0.045, 0.04]), f
	rings, segs = TEMPLATE_GRIDS[n_verts]
	return ellipsoid_mesh(rings, segs)


def make_model(n_verts=6890, train_size=16, val_size=2, device='cuda', latent_size=100, nonzero_disp_head=True, **kw):
	"""NeuralDisplacementField at FIND's training settings (train.py:148-157) with an ellipsoid template.
	The last disp layer is re-drawn N(0, 0.01^2) (generator seed 1234) so the displacement head carries gradient:
	at the reference's zero init (model.py:516-518) most of that head's backward is identically zero."""
	from .model import NeuralDisplacementField
	m = NeuralDisplacementField(template_mesh_loc=None, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True,
								train_size=train_size, val_size=val_size, shapevec_size=latent_size, texvec_size=latent_size,
								posevec_size=latent_size, **kw)
	if nonzero_disp_head:
		g = torch.Generator().manual_seed(1234)
		with torch.no_grad():
			m.mlp_disp[-1].weight.copy_(torch.randn(m.mlp_disp[-1].weight.shape, generator=g) * 0.01)
			m.mlp_disp[-1].bias.copy_(torch.randn(m.mlp_disp[-1].bias.shape, generator=g) * 0.01)
	m = m.to(device)
	v, f = template(n_verts)
	m.set_template(v.to(device), f.to(device))
	return m


def latents(n_feet, latent_size=100, seed=0, device='cuda'):
	"""shape / tex / pose codes ~ N(0, 0.1^2); reg: t~U(-0.01,0.01), euler~U(-0.1,0.1), S~U(0.9,1.1)."""
	g = torch.Generator().manual_seed(seed)
	sv, tv, pv = [(torch.randn(n_feet, latent_size, generator=g) * 0.1) for _ in range(3)]
	reg = torch.cat([torch.rand(n_feet, 3, generator=g) * 0.02 - 0.01, torch.rand(n_feet, 3, generator=g) * 0.2 - 0.1,
					 torch.rand(n_feet, 3, generator=g) * 0.2 + 0.9], dim=1)
	return dict(shapevec=sv.to(device), texvec=tv.to(device), posevec=pv.to(device), reg=reg.to(device))


def gt_feet(n_feet, n_verts=10002, seed=0, device='cuda'):
	"""Per-foot GT scans: ellipsoid with axes scaled U(0.9,1.1) plus three low-frequency sinusoidal bumps (3 mm),
	per-vertex RGB.  Returns verts (N,V,3), faces (F,3) int64 (shared topology), colours (N,V,3)."""
	rng = np.random.RandomState(seed)
	base, faces = sphere_mesh(n_verts)
	base = base.numpy()
	verts, cols = [], []
	for _ in range(n_feet):
		ax = np.array([0.12, 0.045, 0.04]) * rng.uniform(0.9, 1.1, 3)
		r = np.ones(len(base))
		for _k in range(3):
			w = rng.uniform(1.0, 3.0, 3)
			ph = rng.uniform(0, 2 * np.pi)
			r = r + (0.003 / 0.04) * np.sin(base @ w + ph)
		v = base * r[:, None] * ax[None]
		verts.append(v.astype(np.float32))
		c = 0.5 + 0.4 * np.sin(base * rng.uniform(2, 6, 3)[None] + rng.uniform(0, 6, 3)[None])
		cols.append(c.astype(np.float32))
	return (torch.from_numpy(np.stack(verts)).to(device), faces.to(device), torch.from_numpy(np.stack(cols)).to(device))


def surface_draws(n_meshes, n_samples, n_faces, seed=0, device='cuda', areas=None):
	"""Pre-drawn sampler inputs (face index, u, v): face ~ multinomial(areas) if given else uniform."""
	g = torch.Generator().manual_seed(seed)
	if areas is not None:
		face = torch.multinomial(areas.detach().cpu().float(), n_samples, replacement=True, generator=g)
	else:
		face = torch.randint(0, n_faces, (n_meshes, n_samples), generator=g)
	uv = torch.rand(n_meshes, n_samples, 2, generator=g)
	return face.to(torch.int32).to(device), uv.to(device)


def scan_labels(n_items, scans_per_foot=2, n_val=2):
	"""Label strings as Foot3DDataset.get_keys / get_all_keys produce them (reference src/data/dataset.py:176-198): every scan is
	'<foot>-<scan>'; shape / tex codes are shared by the scans of one foot, pose / reg codes are unique per scan.
	Returns (foot id per item, scan name per item, latent_labels dict for the model incl. n_val validation scans of the feet 9000, 9001, ...:
	labels['pose_val'] lists them, scan i belongs to foot 9000 + i // scans_per_foot)."""
	feet = [f'{i // scans_per_foot:04d}' for i in range(n_items)]
	names = [f'{feet[i]}-{chr(65 + i % scans_per_foot)}' for i in range(n_items)]
	uniq = list(dict.fromkeys(feet))
	feet_val = [f'{9000 + i // scans_per_foot:04d}' for i in range(n_val)]
	names_val = [f'{feet_val[i]}-{chr(65 + i % scans_per_foot)}' for i in range(n_val)]
	uniq_val = list(dict.fromkeys(feet_val))
	labels = dict(shape=list(uniq), tex=list(uniq), pose=list(names), reg=list(names),
				  shape_val=list(uniq_val), tex_val=list(uniq_val), pose_val=list(names_val), reg_val=list(names_val))
	return feet, names, labels
