#!/usr/bin/env python3
"""Golden vectors for the COMPOSITION of a training step -- the part of the reference that is plain Python / torch around PyTorch3D calls and
can therefore run in the build container once those calls exist.

Runs ONLY in the build container (needs /root/reference).  Executed for real, as written:

  * src/model/model.py:1001-1163    ModelWithLoss.forward: flag handling, the 3-D supervision switch (restrict_3d_n_train), `is_train`
                                    suffixes, which loss object gets what, weights (`opts.weight_*`), the sum
  * src/model/model.py:455-504      NeuralDisplacementField.get_meshes / get_meshes_from_batch: template extension, MLP query, registration
                                    (Transform3d chain), Meshes update
  * src/model/losses.py:22-99       TextureLossGTSpace, DisplacementLoss (plain, z_cutoff = 0.07 on both clouds, gt_z_cutoff: ragged
                                    Pointclouds), MeshSmoothnessLoss (0.1 laplacian + 10 edge)
  * src/train/opts.py:207-215       Opts.net_train_kwargs for the flags of cfgs/train_3d.yaml's FIND experiment; the default weights (:97-100)
  * src/train/trainer.py:29-46      sample_latent_vectors (tables by index and by label)
  * src/data/dataset.py:88-109      get_pose_code (src/cfg.yaml: POSE_VECTOR)

The `pytorch3d.*` names those lines import are stand-ins BACKED BY THIS REPOSITORY'S ORACLE (PyTorch3D @1706eb82 is absent and not
installable here): Meshes / Pointclouds / TexturesVertex as minimal containers, Transform3d as the row-vector affine chain PyTorch3D
documents, sample_points_from_meshes as multinomial(areas) + uniform (u, v) draws fed to oracle.geom_ref.sample_points (the draws are
RECORDED: they are inputs of the fixture), chamfer_distance / mesh_edge_loss / mesh_laplacian_smoothing -> oracle.geom_ref,
euler_angles_to_matrix -> oracle.mlp_ref.  What this pins is therefore the reference's CONTROL FLOW and ARITHMETIC AROUND those calls --
the rows a5 / a11 / a12 / a14 themselves stay "unpinned" (their arithmetic is the oracle's on both sides of the comparison).

Output (data only): tests/golden/composition.npz -- model state, latent tables, template and scans, per case: flags, the sampler draws in
call order, the loss dict, the total, gradients (every latent table, every MLP weight; strided for the large ones).
Checked by tests/test_oracle_pins.py (the hand-typed oracle compositions of tests/ and bench.py, CPU) and tests/test_gpu_pins.py
(find_amd.ModelWithLoss on the GPU).

Usage:  python tests/golden/make_golden_composition.py"""
import os
import sys
import types
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
GRAD_STRIDE = 17
GRID_TEMPLATE, GRID_SCAN, N_FEET = (9, 14), (8, 12), 3   # lat-long grids (rings, segments): 128- and 98-vertex closed meshes


def install_pytorch3d_stand_ins(record, replay):
	"""sys.modules entries for the pytorch3d names the path imports, backed by oracle/ (see the module docstring).  `record`: list that
	receives (num_samples, face_idx, uv) of every sampler call; while `replay` is non-empty the sampler hands out its entries instead."""
	import torch
	from oracle import camera_ref, geom_ref, mlp_ref

	class TexturesVertex:
		def __init__(self, verts_features):
			self._f = verts_features
		def verts_features_padded(self):
			return self._f
		def expand(self, n, *_):
			return TexturesVertex(self._f.expand(n, -1, -1))
		extend = expand

	class Meshes:
		def __init__(self, verts, faces, textures=None):
			self._v = verts if torch.is_tensor(verts) else torch.stack(list(verts))
			self._f = faces if torch.is_tensor(faces) else torch.stack(list(faces))
			self.textures = textures
		device = property(lambda self: self._v.device)
		def to(self, device):
			return self
		def __len__(self):
			return self._v.shape[0]
		def verts_padded(self):
			return self._v
		def faces_padded(self):
			return self._f
		def update_padded(self, new_verts):
			return Meshes(new_verts, self._f, self.textures)

	class Pointclouds:
		def __init__(self, points):
			self._pts = list(points)
		def padded(self):
			n = max(int(p.shape[0]) for p in self._pts)
			out = torch.zeros(len(self._pts), n, 3, dtype=self._pts[0].dtype)
			out = torch.stack([torch.cat([p, p.new_zeros(n - p.shape[0], 3)]) for p in self._pts])
			return out, torch.tensor([p.shape[0] for p in self._pts], dtype=torch.int64)

	class Transform3d:
		"""points @ M for row vectors, transforms composed left to right (PyTorch3D's convention)."""
		def __init__(self, device=None):
			self._ops = []
		def scale(self, s):
			self._ops.append(lambda x: x * s[:, None, :]); return self
		def rotate(self, R):
			self._ops.append(lambda x: x @ R); return self
		def translate(self, t):
			self._ops.append(lambda x: x + t[:, None, :]); return self
		def transform_points(self, x):
			for op in self._ops:
				x = op(x)
			return x

	def euler_angles_to_matrix(e, convention):
		assert convention == 'XYZ'
		return mlp_ref.euler_angles_to_matrix_xyz(e)

	def sample_points_from_meshes(meshes, num_samples=10000, return_textures=False):
		verts, faces = meshes.verts_padded(), meshes.faces_padded()[0].long()
		was_replay = bool(replay)
		if replay:   # (the float64 yardstick run: the draws of the fp32 run again)
			ns, face_idx, uv, _ = replay.pop(0)
			assert ns == num_samples
			uv = uv.to(verts.dtype)
		else:
			with torch.no_grad():
				areas = geom_ref.face_areas(verts, faces)
				face_idx = torch.multinomial(areas, num_samples, replacement=True)   # PyTorch3D: areas_padded.multinomial(num_samples, replacement=True)
				uv = torch.rand(verts.shape[0], num_samples, 2)                      # _rand_barycentric_coords: u, v ~ U[0, 1)
		if return_textures:
			out = geom_ref.sample_points(verts, faces, face_idx, uv, attr=meshes.textures.verts_features_padded())
		else:
			out = geom_ref.sample_points(verts, faces, face_idx, uv)
		if not was_replay:
			record.append((num_samples, face_idx.clone(), uv.clone(), (out[0] if return_textures else out).detach().clone()))
		return out

	def chamfer_distance(x, y):
		xl = yl = None
		if isinstance(x, Pointclouds):
			x, xl = x.padded()
		if isinstance(y, Pointclouds):
			y, yl = y.padded()
		return geom_ref.chamfer_distance(x, y, xl, yl), None

	def mesh_edge_loss(meshes):
		faces = meshes.faces_padded()[0].long()
		return geom_ref.mesh_edge_loss(meshes.verts_padded(), geom_ref.unique_edges(faces))

	def mesh_laplacian_smoothing(meshes, method='cot'):
		assert method == 'cot'
		return geom_ref.mesh_laplacian_smoothing_cot(meshes.verts_padded(), meshes.faces_padded()[0].long())

	def module(name, **attrs):
		m = MagicMock(name=name)   # whatever else the reference imports from it and never calls on this path
		for k, v in attrs.items():
			setattr(m, k, v)
		sys.modules[name] = m
		return m

	module('pytorch3d')
	module('pytorch3d.structures', Meshes=Meshes, Pointclouds=Pointclouds)
	module('pytorch3d.structures.utils')
	module('pytorch3d.transforms', euler_angles_to_matrix=euler_angles_to_matrix, Transform3d=Transform3d)
	module('pytorch3d.renderer', TexturesVertex=TexturesVertex, look_at_view_transform=camera_ref.look_at_view_transform)   # (FootRenderer.__init__ unpacks one)
	module('pytorch3d.io')
	module('pytorch3d.ops', sample_points_from_meshes=sample_points_from_meshes)
	module('pytorch3d.ops.sample_points_from_meshes')
	module('pytorch3d.loss', chamfer_distance=chamfer_distance, mesh_edge_loss=mesh_edge_loss, mesh_laplacian_smoothing=mesh_laplacian_smoothing)
	ch = types.ModuleType('pytorch3d.loss.chamfer')   # `from pytorch3d.loss.chamfer import *` (losses.py:8) is where Pointclouds comes from
	ch.Pointclouds = Pointclouds
	ch.chamfer_distance = chamfer_distance
	sys.modules['pytorch3d.loss.chamfer'] = ch
	for name in ('pytorch3d.renderer.mesh', 'pytorch3d.renderer.mesh.rasterizer', 'pytorch3d.renderer.mesh.rasterize_meshes', 'pytorch3d.renderer.mesh.shader'):
		module(name)
	return types.SimpleNamespace(Meshes=Meshes, TexturesVertex=TexturesVertex)


def main():
	import numpy as np
	import torch
	torch.set_num_threads(4)
	draws, replay = [], []
	P3D = install_pytorch3d_stand_ins(draws, replay)
	import make_golden_mlp as G
	G.import_reference()   # (the remaining absent third-party modules as MagicMock; ours above are kept: it skips names already in sys.modules)
	import src.model.model as ref_model
	from src.train.opts import Opts
	from src.train.trainer import sample_latent_vectors
	sys.path.insert(0, os.path.join(G.REF, 'src', 'data'))   # (dataset.py imports its sibling init_paths)
	from src.data.dataset import get_pose_code
	from find_amd import synthetic   # (data generator only: template / scan geometry)

	out = {}
	# ------------------------------------------------------------------ Opts: the FIND experiment of cfgs/train_3d.yaml
	opts = Opts()
	for k, v in dict(chamf_loss=True, smooth_loss=True, texture_loss=True, use_pose_code=True, use_latent_labels=True, copy_over_masking=True).items():
		setattr(opts, k, v)
	ntk = opts.net_train_kwargs()
	out['opts/net_train_kwargs_keys'] = np.array(sorted(ntk))
	out['opts/net_train_kwargs_true'] = np.array(sorted(k for k, v in ntk.items() if v))
	for k in ('weight_chamf', 'weight_smooth', 'weight_tex', 'weight_pix', 'weight_sil'):
		out[f'opts/{k}'] = np.float64(getattr(opts, k))
	out['opts/gt_z_cutoff_is_none'] = np.bool_(opts.gt_z_cutoff is None)
	out['opts/num_views'] = np.int64(opts.num_views)

	# ------------------------------------------------------------------ get_pose_code
	# (names of src/cfg.yaml: POSE_VECTOR; 'Strong ' is stripped by the function, two-element entries are -1 / +1)
	cases = [['T-Pose'], ['Dorsiflex'], ['Strong Plantarflex', 'Inversion'], ['Toe Extension', 'Eversion', 'Medial'], ['Standing on Floor', 'Tiptoes', 'Toe Abduction'], []]
	import json
	from src.utils.utils import cfg
	out['pose/lookup_json'] = np.array(json.dumps({str(k): v for k, v in cfg['POSE_VECTOR'].items()}))   # the table of src/cfg.yaml the codes were formed with (an input)
	out['pose/n'] = np.int64(len(cases))
	for i, pl in enumerate(cases):
		out[f'pose/{i}/names'] = np.array(pl, dtype=str) if pl else np.array([], dtype=str)
		out[f'pose/{i}/code'] = np.asarray(get_pose_code(pl), np.float64)

	# ------------------------------------------------------------------ the model, with label-addressed tables (use_latent_labels)
	feet, names, lab = synthetic.scan_labels(N_FEET, n_val=N_FEET)
	torch.manual_seed(5)
	mwl = ref_model.ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=N_FEET, val_size=N_FEET,
								  shapevec_size=100, texvec_size=100, posevec_size=100, template_mesh_loc=None, latent_labels=lab)
	m = mwl.model
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():   # the reference zero-initialises this layer (model.py:516-518): re-initialise so that the head carries signal
		m.mlp_disp[-1].weight.copy_(torch.randn(m.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		m.mlp_disp[-1].bias.copy_(torch.randn(3, generator=g) * 0.01)
		for vecs in (m.latent_vectors_train, m.latent_vectors_val):
			for vec in vecs:
				t = vec.data if hasattr(vec, 'data') and torch.is_tensor(vec.data) else next(vec.parameters())
				if vec.name.startswith('reg'):
					t.copy_(torch.cat([torch.randn(t.shape[0], 3, generator=g) * 0.004, torch.randn(t.shape[0], 3, generator=g) * 0.05, 1 + torch.randn(t.shape[0], 3, generator=g) * 0.03], 1))
				else:
					t.copy_(torch.randn(t.shape, generator=g) * 0.1)
	tv, tf = synthetic.ellipsoid_mesh(*GRID_TEMPLATE)
	m.template_verts = torch.nn.Parameter(tv.float().unsqueeze(0), requires_grad=False)   # (1, V, 3) / (1, F, 3), as model.py:285-286 stores them
	m.template_faces = torch.nn.Parameter(tf.long().unsqueeze(0), requires_grad=False)
	m.template_mesh = P3D.Meshes(verts=m.template_verts, faces=m.template_faces)
	for k, v in m.state_dict().items():
		out['sd/' + k] = v.detach().numpy().copy()
	out['B'] = m.encoder[0]._B.numpy().copy()
	# scans: ellipsoids with scaled axes and low-frequency bumps, smooth per-vertex colours (the recipe of find_amd.synthetic.gt_feet)
	rng = np.random.RandomState(3)
	base, gf = synthetic.ellipsoid_mesh(*GRID_SCAN, axes=(1.0, 1.0, 1.0))
	base = base.numpy()
	gv, gc = [], []
	for _ in range(N_FEET):
		ax = np.array([0.12, 0.045, 0.04]) * rng.uniform(0.9, 1.1, 3)
		r = np.ones(len(base))
		for _k in range(3):
			r = r + (0.003 / 0.04) * np.sin(base @ rng.uniform(1.0, 3.0, 3) + rng.uniform(0, 2 * np.pi))
		gv.append((base * r[:, None] * ax[None]).astype(np.float32))
		gc.append((0.5 + 0.4 * np.sin(base * rng.uniform(2, 6, 3)[None] + rng.uniform(0, 6, 3)[None])).astype(np.float32))
	gv, gc = torch.from_numpy(np.stack(gv)), torch.from_numpy(np.stack(gc))
	gc = gc.clamp(0.05, 1.0)
	gc[:, ::5] = 1.0   # a share of saturated (white) vertices: the texture term masks samples whose colour is white in every channel
	out['gt/verts'], out['gt/faces'], out['gt/colours'] = gv.numpy(), gf.numpy(), gc.numpy()
	for k, v in lab.items():
		out[f'labels/{k}'] = np.array(v, dtype=str)
	out['batch/feet'], out['batch/names'] = np.array(feet, dtype=str), np.array(names, dtype=str)

	feet_val = [n.split('-')[0] for n in lab['pose_val']]
	out['batch/feet_val'], out['batch/names_val'] = np.array(feet_val, dtype=str), np.array(lab['pose_val'], dtype=str)

	def batch_of(idx, val=False, model=None, dtype=torch.float32):
		model = m if model is None else model
		sel = torch.tensor(idx)
		ft, nm = (feet_val, lab['pose_val']) if val else (feet, names)   # (validation scans have their own label sets: model.py latent_labels *_val)
		b = dict(mesh=P3D.Meshes(gv[sel].to(dtype), gf[None].expand(len(idx), -1, -1), P3D.TexturesVertex(gc[sel].to(dtype))), idx=sel,
				 name=[nm[i] for i in idx], shape=[ft[i] for i in idx], tex=[ft[i] for i in idx], pose=[nm[i] for i in idx], reg=[nm[i] for i in idx])
		b.update(sample_latent_vectors(b, model.latent_vectors_val if val else model.latent_vectors_train))
		return b

	# sample_latent_vectors itself: the rows it returns for a batch, by label
	b = batch_of([2, 0])
	for k in ('shapevec_train', 'texvec_train', 'posevec_train', 'reg_train'):
		out[f'slv/{k}'] = b[k].detach().numpy().copy()

	# ------------------------------------------------------------------ forward() under the flag sets the trainer uses
	cases = {
		'net': dict(idx=[0, 1, 2], flags=dict(**opts.net_train_kwargs())),                                        # train.py:211-215
		'reg': dict(idx=[1, 2], flags=dict(chamf=True, smooth=False, gt_z_cutoff=0.01)),                          # train.py:201 with a cut-off set
		'val_zcut': dict(idx=[2, 0], flags=dict(chamf=True, smooth=True, texture=True, is_train=False, use_z_cutoff=True)),   # val_epoch + z cut-off on both clouds
		'no3d': dict(idx=[1], flags=dict(chamf=True, smooth=True, texture=True), opts=dict(restrict_3d_n_train=1)),   # batch idx 1 >= 1: 3-D terms withheld
		'one_term': dict(idx=[0], flags=dict(texture=True)),
	}
	out['cases'] = np.array(sorted(cases))
	for name, c in cases.items():
		for p in mwl.parameters():
			p.grad = None
		saved = {k: getattr(opts, k) for k in c.get('opts', {})}
		for k, v in c.get('opts', {}).items():
			setattr(opts, k, v)
		val = c['flags'].get('is_train', True) is False
		batch = batch_of(c['idx'], val=val)
		# The nearest-neighbour choice of the Chamfer term is a discrete decision: where a query's two nearest targets are equally far to
		# ~1e-6 (the rounding of a squared distance of millimetres between coordinates of decimetres), an ulp in a sample position flips it
		# and moves single gradient entries by 1e-3 of the tensor's maximum -- in the reference as in any port.  The fixture's draws are
		# redrawn until no query of any Chamfer call is closer to a tie than 3e-6 (30 000 queries per call: the smallest gap of a random draw is ~1e-6) (as make_golden_pins.py keeps its points free of ReLU ties).
		for attempt in range(200):
			for p in mwl.parameters():
				p.grad = None
			del draws[:]
			torch.manual_seed(100 + len(name) + 1000 * attempt)
			loss, losses = mwl(batch, 0, opts, **c['flags'])
			gap = 1.0
			if c['flags'].get('chamf') and len(draws) >= 2:
				a, bb = draws[1][3].double(), draws[0][3].double()   # prediction, GT (the order DisplacementLoss draws them in is GT, prediction)
				for q, t in ((a, bb), (bb, a)):
					two = torch.topk(((q[:, :, None, :] - t[:, None, :, :]) ** 2).sum(-1), 2, dim=-1, largest=False).values
					gap = min(gap, float(((two[..., 1] - two[..., 0]) / two[..., 1]).min()))
			if gap > 3e-6:
				break
		else:
			raise RuntimeError(f'case {name}: no tie-free draws found')
		out[f'case/{name}/nn_min_relative_gap'] = np.float64(gap)
		out[f'case/{name}/draw_attempts'] = np.int64(attempt + 1)
		for k, v in saved.items():
			setattr(opts, k, v)
		out[f'case/{name}/idx'] = np.array(c['idx'], np.int64)
		sfx = 'val' if val else 'train'
		for vec in (m.latent_vectors_val if val else m.latent_vectors_train):   # the table rows the batch addressed (LatentVector.__getitem__: labels.index)
			out[f'case/{name}/rows/{vec.name}'] = np.array([vec.labels.index(o) for o in batch[vec.key]], np.int64)
			assert torch.equal(batch[vec.name], (vec.data if torch.is_tensor(getattr(vec, 'data', None)) else next(vec.parameters()))[out[f'case/{name}/rows/{vec.name}']])
		out[f'case/{name}/flags'] = np.array([f'{k}={v}' for k, v in sorted(c['flags'].items())])
		out[f'case/{name}/opts'] = np.array([f'{k}={v}' for k, v in sorted(c.get('opts', {}).items())], dtype=str)
		out[f'case/{name}/loss_keys'] = np.array(list(losses), dtype=str)
		for k, v in losses.items():
			out[f'case/{name}/losses/{k}'] = np.float64(v.item())
		out[f'case/{name}/n_draws'] = np.int64(len(draws))
		for i, (ns, fi, uv, _pts) in enumerate(draws):
			out[f'case/{name}/draw/{i}/face_idx'] = fi.numpy().astype(np.int32)
			out[f'case/{name}/draw/{i}/uv'] = uv.numpy()
		if not torch.is_tensor(loss):
			assert loss == 0 and not losses
			out[f'case/{name}/loss'] = np.float64(0.0)
			out[f'case/{name}/loss_is_python_zero'] = np.bool_(True)
			continue
		out[f'case/{name}/loss'] = np.float64(loss.item())
		loss.backward()
		for k, p in mwl.named_parameters():
			if p.grad is None:
				continue
			gnp = p.grad.detach().numpy()
			out[f'case/{name}/grad/{k}'] = gnp.copy() if gnp.size <= 4096 else gnp.reshape(-1)[::GRAD_STRIDE].copy()
		# The same call in float64 (the reference's code on double parameters and inputs, the same draws): how far the reference's OWN fp32
		# gradients are from the exact ones.  A port cannot be asked to agree with the fp32 reference more closely than that (bias gradients
		# are cancelling sums over thousands of rows); the GPU test compares with the float64 values under max(1e-4, 2 x this distance).
		import copy
		mwl64 = copy.deepcopy(mwl).double()
		for p in mwl64.parameters():
			p.grad = None
		m64 = mwl64.model
		m64.encoder[0]._B = m64.encoder[0]._B.double()
		m64.template_mesh = P3D.Meshes(verts=m64.template_verts, faces=m64.template_faces)
		m64.latent_vectors_train = [getattr(m64, v.name.replace('_train', '')) for v in m.latent_vectors_train]
		m64.latent_vectors_val = [getattr(m64, v.name) for v in m.latent_vectors_val]
		replay.extend(draws)
		for k, v in c.get('opts', {}).items():
			setattr(opts, k, v)
		loss64, losses64 = mwl64(batch_of(c['idx'], val=val, model=m64, dtype=torch.float64), 0, opts, **c['flags'])
		for k, v in saved.items():
			setattr(opts, k, v)
		assert not replay and list(losses64) == list(losses)
		out[f'case/{name}/loss64'] = np.float64(loss64.item())
		loss64.backward()
		p32 = dict(mwl.named_parameters())
		worst = {}
		for k, p in mwl64.named_parameters():
			if p.grad is None:
				continue
			g64 = p.grad.detach().numpy()
			out[f'case/{name}/grad64/{k}'] = g64.copy() if g64.size <= 4096 else g64.reshape(-1)[::GRAD_STRIDE].copy()
			scale = max(1e-3, float(np.abs(g64).max()))
			worst[k] = out[f'case/{name}/ref_fp32_error/{k}'] = np.float64(np.abs(p32[k].grad.detach().numpy().astype(np.float64) - g64).max() / scale)
		print(name, 'loss', float(loss), {k: round(float(v), 6) for k, v in losses.items()}, 'draw calls', [d[0] for d in draws],
			  '| reference fp32 vs float64, worst gradient error of a tensor: %.1e (%s)' % (max(worst.values()), max(worst, key=worst.get)))
	np.savez_compressed(os.path.join(HERE, 'composition.npz'), **out)
	print('composition.npz:', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'composition.npz')) // 1024, 'KB')


if __name__ == '__main__':
	main()
