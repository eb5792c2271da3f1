"""TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by find_amd/).

CPU restatement of how the reference COMPOSES a 3-D-loss training step out of the pieces the other oracle modules restate:
ModelWithLoss.forward (reference src/model/model.py:1001-1163) for the terms chamf / smooth / texture, with NeuralDisplacementField.
get_meshes_from_batch (model.py:455-504) in front of it and DisplacementLoss / MeshSmoothnessLoss / TextureLossGTSpace (src/model/losses.py:
22-99) inside it.  Every sampler call takes its draws as an argument (PyTorch3D draws them itself; they are inputs of the comparison).

PINNED: tests/golden/composition.npz holds what the reference's own code returns for five flag sets on a small seeded batch -- executed in
the build container with the `pytorch3d.*` names it imports backed by the oracle modules (tests/golden/make_golden_composition.py) --
and tests/test_oracle_pins.py holds this function to it: flag handling, which rows / suffixes a term reads, the z cut-offs' ragged clouds,
the weights and the sum are the reference's.  (The arithmetic INSIDE the PyTorch3D calls -- rows a5 / a11 / a12 / a14 -- is the oracle's on
both sides of that comparison and stays unpinned: oracle/geom_ref.py says so.)"""
import torch

from . import geom_ref, mlp_ref

# src/train/opts.py:97-100
DEFAULT_WEIGHTS = dict(loss_chamf=10000.0, loss_smooth=1000.0, loss_tex=1.0)


def _compact(points, keep):
	"""Ragged clouds as the reference builds them (losses.py:69-85: Pointclouds of the kept points of every cloud, in order): padded + lengths."""
	n = int(keep.sum(dim=1).max())
	out = points.new_zeros(points.shape[0], max(n, 1), 3)
	for b in range(points.shape[0]):
		p = points[b][keep[b]]
		out[b, :p.shape[0]] = p
	return out, keep.sum(dim=1)


def train3d_losses(sd, B, template_verts, template_faces, lat, gt_verts, gt_faces, gt_cols, draws, chamf=True, smooth=True, texture=True,
				   use_z_cutoff=False, gt_z_cutoff=None, supervise_3d=True, weights=DEFAULT_WEIGHTS, per_foot=False):
	"""(total, {key: weighted term}) of one forward() call.
	sd: state dict (tensors; requires_grad where gradients are wanted); lat: dict shapevec / texvec / posevec / reg -> rows of THIS batch;
	draws: dict 'gt' / 'pred' (5000 samples, DisplacementLoss: GT first, losses.py:63,67) and 'tex' (1000, TextureLossGTSpace, losses.py:39) ->
	(face_idx, uv), or for 'pred' a function of the predicted vertices that returns them; per_foot: Chamfer / smoothness foot by foot (the batched ops need tens of GB at 16 x 6890) -- the batch mean of equal
	per-cloud terms is their mean.  total == 0 (python int) with an empty dict when no term is evaluated, as sum({}.values()) (model.py:1159)."""
	res = mlp_ref.get_meshes_verts(sd, B, template_verts, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])   # model.py:1009
	tf = template_faces.long()
	raw = {}
	if supervise_3d:   # model.py:1019-1030: restrict_3d_n_train / restrict_3d_train_key withhold all three
		if chamf:      # model.py:1033-1036 -> losses.py:59-90
			gt_s = geom_ref.sample_points(gt_verts, gt_faces, *draws['gt'])
			pd = draws['pred']
			if callable(pd):   # (a sampler that looks at the predicted surface, as PyTorch3D's does: face ~ multinomial(areas of the PREDICTION))
				pd = pd(res['verts'].detach())
			pr_s = geom_ref.sample_points(res['verts'], tf, *pd)
			n = gt_verts.shape[0]
			if use_z_cutoff:   # z_cutoff = 0.07 on BOTH clouds (model.py:1034, losses.py:69-77)
				p, pl = _compact(pr_s, pr_s[..., 2] <= 0.07)
				g, gl = _compact(gt_s, gt_s[..., 2] <= 0.07)
				raw['loss_chamf'] = geom_ref.chamfer_distance(p, g, pl, gl)
			elif gt_z_cutoff is not None:   # the GT cloud only (losses.py:79-85)
				g, gl = _compact(gt_s, gt_s[..., 2] <= gt_z_cutoff)
				raw['loss_chamf'] = geom_ref.chamfer_distance(pr_s, g, None, gl)
			elif per_foot:
				raw['loss_chamf'] = sum(geom_ref.chamfer_distance(pr_s[i:i + 1], gt_s[i:i + 1]) for i in range(n)) / n
			else:
				raw['loss_chamf'] = geom_ref.chamfer_distance(pr_s, gt_s)
		if smooth:     # model.py:1038-1039 -> losses.py:93-99 (0.1 laplacian + 10 edge inside geom_ref.mesh_smoothness)
			if per_foot:
				n = res['verts'].shape[0]
				edges = geom_ref.unique_edges(tf)
				raw['loss_smooth'] = sum(geom_ref.mesh_smoothness(res['verts'][i:i + 1], tf, edges) for i in range(n)) / n
			else:
				raw['loss_smooth'] = geom_ref.mesh_smoothness(res['verts'], tf)
		if texture:    # model.py:1041-1046 -> losses.py:22-57: colour field at samples of the GT surface, squared error where the GT colour is not white
			tx_p, tx_c = geom_ref.sample_points(gt_verts, gt_faces, *draws['tex'], attr=gt_cols)
			col = mlp_ref.mlp_forward(sd, B, tx_p, lat['shapevec'], lat['texvec'], lat['posevec'])['col']
			mask = (tx_c < 1).any(dim=-1, keepdim=True).expand(-1, -1, 3)
			raw['loss_tex'] = (torch.nn.functional.mse_loss(col, tx_c, reduction='none') * mask).mean()
	losses = {k: v * weights[k] for k, v in raw.items()}   # model.py:1157-1158
	return sum(losses.values()), losses                    # model.py:1159
