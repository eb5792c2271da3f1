"""FIND's 3-D and silhouette losses on the MI355X hot path: host-side mirror of reference src/model/losses.py (same class
names and forward signatures); the arithmetic runs in libfind_hip.so through find_amd.functional.

In scope: TextureLossGTSpace, DisplacementLoss (Chamfer, incl. the z-cut-off variants),
MeshSmoothnessLoss, SilhouetteLoss.  Perceptual / Restyle / Contrastive losses need absent network weights or
submodules and are out of scope."""
import torch

from . import functional as FN
from . import functional_render as FR
from .structures import Meshes, TexturesUV, TexturesVertex

nn = torch.nn


def sample_points_from_meshes(meshes: Meshes, num_samples: int = 10000, return_textures: bool = False, generator=None, draws=None):
	"""pytorch3d.ops.sample_points_from_meshes: faces ~ multinomial(area) with replacement, (w0,w1,w2) = (1-sqrt(u),
	sqrt(u)(1-v), sqrt(u)v).  The uniform draws come from torch's generator on the mesh device; the areas, the face
	choice and the gather / lerp run in find_sample_surface_fwd.  draws=(face_idx (N,S) int, uv (N,S,2)) replays given draws for
	reproducible CPU/GPU comparisons (find_sample_points_fwd)."""
	verts = meshes.verts_padded()
	faces = meshes.faces_shared()
	if faces is None:
		faces = meshes.faces_padded()
	N = verts.shape[0]
	tex = meshes.textures if return_textures else None
	if return_textures and not isinstance(tex, (TexturesUV, TexturesVertex)):
		raise NotImplementedError('return_textures needs TexturesVertex or TexturesUV')
	attr = tex.verts_features_padded()[..., :3].contiguous() if isinstance(tex, TexturesVertex) else None
	if draws is None:
		# one launch for the draws (torch's device generator, where PyTorch3D draws), two for areas -> running sum -> search -> gather
		rnd = torch.rand(N, num_samples, 3, device=verts.device, generator=generator)
		pts, cols, face_idx, uv = FN.sample_surface(verts, faces, rnd, attr)
	else:
		face_idx, uv = draws
		r = FN.sample_points(verts, faces, face_idx, uv, attr)
		pts, cols = r if attr is not None else (r, None)
	if not return_textures:
		return pts
	if isinstance(tex, TexturesUV):
		# PyTorch3D: barycentrics of the sample (w0 = 1 - sqrt(u), w1 = sqrt(u)(1 - v), w2 = sqrt(u) v), then TexturesUV.sample_textures
		su = uv[..., 0].sqrt()
		bary = torch.stack([1.0 - su, su * (1.0 - uv[..., 1]), su * uv[..., 1]], dim=-1)
		return pts, FR.uv_sample(tex.maps_padded(), tex.verts_uvs_padded(), tex.faces_uvs_padded(), face_idx, bary)
	return pts, cols


def _compact_by_mask(points, keep):
	"""Move the kept points of every cloud to the front (stable) and return (padded points, lengths): the padded-tensor form
	of the ragged Pointclouds the reference builds for its z cut-offs (losses.py:69-85)."""
	order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
	return torch.gather(points, 1, order.unsqueeze(-1).expand(-1, -1, 3)), keep.sum(dim=1).to(torch.int32)


def chamfer_distance(x, y, x_lengths=None, y_lengths=None):
	return FN.chamfer_distance(x, y, x_lengths, y_lengths)


class TextureLossGTSpace(nn.Module):
	def forward(self, model, batch: dict, num_samples=1000, shapevec=None, texvec=None, posevec=None, gt_samples=None) -> torch.Tensor:
		"""Sample points + colours on the GT meshes, query the colour field there, masked L2 (reference losses.py:22-57).
		gt_samples (not in the reference): (points, colours) already drawn on batch['mesh'] -- ModelWithLoss draws the GT samples of a step
		before the main pass, beside it (they depend on the batch alone)."""
		mesh_gt = batch['mesh']
		sampled_verts, sampled_gt_colours = gt_samples if gt_samples is not None else sample_points_from_meshes(mesh_gt, num_samples=num_samples, return_textures=True)
		texvec = texvec if texvec is not None else batch.get('texvec', None)
		shapevec = shapevec if shapevec is not None else batch.get('shapevec', None)
		posevec = posevec if posevec is not None else batch.get('posevec', None)
		# (only the colour head is read below: a model that can skip the displacement head does)
		# (... and its weight gradients may trail behind the rest of the backward pass: nothing reads them before the pass ends)
		only_col = dict(want=('col',), defer_wgrad_join=True) if 'want' in getattr(getattr(model.forward, '__code__', None), 'co_varnames', ()) else {}
		res = model(sampled_verts.detach(), texvec=texvec, shapevec=shapevec, posevec=posevec, **only_col)
		# F.mse_loss(reduction='none') * mask, .mean() with mask = any(gt < 1) per point (losses.py:43,55-57), as one kernel
		return FN.masked_mse(res['col'], sampled_gt_colours)


class DisplacementLoss(nn.Module):
	def forward(self, model, res, batch, epoch, num_samples=5000, z_cutoff=None, gt_z_cutoff=None, gt_samples=None):
		"""Chamfer distance between surface samples of the GT and the predicted meshes (reference losses.py:59-90).
		gt_samples (not in the reference): the GT samples, already drawn (see TextureLossGTSpace.forward)."""
		if gt_samples is None:
			gt_samples = sample_points_from_meshes(batch['mesh'], num_samples=num_samples)
		pred_samples = sample_points_from_meshes(res['meshes'], num_samples=num_samples)
		if z_cutoff is not None:
			p, pl = _compact_by_mask(pred_samples, pred_samples[..., 2] <= z_cutoff)
			g, gl = _compact_by_mask(gt_samples, gt_samples[..., 2] <= z_cutoff)
			chamf_loss, _ = chamfer_distance(p, g, pl, gl)
		elif gt_z_cutoff is not None:
			g, gl = _compact_by_mask(gt_samples, gt_samples[..., 2] <= gt_z_cutoff)
			chamf_loss, _ = chamfer_distance(pred_samples, g, None, gl)
		else:
			chamf_loss, _ = chamfer_distance(pred_samples, gt_samples)
		return dict(loss=chamf_loss)


class MeshSmoothnessLoss(nn.Module):
	def forward(self, meshes: Meshes):
		"""0.1 * cotangent-Laplacian smoothing + 10 * edge-length loss (reference losses.py:93-99)."""
		faces = meshes.faces_shared()
		if faces is None:
			raise NotImplementedError('MeshSmoothnessLoss expects meshes sharing one topology (the template), as on the FIND path')
		verts = meshes.verts_padded()
		topo = FN.MeshTopology.get(faces, verts.shape[1])
		return FN.mesh_smoothness_loss(verts, topo, w_edge=10.0, w_lap=0.1)


class SilhouetteLoss(nn.Module):
	def forward(self, pred, gt):
		"""MSE of the soft silhouettes (reference losses.py:122-128: nn.MSELoss), one pass each way in find_image_mse_*."""
		return FN.image_mse(pred, gt)
