"""FIND's 3-D and silhouette losses on the MI355X hot path: host-side mirror of reference src/model/losses.py (same class
names and forward signatures); the arithmetic runs in libfind_hip.so through find_amd.functional.

In scope: TextureLossGTSpace, DisplacementLoss (Chamfer, incl. the z-cut-off variants),
MeshSmoothnessLoss, SilhouetteLoss.  Perceptual / Restyle / Contrastive losses need absent network weights or
submodules and are out of scope."""
import torch
from torch.nn import functional as F

from . import functional as FN
from . import functional_render as FR
from .structures import Meshes, TexturesUV, TexturesVertex

nn = torch.nn


def sample_points_from_meshes(meshes: Meshes, num_samples: int = 10000, return_textures: bool = False, generator=None, draws=None):
	"""pytorch3d.ops.sample_points_from_meshes: faces ~ multinomial(area) with replacement, (w0,w1,w2) = (1-sqrt(u),
	sqrt(u)(1-v), sqrt(u)v).  The random draws use torch's generator on the mesh device (or are passed in
	as draws=(face_idx (N,S) int, uv (N,S,2)) for reproducible CPU/GPU comparisons); the gather/lerp is the HIP kernel."""
	verts = meshes.verts_padded()
	faces = meshes.faces_shared()
	if faces is None:
		faces = meshes.faces_padded()
	N = verts.shape[0]
	if draws is None:
		with torch.no_grad():
			areas = FN.face_areas(verts, faces)
			face_idx = torch.multinomial(areas, num_samples, replacement=True, generator=generator)
			uv = torch.rand(N, num_samples, 2, device=verts.device, generator=generator)
	else:
		face_idx, uv = draws
	if return_textures:
		tex = meshes.textures
		if isinstance(tex, TexturesUV):
			# PyTorch3D: barycentrics of the sample (w0 = 1 - sqrt(u), w1 = sqrt(u)(1 - v), w2 = sqrt(u) v), then TexturesUV.sample_textures
			pts = FN.sample_points(verts, faces, face_idx, uv)
			su = uv[..., 0].sqrt()
			bary = torch.stack([1.0 - su, su * (1.0 - uv[..., 1]), su * uv[..., 1]], dim=-1)
			return pts, FR.uv_sample(tex.maps_padded(), tex.verts_uvs_padded(), tex.faces_uvs_padded(), face_idx, bary)
		if not isinstance(tex, TexturesVertex):
			raise NotImplementedError('return_textures needs TexturesVertex or TexturesUV')
		return FN.sample_points(verts, faces, face_idx, uv, tex.verts_features_padded()[..., :3].contiguous())
	return FN.sample_points(verts, faces, face_idx, uv)


def _compact_by_mask(points, keep):
	"""Move the kept points of every cloud to the front (stable) and return (padded points, lengths): the padded-tensor form
	of the ragged Pointclouds the reference builds for its z cut-offs (losses.py:69-85)."""
	order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
	return torch.gather(points, 1, order.unsqueeze(-1).expand(-1, -1, 3)), keep.sum(dim=1).to(torch.int32)


def chamfer_distance(x, y, x_lengths=None, y_lengths=None):
	return FN.chamfer_distance(x, y, x_lengths, y_lengths)


class TextureLossGTSpace(nn.Module):
	def forward(self, model, batch: dict, num_samples=1000, shapevec=None, texvec=None, posevec=None) -> torch.Tensor:
		"""Sample points + colours on the GT meshes, query the colour field there, masked L2 (reference losses.py:22-57)."""
		mesh_gt = batch['mesh']
		sampled_verts, sampled_gt_colours = sample_points_from_meshes(mesh_gt, num_samples=num_samples, return_textures=True)
		mask = (sampled_gt_colours < 1).any(dim=-1).unsqueeze(-1).expand(-1, -1, 3)
		texvec = texvec if texvec is not None else batch.get('texvec', None)
		shapevec = shapevec if shapevec is not None else batch.get('shapevec', None)
		posevec = posevec if posevec is not None else batch.get('posevec', None)
		res = model(sampled_verts.detach(), texvec=texvec, shapevec=shapevec, posevec=posevec)
		loss = F.mse_loss(res['col'], sampled_gt_colours, reduction='none')
		return (loss * mask).mean()


class DisplacementLoss(nn.Module):
	def forward(self, model, res, batch, epoch, num_samples=5000, z_cutoff=None, gt_z_cutoff=None):
		"""Chamfer distance between surface samples of the GT and the predicted meshes (reference losses.py:59-90)."""
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
		loss_edge, loss_laplacian = FN.mesh_edge_and_laplacian(verts, topo)
		return 0.1 * loss_laplacian + 10 * loss_edge


class SilhouetteLoss(nn.Module):
	def __init__(self):
		super().__init__()
		self.crit = nn.MSELoss()

	def forward(self, pred, gt):
		return self.crit(pred, gt)
