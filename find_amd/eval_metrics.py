"""Metric definitions of reference src/eval/eval_3d.py on the HIP path (row a16): keypoint error in mm (eval_3d.py:142,
220-221), Chamfer over 10 000 surface samples reported x1e6 as "um" (eval_3d.py:148-151, 223 -- the reference's
scaling of a m^2 quantity is kept) and the per-foot z <= 0.07 cut-off variant (eval_3d.py:154-161).
Tables, plots, spins and OBJ export of the eval script are out of scope."""
import torch

from . import functional as FN
from .losses import sample_points_from_meshes


def keypoint_error_mm(pred_verts, template_kp_idxs, gt_kps):
	"""pred_verts (N,V,3) registered predictions, template_kp_idxs (K) vertex ids, gt_kps (N,K,3) -> scalar mm."""
	idx = torch.as_tensor(template_kp_idxs, device=pred_verts.device, dtype=torch.long)
	dists = torch.norm(pred_verts[:, idx] - gt_kps, dim=-1)
	return dists.mean() * 1e3


def eval_3d_metrics(pred_meshes, gt_meshes, pred_verts=None, template_kp_idxs=None, gt_kps=None, samples=10000, z_cutoff=0.07,
					draws_gt=None, draws_pred=None):
	"""Returns {'Keypoint (mm)', 'Chamf z-cutoff <z> (um)', 'Chamf (um)'} as 0-d tensors (keypoints only when given)."""
	with torch.no_grad():
		gt_pts = sample_points_from_meshes(gt_meshes, num_samples=samples, draws=draws_gt)
		pred_pts = sample_points_from_meshes(pred_meshes, num_samples=samples, draws=draws_pred)
		chamf, _ = FN.chamfer_distance(gt_pts, pred_pts)
		# per-foot cut-off clouds, each its own batch of one, then the mean over feet (eval_3d.py:154-161)
		keep_p, keep_g = pred_pts[..., 2] <= z_cutoff, gt_pts[..., 2] <= z_cutoff
		order_p = torch.argsort((~keep_p).to(torch.int8), dim=1, stable=True)
		order_g = torch.argsort((~keep_g).to(torch.int8), dim=1, stable=True)
		pp = torch.gather(pred_pts, 1, order_p.unsqueeze(-1).expand(-1, -1, 3))
		gp = torch.gather(gt_pts, 1, order_g.unsqueeze(-1).expand(-1, -1, 3))
		pl, gl = keep_p.sum(1).to(torch.int32), keep_g.sum(1).to(torch.int32)
		# batch-mean of per-cloud (mean_x + mean_y) == mean over feet of single-cloud Chamfer distances
		chamf_cut, _ = FN.chamfer_distance(gp, pp, gl, pl)
		out = {f'Chamf z-cutoff {z_cutoff} (μm)': chamf_cut * 1e6, 'Chamf (μm)': chamf * 1e6}
		if template_kp_idxs is not None:
			out['Keypoint (mm)'] = keypoint_error_mm(pred_verts, template_kp_idxs, gt_kps)
	return out
