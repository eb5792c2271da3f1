"""Host-side camera set-up FIND takes from PyTorch3D (tiny 3x3 math, stays on the host as in the reference):
look_at_view_transform as used by FootRenderer.sample_views / linspace_views / view_from (reference
src/model/renderer.py:145-203).  Conventions: SURVEY.md Appendix A.2 -- angles in degrees, camera centre
C = at + dist*(cos e sin a, sin e, cos e cos a), R's columns are the camera axes (x = up x z, y = z x x, z = at - C),
T = -R^T C, row-vector use p_view = p_world @ R + T."""
import math

import numpy as np
import torch


def _normalize(v, eps=1e-5):
	return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), eps)


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, at=((0, 0, 0),), up=((0, 1, 0),)):
	"""Returns R (M,3,3), T (M,3) float32 torch tensors (CPU)."""
	def arr(a):
		if torch.is_tensor(a):
			a = a.detach().cpu().numpy()
		return np.atleast_1d(np.asarray(a, dtype=np.float64))
	dist, elev, azim = arr(dist), arr(elev), arr(azim)
	at = np.atleast_2d(np.asarray(at, dtype=np.float64))
	up = np.atleast_2d(np.asarray(up, dtype=np.float64))
	M = max(len(dist), len(elev), len(azim), len(at), len(up))
	dist, elev, azim = [np.broadcast_to(a, (M,)) for a in (dist, elev, azim)]
	at, up = np.broadcast_to(at, (M, 3)), np.broadcast_to(up, (M, 3))
	if degrees:
		elev, azim = elev * math.pi / 180.0, azim * math.pi / 180.0
	C = np.stack([dist * np.cos(elev) * np.sin(azim), dist * np.sin(elev), dist * np.cos(elev) * np.cos(azim)], axis=1) + at
	z = _normalize(at - C)
	x = _normalize(np.cross(up, z))
	y = _normalize(np.cross(z, x))
	close = np.all(np.isclose(x, 0.0, atol=5e-3), axis=1, keepdims=True)
	if close.any():
		x = np.where(close, _normalize(np.cross(y, z)), x)
	R = np.stack([x, y, z], axis=1).transpose(0, 2, 1)
	T = -np.einsum('mji,mj->mi', R, C)
	return torch.from_numpy(R.astype(np.float32)), torch.from_numpy(T.astype(np.float32))
