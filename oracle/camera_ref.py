"""Oracle (TEST INFRASTRUCTURE ONLY) for the host-side camera setup FIND takes from PyTorch3D (row a7):
look_at_view_transform and the FoV perspective constants.  PARITY UNPINNED (PyTorch3D absent); follows
pytorch3d/renderer/cameras.py as recalled; anchored by tests/test_oracle_raster.py."""
import math

import numpy as np


def _normalize(v, eps=1e-5):
	n = np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), eps)
	return v / n


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True, at=((0, 0, 0),), up=((0, 1, 0),)):
	"""R (M,3,3), T (M,3) float32 in PyTorch3D's row-vector convention: p_view = p_world @ R + T."""
	dist, elev, azim = [np.atleast_1d(np.asarray(a, dtype=np.float64)) for a in (dist, elev, azim)]
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
	R = np.stack([x, y, z], axis=1).transpose(0, 2, 1)  # columns are the camera axes
	T = -np.einsum('mji,mj->mi', R, C)                  # -R^T C
	return R.astype(np.float32), T.astype(np.float32)


def camera_center(R, T):
	"""World-space camera centres: C = -T @ R^T."""
	return -np.einsum('mj,mij->mi', T, R).astype(np.float32)


def fov_scale(fov_deg=60.0):
	return 1.0 / math.tan(math.radians(fov_deg) / 2.0)
