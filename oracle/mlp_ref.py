"""Oracle (TEST INFRASTRUCTURE ONLY) for rows a1-a5: torch-CPU fp32, op-for-op what the reference executes.

PINNED for a1-a4 by tests/golden/mlp_main.npz + mlp_variants.npz (tests/test_oracle_mlp.py).
a5 (registration) follows PyTorch3D's euler_angles_to_matrix / Transform3d conventions: PARITY UNPINNED
(dependency absent), anchored by known-answer tests against scipy.spatial.transform.Rotation.

Weights are passed as a reference-format state_dict (SURVEY §8b key names).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def fourier_B(num_input_channels=3, mapping_size=256, scale=10.0):
	"""fourier_feature_transform.py:17-26.  Reseeds the *global* torch RNG to 1 (a reference quirk that makes
	every later nn.Linear init deterministic) and sorts the num_input_channels ROWS of B by L2 norm."""
	torch.manual_seed(1)
	B = torch.randn((num_input_channels, mapping_size)) * scale
	B_sort = sorted(B, key=lambda x: torch.norm(x, p=2))
	return torch.stack(B_sort)


def fourier_features(x, B):
	"""fourier_feature_transform.py:28-55:  [x, sin(2*pi*x@B), cos(2*pi*x@B)] on the last dim."""
	shp = x.shape
	x2 = x.reshape(-1, shp[-1])
	res = x2 @ B
	res = 2 * np.pi * res
	out = torch.cat([x2, torch.sin(res), torch.cos(res)], dim=1)
	return out.reshape(*shp[:-1], out.shape[-1])


def _seq(sd, prefix, x, final_linear):
	"""nn.Sequential / ModuleList of Linear(+ReLU) stored at even indices (model.py:250-257, 351-371)."""
	idxs = sorted({int(k.split('.')[1]) for k in sd if k.startswith(prefix + '.') and k.endswith('.weight')})
	for n, i in enumerate(idxs):
		x = F.linear(x, sd[f'{prefix}.{i}.weight'], sd[f'{prefix}.{i}.bias'])
		if not (final_linear and n == len(idxs) - 1):
			x = torch.relu(x)
	return x


def mlp_forward(sd, B, pos, shapevec=None, texvec=None, posevec=None, use_avg_colour=False, positional_encoding=True):
	"""NeuralDisplacementField.forward, model.py:393-453.  Returns dict(disp, col, trunk)."""
	batch, npts, _ = pos.shape
	if batch == 1 and shapevec is not None:  # model.py:404-406
		batch = shapevec.shape[0]
		pos = pos.expand(batch, -1, -1)
	if shapevec is not None:
		shapevec = shapevec.unsqueeze(1).expand(-1, npts, -1)
	if posevec is not None:
		posevec = posevec.unsqueeze(1).expand(-1, npts, -1)
	if texvec is not None:
		texvec = texvec.unsqueeze(1).expand(-1, npts, -1)

	x = fourier_features(pos, B) if positional_encoding else pos  # model.py:421-422
	x = _seq(sd, 'base', x, final_linear=False)  # model.py:424-426

	disp_input = x
	if shapevec is not None:
		disp_input = torch.cat([disp_input, shapevec], dim=-1)
	if posevec is not None:
		disp_input = torch.cat([disp_input, posevec], dim=-1)
	col_input = x
	if texvec is not None:
		col_input = torch.cat([col_input, texvec], dim=-1)

	disp = _seq(sd, 'mlp_disp', disp_input, final_linear=True)
	col = _seq(sd, 'mlp_col', col_input, final_linear=True)
	disp = 0.1 * torch.tanh(disp)  # model.py:444
	if use_avg_colour:
		col = sd['avg_col'][None, None, :] + 0.5 * (1 + torch.tanh(col))  # model.py:447
	else:
		col = 0.5 * (1 + torch.tanh(col))  # model.py:449
	return dict(disp=disp, col=col, trunk=x)


def euler_angles_to_matrix_xyz(e):
	"""PyTorch3D euler_angles_to_matrix(e, 'XYZ') = Rx(e0) @ Ry(e1) @ Rz(e2) [P3D-recall; SURVEY A.1].
	e: (..., 3) radians -> (..., 3, 3)."""
	c, s = torch.cos(e), torch.sin(e)
	one, zero = torch.ones_like(c[..., 0]), torch.zeros_like(c[..., 0])

	def mat(rows):
		return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)

	Rx = mat([[one, zero, zero], [zero, c[..., 0], -s[..., 0]], [zero, s[..., 0], c[..., 0]]])
	Ry = mat([[c[..., 1], zero, s[..., 1]], [zero, one, zero], [-s[..., 1], zero, c[..., 1]]])
	Rz = mat([[c[..., 2], -s[..., 2], zero], [s[..., 2], c[..., 2], zero], [zero, zero, one]])
	return Rx @ Ry @ Rz


def registration(verts, disp, reg):
	"""get_meshes, model.py:481-491:  Transform3d().scale(S).rotate(R).translate(t).transform_points(v+disp).
	Row-vector convention: X = ((v + disp) * S) @ R + t   [P3D-recall; SURVEY A.1]."""
	if reg is None:
		return verts + disp
	S = reg[..., 6:9]
	t = reg[..., :3]
	R = euler_angles_to_matrix_xyz(reg[..., 3:6])
	p = (verts + disp) * S[:, None, :]
	return torch.bmm(p, R) + t[:, None, :]


def get_meshes_verts(sd, B, template_verts, shapevec, reg, texvec, posevec, use_avg_colour=False):
	"""get_meshes numeric core (model.py:455-504): returns dict(verts, disp, col)."""
	N = 0 if shapevec is None else shapevec.shape[0]
	verts = template_verts.expand(N, -1, -1)
	res = mlp_forward(sd, B, verts, shapevec, texvec, posevec, use_avg_colour)
	X = registration(verts, res['disp'], reg)
	return dict(verts=X, disp=res['disp'], col=res['col'])
