"""torch.autograd bindings of the HIP hot path (thin: argument checks, workspace allocation, C-ABI call).

Every function requires CUDA(HIP) fp32 tensors and raises otherwise -- there is no CPU or eager fallback."""
import ctypes

import torch

from . import _lib
from ._lib import MlpGrads, MlpParams, check, current_stream, ptr


def _require_gpu(*tensors):
	for t in tensors:
		if t is None:
			continue
		if not t.is_cuda:
			raise RuntimeError('find_amd: the HIP path needs tensors on a ROCm device (got a CPU tensor); '
							   'there is no CPU fallback -- use oracle/ only for testing')
		if t.dtype != torch.float32 and t.is_floating_point():
			raise RuntimeError(f'find_amd: fp32 tensors required, got {t.dtype}')


def _c(t):
	return None if t is None else t.contiguous()


def _ws(nbytes, device):
	return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------------- MLP
class MLPSpec:
	"""Static description of the network (layer counts / sizes) shared by forward and backward."""

	def __init__(self, n_trunk, n_disp, n_col, pe_size, lat_disp, lat_col, in_dim=3, width=256):
		self.n_trunk, self.n_disp, self.n_col = n_trunk, n_disp, n_col
		self.pe_size, self.lat_disp, self.lat_col = pe_size, lat_disp, lat_col
		self.in_dim, self.width = in_dim, width

	@property
	def n_weights(self):
		return 2 * (self.n_trunk + self.n_disp + 1 + self.n_col + 1)


def _fill_params(spec, B, avg_col, weights):
	p = MlpParams()
	p.width, p.in_dim, p.pe_size = spec.width, spec.in_dim, spec.pe_size
	p.n_trunk, p.n_disp, p.n_col = spec.n_trunk, spec.n_disp, spec.n_col
	p.lat_disp, p.lat_col = spec.lat_disp, spec.lat_col
	p.B = None if B is None else B.data_ptr()
	p.avg_col = None if avg_col is None else avg_col.data_ptr()
	it = iter(weights)
	for i in range(spec.n_trunk):
		p.trunk_w[i] = next(it).data_ptr()
		p.trunk_b[i] = next(it).data_ptr()
	for i in range(spec.n_disp + 1):
		p.disp_w[i] = next(it).data_ptr()
		p.disp_b[i] = next(it).data_ptr()
	for i in range(spec.n_col + 1):
		p.col_w[i] = next(it).data_ptr()
		p.col_b[i] = next(it).data_ptr()
	return p


class _MLP(torch.autograd.Function):
	"""disp, col = MLP(pos; latents, weights)   -- find_mlp_fwd / find_mlp_bwd."""

	@staticmethod
	def forward(ctx, spec, pos, lat_disp, lat_col, B, avg_col, *weights):
		if len(weights) != spec.n_weights:
			raise RuntimeError(f'find_amd.mlp: expected {spec.n_weights} weight tensors, got {len(weights)}')
		_require_gpu(pos, lat_disp, lat_col, B, avg_col, *weights)
		if pos.requires_grad:
			raise RuntimeError('find_amd.mlp: gradient w.r.t. positions is not part of the FIND path (template / sampled '
							   'GT points carry no grad, model.py:285, losses.py:39-51)')
		L = _lib.lib()
		pos, lat_disp, lat_col, B, avg_col = _c(pos), _c(lat_disp), _c(lat_col), _c(B), _c(avg_col)
		weights = tuple(w.contiguous() for w in weights)
		pos_batch, V, _ = pos.shape
		n_feet = pos_batch
		for lat in (lat_disp, lat_col):
			if lat is not None:
				n_feet = lat.shape[0]
		if pos_batch != 1 and pos_batch != n_feet:
			raise RuntimeError(f'find_amd.mlp: pos batch {pos_batch} does not match latent batch {n_feet}')
		p = _fill_params(spec, B, avg_col, weights)
		save = any(ctx.needs_input_grad)
		nbytes = L.find_mlp_ws_bytes(ctypes.byref(p), pos_batch, n_feet, V, int(save))
		if nbytes < 0:
			check(-1, 'find_mlp_ws_bytes')
		ws = _ws(nbytes, pos.device)
		disp = torch.empty(n_feet, V, 3, device=pos.device, dtype=torch.float32)
		col = torch.empty(n_feet, V, 3, device=pos.device, dtype=torch.float32)
		check(L.find_mlp_fwd(ctypes.byref(p), ptr(pos), pos_batch, n_feet, V, ptr(lat_disp), ptr(lat_col), ptr(disp), ptr(col),
							 ptr(ws), ws.numel(), int(save), current_stream(pos.device)), 'find_mlp_fwd')
		if save:
			ctx.spec = spec
			ctx.dims = (pos_batch, n_feet, V)
			ctx.ws = ws
			ctx.save_for_backward(pos, lat_disp, lat_col, B, avg_col, *weights)
		return disp, col

	@staticmethod
	def backward(ctx, g_disp, g_col):
		L = _lib.lib()
		spec = ctx.spec
		pos, lat_disp, lat_col, B, avg_col, *weights = ctx.saved_tensors
		pos_batch, n_feet, V = ctx.dims
		g_disp, g_col = _c(g_disp), _c(g_col)
		p = _fill_params(spec, B, avg_col, weights)
		grads = [torch.empty_like(w) for w in weights]
		g_lat_disp = torch.empty_like(lat_disp) if lat_disp is not None else None
		g_lat_col = torch.empty_like(lat_col) if lat_col is not None else None
		G = MlpGrads()
		it = iter(grads)
		for i in range(spec.n_trunk):
			G.trunk_w[i] = next(it).data_ptr()
			G.trunk_b[i] = next(it).data_ptr()
		for i in range(spec.n_disp + 1):
			G.disp_w[i] = next(it).data_ptr()
			G.disp_b[i] = next(it).data_ptr()
		for i in range(spec.n_col + 1):
			G.col_w[i] = next(it).data_ptr()
			G.col_b[i] = next(it).data_ptr()
		G.lat_disp = None if g_lat_disp is None else g_lat_disp.data_ptr()
		G.lat_col = None if g_lat_col is None else g_lat_col.data_ptr()
		sb = L.find_mlp_bwd_scratch_bytes(ctypes.byref(p), pos_batch, n_feet, V)
		scratch = _ws(sb, pos.device)
		check(L.find_mlp_bwd(ctypes.byref(p), ptr(pos), pos_batch, n_feet, V, ptr(lat_disp), ptr(lat_col), ptr(g_disp), ptr(g_col),
							 ptr(ctx.ws), ctx.ws.numel(), ptr(scratch), scratch.numel(), ctypes.byref(G),
							 current_stream(pos.device)), 'find_mlp_bwd')
		return (None, None, g_lat_disp, g_lat_col, None, None, *grads)


def mlp(spec, pos, lat_disp, lat_col, B, avg_col, weights):
	"""Fused Fourier-PE + trunk + heads.  pos (1|N, V, 3); lat_disp (N, Ld)|None; lat_col (N, Lc)|None;
	weights: flat list [trunk w,b ..., disp w,b ..., col w,b ...] in reference state_dict order.
	Returns disp (N,V,3), col (N,V,3)   (reference: NeuralDisplacementField.forward, model.py:393-453)."""
	return _MLP.apply(spec, pos, lat_disp, lat_col, B, avg_col, *weights)


# ----------------------------------------------------------------------------------------------- registration
class _Register(torch.autograd.Function):
	"""X = ((verts + disp) * S) @ R(euler XYZ) + t     (model.py:481-491)."""

	@staticmethod
	def forward(ctx, verts, disp, reg):
		_require_gpu(verts, disp, reg)
		if verts.requires_grad:
			raise RuntimeError('find_amd.register_points: template vertices carry no gradient in FIND (model.py:285)')
		L = _lib.lib()
		verts, disp, reg = _c(verts), _c(disp), _c(reg)
		n_feet, V, _ = disp.shape
		vb = verts.shape[0]
		if vb not in (1, n_feet) or reg.shape != (n_feet, 9):
			raise RuntimeError(f'find_amd.register_points: bad shapes verts {tuple(verts.shape)} disp {tuple(disp.shape)} reg {tuple(reg.shape)}')
		out = torch.empty_like(disp)
		check(L.find_register_fwd(ptr(verts), vb, ptr(disp), ptr(reg), n_feet, V, ptr(out), current_stream(disp.device)), 'find_register_fwd')
		ctx.save_for_backward(verts, disp, reg)
		return out

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		verts, disp, reg = ctx.saved_tensors
		n_feet, V, _ = disp.shape
		g = _c(g)
		d_disp = torch.empty_like(disp)
		d_reg = torch.empty_like(reg)
		ws = _ws(L.find_register_bwd_ws_bytes(n_feet, V), disp.device)
		check(L.find_register_bwd(ptr(verts), verts.shape[0], ptr(disp), ptr(reg), ptr(g), n_feet, V, ptr(d_disp), ptr(d_reg),
								  ptr(ws), ws.numel(), current_stream(disp.device)), 'find_register_bwd')
		return None, d_disp, d_reg


def register_points(verts, disp, reg):
	return _Register.apply(verts, disp, reg)
