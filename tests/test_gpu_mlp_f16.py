"""Opt-in fp16 matrix-pipe mode of the MLP's 256 -> 256 layers (BASELINE.json configs[4]: "fp16 MLP with MFMA tiles";
find_amd.functional.set_mlp_precision, gemm5_kernel).  The reference computes in fp32 only, so this mode has no reference
counterpart: the kernel is checked exactly against a float64 product of the fp16-ROUNDED operands (what the matrix pipe is
meant to compute: exact products, fp32 sums), and the whole model against the fp32 path with the tolerance fp16 rounding implies."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fp32_baseline():
	"""The comparisons of this file are against the fp32-MFMA kernels, whatever the process default is (bf16x3 since round 4)."""
	from find_amd import functional as F
	prev = F.set_mlp_precision('fp32')
	try:
		yield
	finally:
		F.set_mlp_precision(prev)


@pytest.fixture
def fp16_mode():
	"""fp16 mode with gemm5 forced for every row count (by default launches of fewer than 1024 32-row units stay on the fp32 kernels,
	which are faster there): the isolated-kernel tests below pin gemm5's tile-edge arithmetic at tiny shapes too."""
	from find_amd import _lib, functional as F
	prev = F.set_mlp_precision('fp16')
	_lib.set_tuning('gemm5_min_units', 1)
	try:
		yield
	finally:
		_lib.set_tuning('gemm5_min_units', 1024)
		F.set_mlp_precision(prev)


@pytest.mark.parametrize('n_feet,n_pts', [(1, 1), (1, 70), (3, 1002), (2, 6890), (16, 6890), (16, 50002)])
def test_linear_relu_fp16_operands_exact(fp16_mode, n_feet, n_pts):
	"""y = relu(x w^T + b) with x, w rounded to fp16 (RNE), exact products, fp32 accumulation: against float64 on the rounded operands
	the only error left is the fp32 summation (<= 1e-4 at these magnitudes); rows past a foot's last 32-row unit are never written."""
	from find_amd import _lib
	L = _lib.lib()
	gen = torch.Generator().manual_seed(n_feet * 7919 + n_pts)
	x = torch.randn(n_feet * n_pts, 256, generator=gen)
	w = torch.randn(256, 256, generator=gen) / 16
	b = torch.randn(256, generator=gen)
	xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
	pad = torch.full((n_feet * n_pts + 64, 256), float('nan'), device='cuda')
	y = pad[:n_feet * n_pts]
	_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), n_feet, n_pts, _lib.ptr(y),
									  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'find_linear_relu_fwd')
	xr = x.half().double().numpy()
	wr = w.half().double().numpy()
	want = np.maximum(xr @ wr.T + b.double().numpy(), 0.0)
	got = y.cpu().numpy()
	assert np.isfinite(got).all()
	assert np.abs(got - want).max() < 1e-4
	assert torch.isnan(pad[n_feet * n_pts:]).all()   # nothing stored past the last row
	# and it is the fp16 path that ran: the fp32 product differs from it by what rounding the operands costs
	full = np.maximum(x.double().numpy() @ w.double().numpy().T + b.double().numpy(), 0.0)
	assert 1e-4 < np.abs(got - full).max() < 5e-2


def _run_model(n_feet, n_verts, shared, backward_precision=None, between=None):
	"""between: called after the forward, before the backward (a knob turned in between).  backward_precision: run the BACKWARD of every MLP call in this arithmetic whatever the forward ran in (the precision code saved on the
	autograd node is replaced): same saved activations, hence the same ReLU masks, on both sides of a comparison."""
	from find_amd import synthetic
	dev = torch.device('cuda:0')
	model = synthetic.make_model(n_verts if shared else 1002, train_size=n_feet, val_size=1, device=dev)   # (free points need no template of their size)
	lat = synthetic.latents(n_feet, seed=3, device=dev)
	lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
	for p in model.parameters():
		p.grad = None
	if shared:
		res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
		out = torch.cat([res['verts'], res['col']], -1)
	else:
		g = torch.Generator().manual_seed(5)
		pos = (torch.rand(n_feet, n_verts, 3, generator=g) * 0.2 - 0.1).to(dev)
		res = model(pos, shapevec=lv['shapevec'], texvec=lv['texvec'], posevec=lv['posevec'])
		out = torch.cat([res['disp'], res['col']], -1)
	wgt = torch.linspace(0.5, 1.5, out.numel(), device=dev).reshape(out.shape)
	if backward_precision is not None:
		from find_amd import functional as FF
		code, seen, todo, hit = FF._PRECISION_CODE[backward_precision], set(), [out.grad_fn], 0
		while todo:
			fn = todo.pop()
			if fn is None or fn in seen:
				continue
			seen.add(fn)
			if hasattr(fn, 'precision'):
				fn.precision = code
				hit += 1
			todo.extend(f for f, _ in fn.next_functions)
		assert hit >= 1
	if between is not None:
		between()
	(out * wgt).sum().backward()
	torch.cuda.synchronize()
	grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
	grads.update({'latent.' + k: v.grad.detach().clone() for k, v in lv.items() if v.grad is not None})
	return out.detach().clone(), grads


@pytest.mark.parametrize('n_feet,n_verts,shared', [(3, 1002, True), (2, 1002, False), (2, 70, False), (16, 6890, True)])
def test_model_fp16_close_to_fp32(n_feet, n_verts, shared):
	"""Whole model, forward and every gradient (gemm5 forward / dX, dw3 weight gradients), fp16 mode against the fp32 path: operands
	rounded to 2^-11 relative through 11 layers give outputs within 1e-4 absolute (observed 5e-6 .. 7e-6 at the seeded initialisation:
	disp is bounded by 0.1, colours by 1) and gradients within 1e-2 of each tensor's largest entry (observed <= 5e-3, tools/f16_deviation.py)."""
	from find_amd import functional as F
	assert F.get_mlp_precision() == 'fp32'
	out32, g32 = _run_model(n_feet, n_verts, shared)
	from find_amd import _lib
	prev = F.set_mlp_precision('fp16')
	_lib.set_tuning('gemm5_min_units', 1)   # every launch on the fp16 kernels, whatever its size
	try:
		out16, g16 = _run_model(n_feet, n_verts, shared)
	finally:
		_lib.set_tuning('gemm5_min_units', 1024)
		F.set_mlp_precision(prev)
	assert torch.isfinite(out16).all()
	d = (out16 - out32).abs().max().item()
	assert d < 1e-4, d
	if n_feet * n_verts >= 1000:
		assert d > 0.0   # the fp16 kernels did run
	assert g16.keys() == g32.keys()
	# (a handful of points: one pre-activation whose sign flips under fp16 rounding moves a gradient by a visible fraction -- there the
	# comparison is a sanity bound; the tile-edge arithmetic of tiny shapes is pinned exactly by test_linear_relu_fp16_operands_exact
	# and test_linear_wgrad_fp16_operands_exact)
	rel = 1e-2 if n_feet * n_verts >= 1000 else 2e-1
	for n in g32:
		scale = max(1e-6, g32[n].abs().max().item())
		assert torch.isfinite(g16[n]).all(), n
		assert (g16[n] - g32[n]).abs().max().item() < rel * scale, n
	# back in fp32 mode the result is the fp32 result again, bit for bit
	out32b, _ = _run_model(n_feet, n_verts, shared)
	assert torch.equal(out32b, out32)


@pytest.mark.parametrize('n_feet,n_pts', [(1, 1), (5, 15), (1, 70), (3, 1002), (2, 6890), (16, 6890), (16, 50002)])
def test_linear_wgrad_fp16_operands_exact(fp16_mode, n_feet, n_pts):
	"""dw3_kernel: dW = dz^T x with both operands rounded to fp16, exact products, fp32 sums -- against float64 on the rounded
	operands; the bias gradient is summed from the un-rounded dz.  Covers feet shorter than one 64-row chunk and zero-filled tails."""
	from test_gpu_mlp import _wgrad
	dz, x, dw, db = _wgrad(n_feet, n_pts, seed=n_feet * 131 + n_pts)
	want = dz.half().double().t() @ x.half().double()
	wantb = dz.double().sum(0)
	assert torch.isfinite(dw).all() and torch.isfinite(db).all()
	scale = max(1.0, want.abs().max().item())
	# (fp32 sums of up to 800 032 exact products per element: the summation error grows with the row count)
	assert (dw.double() - want).abs().max().item() < (1e-5 if n_feet * n_pts <= 16 * 6890 else 4e-5) * scale
	assert (db.double() - wantb).abs().max().item() < 1e-4 * max(1.0, wantb.abs().max().item())
	full = dz.double().t() @ x.double()
	if n_feet * n_pts >= 64:
		assert (dw.double() - full).abs().max().item() > 1e-6 * scale   # it is the fp16 path that ran


def test_c5_dense_template_fp16_as_configured():
	"""BASELINE.json configs[4] as it is specified: the 50 002-vertex dense template evaluated for several feet WITH the fp16 matrix
	pipe at its default launch thresholds (every launch here has >= 1024 32-row units, so trunk, heads, dX chain and weight gradients
	all run on gemm5 / dw3), against the fp32 path on the same inputs: outputs within 1e-4 absolute, every gradient within the
	documented bound (5e-3 of the tensor's largest entry; asserted at 1e-2 like the smaller shapes)."""
	from find_amd import functional as F
	assert F.get_mlp_precision() == 'fp32'
	n_feet, n_verts = 3, 50002
	out32, g32 = _run_model(n_feet, n_verts, True)
	prev = F.set_mlp_precision('fp16')
	try:
		out16, g16 = _run_model(n_feet, n_verts, True)
	finally:
		F.set_mlp_precision(prev)
	assert out16.shape == (n_feet, n_verts, 6) and torch.isfinite(out16).all()
	d = (out16 - out32).abs().max().item()
	assert 0.0 < d < 1e-4, d
	worst = 0.0
	for n in g32:
		scale = max(1e-6, g32[n].abs().max().item())
		assert torch.isfinite(g16[n]).all(), n
		e = (g16[n] - g32[n]).abs().max().item() / scale
		worst = max(worst, e)
		assert e < 1e-2, (n, e)
	print(f'C5 fp16 vs fp32: outputs {d:.2e}, worst gradient deviation {worst:.2e} of the tensor maximum')


def test_fp16_stored_activations_agree_with_fp32_stored_ones():
	"""Inside the opt-in fp16 mode the heads' hidden activations and their gradients are STORED as fp16 at the large shared-template shapes
	("act16" knob, csrc/mlp.hip use_act16; BASELINE configs[4] is bound by these bytes).  The matrix pipe rounds the same values to fp16 when
	they are read as fp32, so switching the storage changes little: outputs within 2e-5, every gradient within 5e-3 of its tensor's largest
	entry of the act16 = 0 result -- and not nothing (the 3-wide output layers and the bias / latent sums now see rounded values)."""
	from find_amd import functional as F
	from find_amd import _lib
	prev = F.set_mlp_precision('fp16')
	try:
		out_a, g_a = _run_model(16, 6890, True)
		_lib.set_tuning('act16', 0)
		out_b, g_b = _run_model(16, 6890, True)
	finally:
		_lib.set_tuning('act16', 1)
		F.set_mlp_precision(prev)
	# a knob turned between a forward and its backward does not split them: the backward follows the note its forward left in the context
	try:
		F.set_mlp_precision('fp16')
		out_c, g_c = _run_model(16, 6890, True, between=lambda: _lib.set_tuning('act16', 0))
	finally:
		_lib.set_tuning('act16', 1)
		F.set_mlp_precision(prev)
	assert torch.equal(out_c, out_a)
	for n in g_a:
		assert torch.equal(g_c[n], g_a[n]), n
	d = (out_a - out_b).abs().max().item()
	assert 0.0 < d < 2e-5, d
	for n in g_b:
		scale = max(1e-6, g_b[n].abs().max().item())
		assert torch.isfinite(g_a[n]).all(), n
		assert (g_a[n] - g_b[n]).abs().max().item() < 5e-3 * scale, (n, (g_a[n] - g_b[n]).abs().max().item() / scale)


@pytest.mark.parametrize('n_feet,n_verts', [(16, 6890), (5, 6890), (4, 10002)])
def test_broadcast_layer_formed_by_its_readers_equals_the_stored_one(n_feet, n_verts):
	"""bcast_fold (csrc/mlp.hip use_fold, mlp_gemm5.h VIRT, dw3_h16v_kernel): inside act16 the output of a head's broadcast first layer,
	fp16(relu(P[v] + bias[foot])), is not stored -- the second layer's forward GEMM, the ReLU mask of its dX GEMM and the x operand of its
	weight gradient form it from the V x 256 product and the bias rows.  Same add, same max, same rounding, same summation order everywhere:
	outputs and every gradient are BIT-IDENTICAL to the stored path; a knob turned between forward and backward does not split them."""
	from find_amd import functional as F
	from find_amd import _lib
	prev = F.set_mlp_precision('fp16')
	try:
		assert _lib.get_tuning('bcast_fold') == 1
		out_a, g_a = _run_model(n_feet, n_verts, True)
		out_c, g_c = _run_model(n_feet, n_verts, True, between=lambda: _lib.set_tuning('bcast_fold', 0))
		out_b, g_b = _run_model(n_feet, n_verts, True)   # (the knob is still 0: the stored path)
	finally:
		_lib.set_tuning('bcast_fold', 1)
		F.set_mlp_precision(prev)
	assert torch.equal(out_a, out_b) and torch.equal(out_c, out_a)
	for n in g_b:
		assert torch.isfinite(g_a[n]).all(), n
		assert torch.equal(g_a[n], g_b[n]), (n, (g_a[n] - g_b[n]).abs().max().item() / max(1e-30, g_b[n].abs().max().item()))
		assert torch.equal(g_c[n], g_a[n]), n


def test_two_models_in_one_process_may_differ_in_precision():
	"""find_mlp_params.precision travels with each call: a model pinned to fp16 and a model pinned to fp32 interleave their passes,
	and the fp32 one stays bit-identical to a run without any fp16 model around (the precision is no longer process-wide state)."""
	from find_amd import functional as F
	from find_amd import synthetic
	assert F.get_mlp_precision() == 'fp32'
	dev = torch.device('cuda:0')
	n_feet, n_verts = 16, 6890   # >= 1024 32-row units per head layer: the fp16 kernels engage at their default thresholds
	lat = synthetic.latents(n_feet, seed=3, device=dev)
	a = synthetic.make_model(n_verts, train_size=n_feet, val_size=1, device=dev)
	b = synthetic.make_model(n_verts, train_size=n_feet, val_size=1, device=dev)

	def run(m):
		lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
		m.zero_grad()
		res = m.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
		(res['verts'].sum() + res['col'].sum()).backward()
		return res['verts'].detach().clone(), m.mlp_disp[2].weight.grad.detach().clone()

	base_v, base_g = run(b)
	a.set_mlp_precision('fp16')
	b.set_mlp_precision('fp32')
	va, ga = run(a)
	vb, gb = run(b)
	va2, _ = run(a)
	assert torch.equal(vb, base_v) and torch.equal(gb, base_g)
	assert torch.equal(va, va2)
	d = (va - vb).abs().max().item()
	assert 0.0 < d < 1e-4, d
	assert not torch.equal(ga, gb)
	# a forward in fp16 followed by a change of the process default still runs its backward in fp16
	lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
	a.set_mlp_precision(None)
	prev = F.set_mlp_precision('fp16')
	try:
		a.zero_grad()
		res = a.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
	finally:
		F.set_mlp_precision(prev)
	(res['verts'].sum() + res['col'].sum()).backward()
	assert torch.equal(a.mlp_disp[2].weight.grad, ga)
