"""bf16x3: the default arithmetic of the MLP's large 256 -> 256 layers (find_amd.functional.set_mlp_precision; csrc/mlp_gemm7.h, mlp_dw6.h).
Every fp32 operand is split EXACTLY into three bf16 pieces (a = a1 + a2 + a3: round to nearest, subtract, twice; bf16 has fp32's exponent
range) and the six products of relative size >= 2^-18 run on the bf16 matrix pipe with fp32 accumulation; what is left out is <= 2^-26 of a
product -- a quarter of ONE fp32 rounding.  The claim tested here is that this is fp32 arithmetic as far as anyone can tell: against a
float64 evaluation of the same layer the bf16x3 kernels are as close as the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32), including for
operands of very different magnitudes, and the model's outputs and gradients agree between the two to the noise of fp32 summation order.
(The fp16 mode -- operands ROUNDED to 11 bits, tests/test_gpu_mlp_f16.py -- differs from both by four orders of magnitude more.)"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def every_size():
	"""bf16x3 kernels for every row count (by default launches of fewer than 1024 32-row units stay on the fp32 kernels, which are faster
	there): pins the tile-edge arithmetic at tiny shapes."""
	from find_amd import _lib
	_lib.set_tuning('gemm6_min_units', 1)
	try:
		yield
	finally:
		_lib.set_tuning('gemm6_min_units', 1024)


def _linear(n_feet, n_pts, x, w, b, precision):
	from find_amd import _lib, functional as F
	L = _lib.lib()
	prev = F.set_mlp_precision(precision)
	try:
		pad = torch.full((n_feet * n_pts + 64, 256), float('nan'), device='cuda')
		y = pad[:n_feet * n_pts]
		_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), n_feet, n_pts, _lib.ptr(y),
										  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'find_linear_relu_fwd')
		torch.cuda.synchronize()
	finally:
		F.set_mlp_precision(prev)
	return y, pad


@pytest.mark.parametrize('n_feet,n_pts', [(1, 1), (1, 70), (3, 1002), (2, 6890), (16, 6890)])
def test_linear_relu_bf16x3_as_accurate_as_fp32_mfma(every_size, n_feet, n_pts):
	"""y = relu(x w^T + b) at tile edges and at the C2 shape, columns of x spanning nine orders of magnitude: the bf16x3 kernel and the
	fp32-MFMA kernel against float64.  Both carry the error of fp32 accumulation; bf16x3 may not be further from float64 than 1.5 x the
	fp32 kernel (+ one ulp of the largest output), and rows past a foot's last 32-row unit are never written."""
	gen = torch.Generator().manual_seed(n_feet * 7919 + n_pts)
	x = torch.relu(torch.randn(n_feet * n_pts, 256, generator=gen))
	x[:, :9] *= torch.logspace(-6, 3, 9)
	w = torch.randn(256, 256, generator=gen) / 16
	b = torch.randn(256, generator=gen)
	xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
	y6, pad6 = _linear(n_feet, n_pts, xd, wd, bd, 'bf16x3')
	y4, _ = _linear(n_feet, n_pts, xd, wd, bd, 'fp32')
	want = torch.relu(x.double() @ w.double().t() + b.double())
	scale = want.abs().max().item()
	e6 = (y6.cpu().double() - want).abs().max().item()
	e4 = (y4.cpu().double() - want).abs().max().item()
	assert torch.isfinite(y6).all()
	assert e6 <= 1.5 * e4 + 1.2e-7 * scale, (e6, e4, scale)
	assert e6 < 2e-6 * scale
	assert torch.isnan(pad6[n_feet * n_pts:]).all()   # nothing stored past the last row
	if n_feet * n_pts >= 64:
		assert not torch.equal(y6, y4)                 # (it is the other kernel that ran: same values to rounding, another summation order)


@pytest.mark.parametrize('n_feet,n_pts', [(1, 1), (5, 15), (1, 70), (3, 1002), (2, 6890), (16, 6890)])
def test_linear_wgrad_bf16x3_as_accurate_as_fp32_mfma(every_size, n_feet, n_pts):
	"""dW = dz^T x, db = sum dz (dw6_kernel + the slab reduce) against float64, beside the fp32-MFMA kernel on the same operands:
	feet shorter than one 16-row chunk, zero-filled tails, the C2 shape."""
	from find_amd import functional as F
	from test_gpu_mlp import _wgrad
	res = {}
	for prec in ('bf16x3', 'fp32'):
		prev = F.set_mlp_precision(prec)
		try:
			res[prec] = _wgrad(n_feet, n_pts, seed=n_feet * 131 + n_pts)
		finally:
			F.set_mlp_precision(prev)
	dz, x, dw6, db6 = res['bf16x3']
	_, _, dw4, db4 = res['fp32']
	want = dz.double().t() @ x.double()
	wantb = dz.double().sum(0)
	scale = max(1e-30, want.abs().max().item())
	e6 = (dw6.double() - want).abs().max().item()
	e4 = (dw4.double() - want).abs().max().item()
	assert torch.isfinite(dw6).all() and torch.isfinite(db6).all()
	# (the same fp32 sums over up to 110 240 rows in another order: within a small factor of each other, both far below 1e-4)
	assert e6 <= 2.5 * e4 + 2.4e-7 * scale, (e6, e4, scale)
	assert e6 < 3e-6 * scale
	assert (db6.double() - wantb).abs().max().item() < 1e-5 * max(1.0, wantb.abs().max().item())


def _run_model(n_feet, n_verts, shared, precision, backward_precision=None):
	from find_amd import functional as F
	from test_gpu_mlp_f16 import _run_model as run
	prev = F.set_mlp_precision(precision)
	try:
		return run(n_feet, n_verts, shared, backward_precision=backward_precision)
	finally:
		F.set_mlp_precision(prev)


@pytest.mark.parametrize('n_feet,n_verts,shared', [(3, 1002, True), (2, 1002, False), (16, 6890, True), (16, 1000, False),
													 (2, 1, False), (3, 33, False), (2, 70, False), (16, 257, False)])   # (tile edges of the fused chains: 1 row, 32 + 1, 64 + 6, 4 x 64 + 1)
def test_model_bf16x3_equals_fp32_mfma_to_summation_order(every_size, n_feet, n_verts, shared):
	"""Whole model, bf16x3 (gemm7 forward / dX, dw6 weight gradients, fused6 layer chains) against the fp32-MFMA path.
	Forward: outputs within 2e-6 absolute (disp is bounded by 0.1, colours by 1: a few ulps).
	Backward, on the SAME saved activations (forward in fp32 on both sides, the backward's arithmetic switched on the autograd node): every
	gradient within 5e-5 of its tensor's largest entry -- what two fp32 evaluations with different summation orders differ by.
	End to end (forward AND backward in bf16x3) the same bound holds where no ReLU sits on a tie (the cases of up to 3 000 rows); from 16 x 257
	free points and 16 x 6890 -- tens of millions of activations, a pre-activation within 1e-7 of zero is expected once or twice, its mask then differs between two fp32-accurate forwards and one row's
	contribution moves by its full size (~3e-4 of the first layer's gradient) -- the bound is 1e-3: the noise of a discontinuous function,
	not of the arithmetic (the fp16 mode is allowed 1e-4 / 1e-2 in the same test, and its deviations are not ties)."""
	out32, g32 = _run_model(n_feet, n_verts, shared, 'fp32')
	outx3, gx3 = _run_model(n_feet, n_verts, shared, 'bf16x3')
	_, gmix = _run_model(n_feet, n_verts, shared, 'fp32', backward_precision='bf16x3')
	assert torch.isfinite(outx3).all()
	d = (outx3 - out32).abs().max().item()
	assert d < 2e-6, d
	assert gx3.keys() == g32.keys() == gmix.keys()
	worst, worst_e2e = 0.0, 0.0
	for n in g32:
		scale = max(1e-12, g32[n].abs().max().item())
		assert torch.isfinite(gx3[n]).all() and torch.isfinite(gmix[n]).all(), n
		e = (gmix[n] - g32[n]).abs().max().item() / scale
		worst = max(worst, e)
		assert e < 5e-5, (n, e)
		e2 = (gx3[n] - g32[n]).abs().max().item() / scale
		worst_e2e = max(worst_e2e, e2)
		assert e2 < (1e-3 if n_feet * n_verts > 3500 else 5e-5), (n, e2)
	print(f'bf16x3 vs fp32 MFMA ({n_feet} x {n_verts}, shared={shared}): outputs {d:.1e}, worst gradient deviation {worst:.1e} of the tensor maximum '
		  f'on the same activations, {worst_e2e:.1e} end to end')


@pytest.mark.parametrize('case', list('abcdef'))
def test_forward_matches_reference_golden_under_bf16x3(every_size, golden_main, case):
	"""The reference's own forward (tests/golden/mlp_main.npz, generated by importing NeuralDisplacementField) with EVERY 256 -> 256 layer
	on the bf16x3 kernels, whatever its size: the same 2e-5 bound the fp32-MFMA path is held to (tests/test_gpu_mlp.py)."""
	from find_amd import functional as F
	from test_gpu_mlp import _model_from_golden, _t
	assert F.get_mlp_precision() == 'bf16x3'   # the default
	m = _model_from_golden(golden_main)
	g = {k: _t(golden_main[f'fwd/{case}/{k}']) for k in ['pos', 'shapevec', 'texvec', 'posevec', 'disp', 'col']}
	with torch.no_grad():
		res = m(g['pos'], shapevec=g['shapevec'], texvec=g['texvec'], posevec=g['posevec'])
	ed = (res['disp'] - g['disp']).abs().max().item()
	ec = (res['col'] - g['col']).abs().max().item()
	assert ed < 2e-5 and ec < 2e-5, (case, ed, ec)


@pytest.mark.parametrize('case', ['b', 'f'])
def test_backward_matches_reference_golden_under_bf16x3(every_size, golden_main, case):
	"""... and its gradients (autograd through the reference model: loss, latents, every weight tensor of case 'b' -- 16 feet on one
	1000-vertex template -- and 'f', BASELINE configs[0]'s single foot on a 1k-vertex template)."""
	from test_gpu_mlp import _model_from_golden, _t
	m = _model_from_golden(golden_main)
	lat = {k: _t(golden_main[f'fwd/{case}/{k}']).requires_grad_(True) for k in ['shapevec', 'texvec', 'posevec']}
	res = m(_t(golden_main[f'fwd/{case}/pos']), **lat)
	loss = (res['disp'] ** 2).sum() + (res['col'] ** 2).sum()
	loss.backward()
	ref_loss = float(golden_main[f'grad/{case}/loss'])
	assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss)
	for k in lat:
		ref = golden_main[f'grad/{case}/{k}']
		assert np.abs(lat[k].grad.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k
	n = 0
	for k, prm in m.named_parameters():
		key = f'grad/{case}/sd/{k}'
		if key not in golden_main:
			continue
		got, ref = prm.grad.cpu().numpy(), golden_main[key]
		if ref.shape != got.shape:
			got = got.reshape(-1)[::17]
		assert float(np.abs(got - ref).max()) < 1e-4 * max(1.0, float(np.abs(ref).max())), k
		n += 1
	assert n >= 20


@pytest.mark.parametrize('n_feet,n_verts', [(16, 6890), (5, 6890), (4, 10002), (3, 1002)])
def test_broadcast_layer_formed_by_its_readers_equals_the_stored_one(every_size, n_feet, n_verts):
	"""bcast_fold in the default arithmetic (csrc/mlp.hip use_fold; mlp_gemm7.h VIRT, dw6v_kernel): the output of a head's broadcast first
	layer, relu(P[v] + bias[foot]), is not stored -- the second layer's forward GEMM, the ReLU mask of its dX GEMM and the X operand of its
	weight gradient form it from the V x 256 product and the bias rows.  Same add, same max, same splits, same summation order: outputs and
	every gradient BIT-IDENTICAL to the stored path (16 / 5 feet of the 6890-vertex template: a partial last unit per foot; 10 002 and
	1002 vertices: other tile edges -- the latter through gemm7 only under the every_size fixture)."""
	from find_amd import _lib
	assert _lib.get_tuning('bcast_fold') == 1
	try:
		_lib.set_tuning('footsum_fold', 0)   # (it rides on the virtual mask and sums in another order: its own test below)
		out_a, g_a = _run_model(n_feet, n_verts, True, 'bf16x3')
		_lib.set_tuning('bcast_fold', 0)
		out_b, g_b = _run_model(n_feet, n_verts, True, 'bf16x3')
	finally:
		_lib.set_tuning('bcast_fold', 1)
		_lib.set_tuning('footsum_fold', 1)
	assert torch.equal(out_a, out_b)
	for n in g_b:
		assert torch.isfinite(g_a[n]).all(), n
		assert torch.equal(g_a[n], g_b[n]), (n, (g_a[n] - g_b[n]).abs().max().item() / max(1e-30, g_b[n].abs().max().item()))


@pytest.mark.parametrize('n_feet,n_verts', [(16, 6890), (5, 6890), (4, 10002), (3, 1002), (33, 1002)])
def test_foot_sums_formed_inside_the_dx_gemm_equal_the_separate_pass(every_size, n_feet, n_verts):
	"""footsum_fold (csrc/mlp.hip; mlp_gemm7.h FSUM): the dX GEMM of a head's second layer does not store the broadcast layer's dZ -- it
	forms the two sums the backward reads of it, over the feet (per template row) and over the rows (per foot), in its epilogue: units in
	tile-major order, a wave summing its 16 columns over a tile's feet in registers, at most two partial sums per tile, the column sums
	through an LDS table.  Against the separate footsum pass: the same numbers in another summation order -- every gradient within 2e-6 of
	its tensor's largest entry, outputs identical (the forward is not touched); both deterministic.  (16 / 5 / 33 feet: tiles cut at
	every position by the workgroup ranges; 10 002 and 1002 vertices: partial last tiles.)"""
	from find_amd import _lib
	assert _lib.get_tuning('footsum_fold') == 1
	out_a, g_a = _run_model(n_feet, n_verts, True, 'bf16x3')
	out_a2, g_a2 = _run_model(n_feet, n_verts, True, 'bf16x3')
	try:
		_lib.set_tuning('footsum_fold', 0)
		out_b, g_b = _run_model(n_feet, n_verts, True, 'bf16x3')
	finally:
		_lib.set_tuning('footsum_fold', 1)
	assert torch.equal(out_a, out_b)
	worst = 0.0
	for n in g_b:
		assert torch.isfinite(g_a[n]).all(), n
		assert torch.equal(g_a[n], g_a2[n]), n
		e = (g_a[n] - g_b[n]).abs().max().item() / max(1e-30, g_b[n].abs().max().item())
		worst = max(worst, e)
		assert e < 2e-6, (n, e)
	print(f'footsum_fold vs separate pass ({n_feet} x {n_verts}): worst gradient deviation {worst:.1e} of the tensor maximum')


@pytest.mark.parametrize('n_feet,n_verts,shared', [(16, 6890, True), (16, 1000, False), (8, 6890, True), (6, 6890, True), (32, 6890, True)])
def test_bf16x3_backward_is_bit_reproducible(n_feet, n_verts, shared):
	"""No float atomics and a fixed split geometry: repeated passes of the same forward + backward give bit-identical outputs and gradients --
	the headline's main pass (gemm7 / dw6, the trunk on 32-row fused6 tiles) and the texture pass's shape (64-row fused6 tiles, grouped weight
	gradients); the slab reduces are deterministic, the chains' LDS tile changes hands behind barriers only (tools/check_determinism.py is
	the long version: 100 - 200 passes).  8, 6 and 32 feet: row ranges of gemm7 with an even unit count -- where round 5 found single dX
	elements changing from run to run (a 128-bit buffer store whose data register the next VALU instruction overwrote: csrc/common.h
	store_b128, tools/store_hazard_probe.hip; a data-parallel rank of the headline holds 8 feet)."""
	a = _run_model(n_feet, n_verts, shared, 'bf16x3')
	for _ in range(3 if n_feet != 16 else 1):
		b = _run_model(n_feet, n_verts, shared, 'bf16x3')
		assert torch.equal(a[0], b[0])
		for n in a[1]:
			assert torch.equal(a[1][n], b[1][n]), n
