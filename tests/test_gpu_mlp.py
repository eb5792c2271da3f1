"""GPU parity: HIP MLP / registration (through the C-ABI) vs the golden vectors captured from the reference and
vs the torch-CPU oracle.  Tolerance: the north_star's 1e-4 (fp32); observed errors are ~1e-6."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model_from_golden(g, dev='cuda'):
	from find_amd.model import NeuralDisplacementField
	m = NeuralDisplacementField(template_mesh_loc=None, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True,
								train_size=4, val_size=2, shapevec_size=100, texvec_size=100, posevec_size=100)
	sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('sd/')}
	m.load_state_dict(sd, strict=True)
	np.testing.assert_array_equal(m.encoder[0]._B.numpy(), g['B'])
	return m.to(dev)


def _t(a, dev='cuda'):
	return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_library_loads_and_reports_gfx950():
	from find_amd import _lib
	L = _lib.lib()
	assert L.find_abi_version() == _lib.ABI_VERSION
	assert L.find_build_arch() == b'gfx950'


@pytest.mark.parametrize('case', list('abcdef'))
def test_forward_matches_reference_golden(golden_main, case):
	m = _model_from_golden(golden_main)
	g = {k: _t(golden_main[f'fwd/{case}/{k}']) for k in ['pos', 'shapevec', 'texvec', 'posevec', 'disp', 'col']}
	with torch.no_grad():
		res = m(g['pos'], shapevec=g['shapevec'], texvec=g['texvec'], posevec=g['posevec'])
	assert res['disp'].shape == g['disp'].shape and res['col'].shape == g['col'].shape
	ed = (res['disp'] - g['disp']).abs().max().item()
	ec = (res['col'] - g['col']).abs().max().item()
	assert ed < TOL and ec < TOL, (case, ed, ec)
	assert ed < 2e-5 and ec < 2e-5, f'fp32 MFMA path unexpectedly loose: {ed} {ec}'


@pytest.mark.parametrize('case', list('abdf'))
def test_backward_matches_reference_golden(golden_main, case):
	m = _model_from_golden(golden_main)
	lat = {k: _t(golden_main[f'fwd/{case}/{k}']).requires_grad_(True) for k in ['shapevec', 'texvec', 'posevec']}
	res = m(_t(golden_main[f'fwd/{case}/pos']), **lat)
	loss = (res['disp'] ** 2).sum() + (res['col'] ** 2).sum()
	loss.backward()
	ref_loss = float(golden_main[f'grad/{case}/loss'])
	assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss)
	for k in lat:
		ref = golden_main[f'grad/{case}/{k}']
		err = np.abs(lat[k].grad.cpu().numpy() - ref).max()
		assert err < TOL * max(1.0, np.abs(ref).max()), (case, k, err)
	for k, prm in m.named_parameters():
		key = f'grad/{case}/sd/{k}'
		if key not in golden_main:
			assert prm.grad is None or k.split('.')[0] in ('shapevec', 'texvec', 'posevec', 'reg', 'shapevec_val', 'texvec_val', 'posevec_val', 'reg_val'), k
			continue
		got = prm.grad.cpu().numpy()
		ref = golden_main[key]
		if ref.shape != got.shape:  # strided subsample (tests/golden/make_golden_mlp.py GRAD_STRIDE)
			got = got.reshape(-1)[::17]
		scale = max(1.0, float(np.abs(ref).max()))
		err = float(np.abs(got - ref).max())
		assert err < TOL * scale, (case, k, err, scale)


def test_variants_match_reference_golden(golden_variants):
	from find_amd.model import NeuralDisplacementField
	kws = {
		'ttf': dict(use_shapevec=True, use_texvec=True, use_posevec=False, shapevec_size=100, texvec_size=100, posevec_size=100),
		'fff': dict(use_shapevec=False, use_texvec=False, use_posevec=False, shapevec_size=100, texvec_size=100, posevec_size=100),
		'sizes': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=64, texvec_size=32, posevec_size=64),
		'depth2': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=100, texvec_size=100, posevec_size=100,
					   depth=2, dispdepth=2, coldepth=1, sigma=5),
		'avgcol': dict(use_shapevec=True, use_texvec=True, use_posevec=True, shapevec_size=100, texvec_size=100, posevec_size=100,
					   use_avg_colour=True),
	}
	for name, kw in kws.items():
		m = NeuralDisplacementField(template_mesh_loc=None, device='cpu', train_size=3, val_size=1, **kw)
		gen = torch.Generator().manual_seed(1234)
		with torch.no_grad():
			m.mlp_disp[-1].weight.copy_(torch.randn(m.mlp_disp[-1].weight.shape, generator=gen) * 0.01)
			m.mlp_disp[-1].bias.copy_(torch.randn(m.mlp_disp[-1].bias.shape, generator=gen) * 0.01)
			if kw.get('use_avg_colour'):
				m.avg_col.copy_(torch.tensor([0.1, -0.2, 0.05]))
		m = m.to('cuda')
		lat = {k: _t(golden_variants[f'{name}/{k}']) for k in ['shapevec', 'texvec', 'posevec'] if f'{name}/{k}' in golden_variants}
		with torch.no_grad():
			res = m(_t(golden_variants[f'{name}/pos']), **lat)
		ed = (res['disp'].cpu() - torch.from_numpy(golden_variants[f'{name}/disp'])).abs().max().item()
		ec = (res['col'].cpu() - torch.from_numpy(golden_variants[f'{name}/col'])).abs().max().item()
		assert ed < TOL and ec < TOL, (name, ed, ec)


def test_latent_width_mismatch_raises(golden_main):
	m = _model_from_golden(golden_main)
	pos = torch.zeros(2, 8, 3, device='cuda')
	with pytest.raises(RuntimeError):
		m(pos, shapevec=torch.zeros(2, 100, device='cuda'))  # tex/pose missing: the reference fails on the matmul shape too


def test_cpu_tensors_fail_loudly(golden_main):
	m = _model_from_golden(golden_main, dev='cpu')
	with pytest.raises(RuntimeError, match='no CPU fallback'):
		m(torch.zeros(1, 4, 3), shapevec=torch.zeros(1, 100), texvec=torch.zeros(1, 100), posevec=torch.zeros(1, 100))


def test_shared_trunk_equals_general_path_full_size(golden_main):
	"""C2-size property (16 feet x 6890 vertices): evaluating a batch-1 template through the shared trunk gives the
	same outputs and gradients as expanding it to one copy per foot (what the reference executes)."""
	m = _model_from_golden(golden_main)
	gen = torch.Generator().manual_seed(3)
	N, V = 16, 6890
	ext = torch.tensor([0.12, 0.045, 0.04])
	pos1 = ((torch.rand(1, V, 3, generator=gen) * 2 - 1) * ext).cuda()
	lat = [(torch.randn(N, 100, generator=gen) * 0.1).cuda() for _ in range(3)]
	outs, grads = [], []
	for pos in (pos1, pos1.expand(N, -1, -1).contiguous()):
		m.zero_grad()
		l3 = [x.clone().requires_grad_(True) for x in lat]
		res = m(pos, shapevec=l3[0], texvec=l3[1], posevec=l3[2])
		w = torch.linspace(0.5, 1.5, N * V * 3, device='cuda').reshape(N, V, 3)
		((res['disp'] * w).sum() * 10 + (res['col'] * w).sum()).backward()
		outs.append((res['disp'].detach(), res['col'].detach()))
		grads.append([x.grad.clone() for x in l3] + [prm.grad.clone() for prm in m._weights()])
	assert (outs[0][0] - outs[1][0]).abs().max() < 1e-5
	assert (outs[0][1] - outs[1][1]).abs().max() < 1e-5
	for a, b in zip(grads[0], grads[1]):
		scale = max(1.0, b.abs().max().item())
		assert (a - b).abs().max().item() < 2e-4 * scale


def test_backward_is_linear_in_upstream_gradient(golden_main):
	m = _model_from_golden(golden_main)
	pos = _t(golden_main['fwd/a/pos'])
	lat = {k: _t(golden_main[f'fwd/a/{k}']) for k in ['shapevec', 'texvec', 'posevec']}
	gen = torch.Generator().manual_seed(5)
	g1 = torch.randn(2, 1000, 3, generator=gen).cuda()
	g2 = torch.randn(2, 1000, 3, generator=gen).cuda()

	def grads(gd, gc):
		m.zero_grad()
		res = m(pos, **lat)
		torch.autograd.backward([res['disp'], res['col']], [gd, gc])
		return [prm.grad.clone() for prm in m._weights()]

	a = grads(g1, g2)
	b = grads(2 * g1, 2 * g2)
	c = grads(g1, torch.zeros_like(g2))
	d = grads(torch.zeros_like(g1), g2)
	for x, y, u, v in zip(a, b, c, d):
		s = max(1.0, x.abs().max().item())
		assert (2 * x - y).abs().max().item() < 1e-4 * s
		assert (u + v - x).abs().max().item() < 1e-4 * s


def test_only_col_gradient_skips_disp_head(golden_main):
	"""TextureLossGTSpace (losses.py:51-57) back-propagates through `col` only."""
	m = _model_from_golden(golden_main)
	pos = _t(golden_main['fwd/d/pos'])
	lat = {k: _t(golden_main[f'fwd/d/{k}']).requires_grad_(True) for k in ['shapevec', 'texvec', 'posevec']}
	res = m(pos, **lat)
	(res['col'] ** 2).sum().backward()
	# a head nothing reads gets NO gradient, as in the reference (pinned: tests/golden/texture_loss.npz 'no_grad') -- not a tensor of zeros
	assert all(p.grad is None for p in m.mlp_disp.parameters())
	assert lat['shapevec'].grad is None and lat['posevec'].grad is None
	assert m.mlp_col[0].weight.grad.abs().max().item() > 0.0
	# oracle check of the col-only gradient
	sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype == torch.float32 and k.split('.')[0] in ('base', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	r = mlp_ref.mlp_forward(sd, m.encoder[0]._B, pos.cpu(), *(lat[k].detach().cpu() for k in ['shapevec', 'texvec', 'posevec']))
	(r['col'] ** 2).sum().backward()
	for k in ['base.0.weight', 'base.8.bias', 'mlp_col.0.weight', 'mlp_col.6.weight']:
		ref = sd[k].grad
		got = dict(m.named_parameters())[k].grad.cpu()
		assert (got - ref).abs().max().item() < TOL * ref.abs().max().item(), k   # relative to the tensor's largest entry


def test_registration_forward_backward_vs_oracle():
	from find_amd import functional as FN
	gen = torch.Generator().manual_seed(9)
	for vb, N, V in [(1, 3, 1500), (3, 3, 257), (1, 1, 1)]:
		verts = torch.randn(vb, V, 3, generator=gen) * 0.1
		disp = (torch.randn(N, V, 3, generator=gen) * 0.01).requires_grad_(True)
		reg = torch.cat([torch.rand(N, 3, generator=gen) * 0.02 - 0.01, torch.rand(N, 3, generator=gen) * 0.6 - 0.3,
						 torch.rand(N, 3, generator=gen) * 0.2 + 0.9], dim=1).requires_grad_(True)
		gout = torch.randn(N, V, 3, generator=gen)
		ref = mlp_ref.registration(verts.expand(N, -1, -1), disp, reg)
		ref.backward(gout)
		d2 = disp.detach().cuda().requires_grad_(True)
		r2 = reg.detach().cuda().requires_grad_(True)
		out = FN.register_points(verts.cuda(), d2, r2)
		out.backward(gout.cuda())
		assert (out.detach().cpu() - ref.detach()).abs().max().item() < 1e-6
		assert (d2.grad.cpu() - disp.grad).abs().max().item() < 1e-5
		scale = max(1.0, reg.grad.abs().max().item())
		assert (r2.grad.cpu() - reg.grad).abs().max().item() < 1e-4 * scale, (r2.grad.cpu(), reg.grad)


def test_get_meshes_matches_oracle(golden_main):
	m = _model_from_golden(golden_main)
	gen = torch.Generator().manual_seed(4)
	V = 333
	tv = (torch.rand(V, 3, generator=gen) * 2 - 1) * torch.tensor([0.12, 0.045, 0.04])
	faces = torch.randint(0, V, (600, 3), generator=gen)
	m.set_template(tv.cuda(), faces.cuda())
	N = 4
	sv, tx, pv = [(torch.randn(N, 100, generator=gen) * 0.1) for _ in range(3)]
	reg = torch.cat([torch.rand(N, 3, generator=gen) * 0.02 - 0.01, torch.rand(N, 3, generator=gen) * 0.2 - 0.1,
					 torch.rand(N, 3, generator=gen) * 0.2 + 0.9], dim=1)
	with torch.no_grad():
		res = m.get_meshes(shapevec=sv.cuda(), reg=reg.cuda(), texvec=tx.cuda(), posevec=pv.cuda())
		sd = {k: v.cpu() for k, v in m.state_dict().items()}
		ref = mlp_ref.get_meshes_verts(sd, m.encoder[0]._B, tv[None], sv, reg, tx, pv)
	assert set(res) >= {'meshes', 'offsets', 'verts', 'disp', 'col'}
	assert (res['verts'].cpu() - ref['verts']).abs().max().item() < TOL
	assert (res['col'].cpu() - ref['col']).abs().max().item() < TOL
	assert len(res['meshes']) == N
	assert res['meshes'].verts_padded().shape == (N, V, 3)
	assert torch.equal(res['meshes'].faces_padded()[2].cpu(), faces)
	assert (res['meshes'].textures.verts_features_padded() - res['col']).abs().max().item() == 0


@pytest.mark.gpu
def test_latent_gather_matches_indexing_and_scatters_deterministically():
	"""LatentVector lookup (model.py:131-152): rows and gradient identical to table[idx], duplicates and negatives included."""
	from find_amd import functional as FN
	dev = torch.device('cuda:0')
	g = torch.Generator().manual_seed(3)
	table = torch.randn(37, 100, generator=g).to(dev).requires_grad_(True)
	idx = torch.tensor([5, 0, 36, 5, -1, 17, 5, 0], device=dev)
	out = FN.latent_gather(table, idx)
	ref_t = table.detach().clone().requires_grad_(True)
	ref = ref_t[idx]
	assert torch.equal(out, ref)
	w = torch.randn(out.shape, generator=g).to(dev)
	(out * w).sum().backward()
	(ref * w).sum().backward()
	# sums of <= 3 terms in index order: equal to fp32 rounding of any order within 1 ulp; compare tightly
	assert torch.allclose(table.grad, ref_t.grad, rtol=1e-6, atol=1e-7)
	assert torch.count_nonzero(table.grad[1]) == 0
	g1 = table.grad.clone()
	table.grad = None
	(FN.latent_gather(table, idx) * w).sum().backward()
	assert torch.equal(g1, table.grad)
	bad = FN.latent_gather(table.detach(), torch.tensor([40], device=dev))
	assert torch.isnan(bad).all()


@pytest.mark.gpu
def test_c5_dense_template_shared_equals_general(golden_main):
	"""C5-size property (50 002-vertex dense template, fp32): shared-trunk evaluation == one template copy per foot, outputs
	and every gradient, at a row count where the long-row-range kernels (gemm4 units, dw2 runs) take their large-V paths."""
	m = _model_from_golden(golden_main)
	gen = torch.Generator().manual_seed(8)
	N, V = 3, 50002
	ext = torch.tensor([0.12, 0.045, 0.04])
	pos1 = ((torch.rand(1, V, 3, generator=gen) * 2 - 1) * ext).cuda()
	lat = [(torch.randn(N, 100, generator=gen) * 0.1).cuda() for _ in range(3)]
	outs, grads = [], []
	for pos in (pos1, pos1.expand(N, -1, -1).contiguous()):
		m.zero_grad()
		l3 = [x.clone().requires_grad_(True) for x in lat]
		res = m(pos, shapevec=l3[0], texvec=l3[1], posevec=l3[2])
		w = torch.linspace(0.5, 1.5, N * V * 3, device='cuda').reshape(N, V, 3)
		((res['disp'] * w).sum() * 10 + (res['col'] * w).sum()).backward()
		outs.append((res['disp'].detach(), res['col'].detach()))
		grads.append([x.grad.clone() for x in l3] + [prm.grad.clone() for prm in m._weights()])
	assert (outs[0][0] - outs[1][0]).abs().max() < 1e-5
	assert (outs[0][1] - outs[1][1]).abs().max() < 1e-5
	for a, b in zip(grads[0], grads[1]):
		scale = max(1.0, b.abs().max().item())
		assert (a - b).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize('n_feet,n_pts', [(1, 70), (1, 2100), (1, 6890), (3, 1000), (16, 6890)])
def test_linear_relu_kernel_every_routing(n_feet, n_pts):
	"""find_linear_relu_fwd (the bench's dominant kernel; model.py:421-426 `Linear` + `ReLU`) at row counts that route to each of
	the tile kernels -- 64-row LDS-DMA tiles, W-resident column quarters, W-resident column halves -- against a float64 matmul."""
	import ctypes
	from find_amd import _lib
	L = _lib.lib()
	gen = torch.Generator().manual_seed(n_feet * 7919 + n_pts)
	x = torch.randn(n_feet * n_pts, 256, generator=gen)
	w = torch.randn(256, 256, generator=gen) / 16
	b = torch.randn(256, generator=gen)
	xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
	y = torch.full_like(xd, float('nan'))
	_lib.check(L.find_linear_relu_fwd(_lib.ctx(), _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), n_feet, n_pts, _lib.ptr(y),
									  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'find_linear_relu_fwd')
	want = np.maximum(x.numpy().astype(np.float64) @ w.numpy().astype(np.float64).T + b.numpy().astype(np.float64), 0.0)
	got = y.cpu().numpy()
	assert np.isfinite(got).all()
	assert np.abs(got - want).max() < TOL


def test_backward_is_bit_reproducible_with_side_streams():
	"""The backward is meant to be deterministic (slab reduces, no float atomics).  Regression test for a fault that showed only
	with the weight-gradient kernels of several streams resident on one CU (mlp.hip, 'Co-residence fault'): a few dW elements off by ~1 % in
	some passes -- every pass when the dX GEMMs ran on gemm3 (gemm4 switched off through its thresholds), so that configuration is screened too."""
	from find_amd import _lib, synthetic
	dev = torch.device('cuda:0')
	n_feet = 4
	model = synthetic.make_model(1002, train_size=n_feet, val_size=2, device=dev)
	lat = synthetic.latents(n_feet, seed=0, device=dev)
	named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]

	def once():
		for _, p in named:
			p.grad = None
		lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
		res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
		((res['verts'] ** 2).sum() + (res['col'] ** 2).sum()).backward()
		torch.cuda.synchronize()
		out = {n: p.grad.detach().clone() for n, p in named if p.grad is not None}
		out.update({'latent.' + k: v.grad.detach().clone() for k, v in lv.items()})
		return out

	for gemm4, reduce_stream in ((1, 0), (0, 0), (1, 1)):
		# gemm4 = 0: every 256 -> 256 launch on the LDS-DMA ring kernel (gemm3) instead of the W-resident gemm4
		# reduce_stream = 1: the slab reduces of the large layers on their own stream with two alternating slab sets (off by default)
		_lib.set_tuning('gemm4_min_units', 1024 if gemm4 else 10 ** 12)
		_lib.set_tuning('gemm4_small', 64 if gemm4 else 0)
		_lib.set_tuning('reduce_stream', reduce_stream)
		try:
			ref = once()
			if reduce_stream:
				assert all(torch.equal(ref[n], first[n]) for n in ref), 'the reduce stream changed a gradient'
			elif gemm4:
				first = ref
			for rep in range(25):
				got = once()
				bad = [n for n in ref if not torch.equal(got[n], ref[n])]
				assert not bad, f'gemm4={gemm4}, reduce_stream={reduce_stream}, pass {rep}: gradients of {bad} differ from the first pass'
		finally:
			_lib.set_tuning('gemm4_min_units', 1024)
			_lib.set_tuning('gemm4_small', 64)
			_lib.set_tuning('reduce_stream', 0)


def _wgrad(n_feet, n_pts, seed, sparse=False):
	"""Runs find_linear_wgrad on seeded dz, x; returns (dz, x, dw, db) on the host."""
	import ctypes
	from find_amd import _lib
	L = _lib.lib()
	gen = torch.Generator().manual_seed(seed)
	rows = n_feet * n_pts
	dz = torch.randn(rows, 256, generator=gen) * 0.1
	x = torch.relu(torch.randn(rows, 256, generator=gen))
	if sparse:
		dz = dz * (torch.rand(rows, 256, generator=gen) < 0.3)
	dzd, xd = dz.cuda(), x.cuda()
	dw = torch.full((256, 256), float('nan'), device='cuda')
	db = torch.full((256,), float('nan'), device='cuda')
	nbytes = L.find_linear_wgrad_scratch_bytes(n_feet)
	scratch = torch.empty(nbytes // 4, dtype=torch.float32, device='cuda')
	_lib.check(L.find_linear_wgrad(_lib.ctx(), _lib.ptr(dzd), _lib.ptr(xd), n_feet, n_pts, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(scratch), nbytes,
								   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'find_linear_wgrad')
	torch.cuda.synchronize()
	return dz, x, dw.cpu(), db.cpu()


@pytest.mark.parametrize('n_feet,n_pts', [(1, 1), (5, 15), (1, 70), (3, 1002), (2, 6890), (16, 6890)])
def test_linear_wgrad_kernel(n_feet, n_pts):
	"""find_linear_wgrad (dw2_kernel + the slab reduce: `weight.grad` / `bias.grad` of an nn.Linear, model.py:255-257) against float64
	at row counts with no full 16-row chunk, with leftover rows, and at the C2 shape."""
	dz, x, dw, db = _wgrad(n_feet, n_pts, seed=n_feet * 131 + n_pts, sparse=(n_pts == 1002))
	want = dz.double().t() @ x.double()
	wantb = dz.double().sum(0)
	assert torch.isfinite(dw).all() and torch.isfinite(db).all()
	assert (dw.double() - want).abs().max().item() < TOL * max(1.0, want.abs().max().item())
	assert (db.double() - wantb).abs().max().item() < TOL * max(1.0, wantb.abs().max().item())


def test_argument_checks_fail_before_the_launch(golden_main):
	"""Shape errors the reference meets in torch.cat / expand (model.py:403-437) must not become out-of-bounds device reads here."""
	m = _model_from_golden(golden_main)
	pos = torch.zeros(2, 8, 3, device='cuda')
	z = lambda n, w=100: torch.zeros(n, w, device='cuda')
	with pytest.raises(RuntimeError, match='different batch sizes'):
		m(pos, shapevec=z(2), texvec=z(3), posevec=z(2))
	with pytest.raises((RuntimeError, ValueError)):
		m(torch.zeros(2, 8, 2, device='cuda'), shapevec=z(2), texvec=z(2), posevec=z(2))
	with pytest.raises(RuntimeError, match='does not match latent batch'):
		m(torch.zeros(3, 8, 3, device='cuda'), shapevec=z(2), texvec=z(2), posevec=z(2))
	with pytest.raises(IndexError):
		m.shapevec[torch.tensor([0, 7])]          # 4 rows: a host index is range-checked like torch's indexing
	assert m.shapevec[torch.tensor([0, 3])].shape == (2, 100)


def test_context_reports_the_device_and_keeps_its_knobs():
	"""find_ctx (include/find_hip.h): per-device state instead of process globals -- device facts, knob round trip, error on unknown keys,
	and the event budget of a backward call stays far below the ring."""
	import ctypes
	from find_amd import _lib, synthetic
	L = _lib.lib()
	assert _lib.get_tuning('num_cus') == 256 and _lib.get_tuning('lds_bytes') >= 160 * 1024 and _lib.get_tuning('device') == torch.cuda.current_device()
	prev = _lib.get_tuning('dw2_min_cps')
	_lib.set_tuning('dw2_min_cps', 4)
	assert _lib.get_tuning('dw2_min_cps') == 4
	_lib.set_tuning('dw2_min_cps', prev)
	with pytest.raises(RuntimeError, match='unknown key'):
		_lib.set_tuning('no_such_knob', 1)
	with pytest.raises(RuntimeError, match='out of range'):
		_lib.set_tuning('bwd_streams', 7)
	# the laboratory's keys and wrong-result bits exist in libfind_hip_diag.so only (include/find_hip_diag.h)
	assert _lib.get_tuning('diag') == 0
	for key in ('gemm7', 'x3_abl', 'dbg', 'dw2_verify'):
		with pytest.raises(RuntimeError, match='unknown key'):
			_lib.set_tuning(key, 0)
	with pytest.raises(RuntimeError, match='out of range'):
		_lib.set_tuning('dw_lds_free', 2)
	for bits in (1, 2, 4, 8, 512, 1024, 64):
		with pytest.raises(RuntimeError, match='result-preserving'):
			_lib.set_tuning('ablate', bits)
	_lib.set_tuning('ablate', 16 | 32 | 128)
	assert _lib.get_tuning('ablate') == 176
	_lib.set_tuning('ablate', 0)
	# a second context on the same device is independent of the first
	h = ctypes.c_void_p()
	_lib.check(L.find_ctx_create(torch.cuda.current_device(), ctypes.byref(h)), 'find_ctx_create')
	try:
		_lib.check(L.find_ctx_set(h, b'fused_max_units', 0), 'find_ctx_set')
		v = ctypes.c_int64()
		_lib.check(L.find_ctx_get(h, b'fused_max_units', ctypes.byref(v)), 'find_ctx_get')
		assert v.value == 0 and _lib.get_tuning('fused_max_units') == 512
	finally:
		_lib.check(L.find_ctx_destroy(h), 'find_ctx_destroy')
	model = synthetic.make_model(1002, train_size=3, val_size=1, device='cuda')
	lat = synthetic.latents(3, seed=0, device='cuda')
	res = model.get_meshes(shapevec=lat['shapevec'].requires_grad_(True), reg=lat['reg'], texvec=lat['texvec'], posevec=lat['posevec'])
	(res['verts'].sum() + res['col'].sum()).backward()
	torch.cuda.synchronize()
	assert 0 < _lib.get_tuning('events_per_call_max') < 128


def test_two_passes_in_one_backward_fold_their_weight_gradients(golden_main):
	"""FIND's step runs the MLP twice on the same weights (main pass + texture pass).  The second backward of one backward() call adds
	its weight gradients to the first's with one multi-tensor launch and reports none of its own: the parameters' .grad must equal the
	sum of the two passes differentiated separately, and a later, separate backward() must accumulate on top as usual."""
	from find_amd import functional as FN
	m = _model_from_golden(golden_main)
	g = torch.Generator().manual_seed(3)
	lat = {k: torch.randn(2, 100, generator=g).cuda() * 0.1 for k in ['shapevec', 'texvec', 'posevec']}
	pos1 = (torch.rand(1, 700, 3, generator=g) * 0.2).cuda()
	pos2 = (torch.rand(2, 300, 3, generator=g) * 0.2).cuda()
	params = [p for p in m.parameters() if p.requires_grad and p.dim() <= 2 and p.shape[0] in (3, 256)]

	def l1():
		r = m(pos1, **lat)
		return (r['disp'] ** 2).sum() + (r['col'] ** 2).sum()

	def l2():
		return (m(pos2, **lat)['col'] ** 2).sum() * 0.5

	m.zero_grad(set_to_none=True)
	l1().backward()
	a = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
	m.zero_grad(set_to_none=True)
	l2().backward()
	b = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
	m.zero_grad(set_to_none=True)
	FN._PENDING_WGRADS.clear()
	(l1() + l2()).backward()
	folded = 0
	for n, p in m.named_parameters():
		if n not in a:
			continue
		want = a[n] + b.get(n, 0)
		assert (p.grad - want).abs().max().item() <= 2e-6 * max(1e-3, want.abs().max().item()), n
		folded += 1
	assert folded >= 26
	# a later backward() is a different graph task: nothing folds into stale tensors, autograd accumulates into .grad
	l2().backward()
	for n, p in m.named_parameters():
		if n in a and n in b:
			want = a[n] + 2 * b[n]
			assert (p.grad - want).abs().max().item() <= 4e-6 * max(1e-3, want.abs().max().item()), n
	assert params


@pytest.mark.parametrize('shape', [(2, 300), (16, 1000), (1, 6890)])
def test_colour_head_alone_equals_the_full_forward(golden_main, shape):
	"""model(..., want=('col',)) -- what the texture loss asks for -- skips the displacement head in the forward: same colours, same
	gradients as the full forward differentiated through its colour output only, none for the displacement head's parameters."""
	m = _model_from_golden(golden_main)
	n, v = shape
	g = torch.Generator().manual_seed(n * 1000 + v)
	lat = {k: (torch.randn(n, 100, generator=g) * 0.1).cuda() for k in ['shapevec', 'texvec', 'posevec']}
	pos = (torch.rand(n, v, 3, generator=g) * 0.2).cuda()
	wgt = torch.randn(n, v, 3, generator=g).cuda()
	out = {}
	for want in (('disp', 'col'), ('col',)):
		m.zero_grad(set_to_none=True)
		lv = {k: t.clone().requires_grad_(True) for k, t in lat.items()}
		res = m(pos, **lv, want=want)
		assert set(res) == set(want)
		(res['col'] * wgt).sum().backward()
		out[want] = (res['col'].detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, {k: t.grad for k, t in lv.items()})
	(c0, g0, l0), (c1, g1, l1) = out[('disp', 'col')], out[('col',)]
	assert torch.equal(c0, c1)
	assert set(g0) == set(g1)
	for k in g0:
		assert (g0[k] - g1[k]).abs().max().item() <= 1e-6 * max(1e-3, g0[k].abs().max().item()), k
		assert not k.startswith('mlp_disp')
	for k in l0:
		assert (l0[k] is None) == (l1[k] is None)
		if l0[k] is not None:
			assert (l0[k] - l1[k]).abs().max().item() <= 1e-6 * max(1e-3, l0[k].abs().max().item()), k
	with pytest.raises(ValueError):
		m(pos, **lat, want=())


def test_latent_gather_many_and_weighted_terms():
	"""One launch for the four latent-table lookups of a step and one for the loss weighting: same values and gradients as the
	per-table gather and as raw * weight / Python's sum()."""
	from find_amd import functional as FN
	g = torch.Generator().manual_seed(8)
	tables = [torch.randn(n, d, generator=g).cuda().requires_grad_(True) for n, d in [(5, 100), (9, 100), (5, 100), (9, 9)]]
	idxs = [torch.tensor(i, device='cuda') for i in ([0, 4, 4], [8, 2, 2], [1, 1, 1], [-1, 3, 0])]
	outs = FN.latent_gather_many(tables, idxs)
	w = [torch.randn(o.shape, generator=g).cuda() for o in outs]
	sum((o * x).sum() for o, x in zip(outs, w[:3])).backward()   # the fourth lookup gets no gradient: its table's gradient is zero
	got = [t.grad.clone() for t in tables]
	for t in tables:
		t.grad = None
	ref = [t[i] for t, i in zip(tables, idxs)]
	for o, r in zip(outs, ref):
		assert torch.equal(o, r)
	sum((o * x).sum() for o, x in zip(ref, w[:3])).backward()
	for k, (a, t) in enumerate(zip(got, tables)):
		want = t.grad if t.grad is not None else torch.zeros_like(t)
		assert (a - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item()), k
	assert torch.isnan(FN.latent_gather_many(tables[:2], [torch.tensor([7], device='cuda'), torch.tensor([0], device='cuda')])[0]).all()   # out of range: NaN row
	# weighted terms
	raws = [torch.tensor(v, device='cuda', requires_grad=True) for v in (0.25, 3.0, 1e-3)]
	total, scaled = FN.weighted_terms(raws, [10000.0, 1000.0, 1.0])
	assert abs(total.item() - (2500.0 + 3000.0 + 1e-3)) < 1e-3 and abs(scaled[1].item() - 3000.0) < 1e-3
	(total * 2.0 + scaled[2]).backward()
	assert [round(r.grad.item(), 4) for r in raws] == [20000.0, 2000.0, 3.0]


@pytest.mark.parametrize('n_feet,variant', [(16, 'bf16x3'), (16, 'dw4'), (1, 'dw4'), (16, 'dw2_whole_file'), (16, 'fp16')])
def test_backward_is_bit_reproducible_under_co_residence_stress(n_feet, variant):
	"""The stress configuration that made round 1's rare fault happen in every pass (mlp.hip, 'Co-residence fault'): the slab reduces are
	replaced by an LDS-free, slow kernel ("reduce_exclusive" = 2), so that reduces of earlier layers stay resident on the CUs beside the
	weight-gradient kernels of later layers.  The victims were waves with the full accumulator set in a 300-328 register allocation (dw2_kernel as round 1
	had it: ~6 wrong weight gradients per pass).  Every weight-gradient kernel the product can select -- the default bf16x3 arithmetic's dw6 (whole
	register file claimed) beside gemm7 (<= 256 registers), the fp32-MFMA path's dw4 (<= 256 registers), dw2 with the whole register file claimed, the
	fp16 mode's dw3 likewise -- must give bit-identical gradients in every pass, with and without the LDS reservation."""
	from find_amd import _lib, synthetic
	from find_amd import functional as F
	dev = torch.device('cuda:0')
	model = synthetic.make_model(6890, train_size=n_feet, val_size=2, device=dev)
	lat = synthetic.latents(n_feet, seed=0, device=dev)
	named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]

	def once():
		for _, p in named:
			p.grad = None
		res = model.get_meshes(shapevec=lat['shapevec'], reg=lat['reg'], texvec=lat['texvec'], posevec=lat['posevec'])
		((res['verts'] ** 2).sum() + (res['col'] ** 2).sum()).backward()
		torch.cuda.synchronize()
		return {n: p.grad.detach().clone() for n, p in named if p.grad is not None}

	assert _lib.get_tuning('dw_lds_free') == 1 and _lib.get_tuning('lds_exclusive') == 0
	_lib.set_tuning('reduce_exclusive', 2)
	_lib.set_tuning('dw_lds_free', 0 if variant == 'dw2_whole_file' else 1)
	prev_prec = F.set_mlp_precision({'fp16': 'fp16', 'bf16x3': 'bf16x3'}.get(variant, 'fp32'))
	try:
		for excl in (0, 1):
			_lib.set_tuning('lds_exclusive', excl)
			ref = once()
			for rep in range(30):
				got = once()
				bad = [n for n in ref if not torch.equal(got[n], ref[n])]
				assert not bad, f'{variant}, lds_exclusive={excl}, pass {rep}: gradients of {bad} differ from the first pass'
	finally:
		_lib.set_tuning('reduce_exclusive', 0)
		_lib.set_tuning('lds_exclusive', 0)
		_lib.set_tuning('dw_lds_free', 1)
		F.set_mlp_precision(prev_prec)


@pytest.mark.parametrize('shape', [(16, 1000), (9, 1000), (5, 2100)])
def test_fused_chain_64_row_tiles_equal_the_32_row_tiles(golden_main, shape):
	"""A fused chain over more 32-row blocks than the chip has CUs runs on 64-row tiles (fused_chain_kernel<2>; 1000 rows per foot leave a
	40-row last tile, 2100 a 52-row one): every output row and every gradient must be bit-identical to the 32-row tiles (ablate bit 128),
	which the golden-vector tests pin -- a row's products and their order do not depend on the tile it sits in."""
	from find_amd import _lib
	m = _model_from_golden(golden_main)
	n, v = shape
	g = torch.Generator().manual_seed(n * 77 + v)
	lat = {k: (torch.randn(n, 100, generator=g) * 0.1).cuda() for k in ['shapevec', 'texvec', 'posevec']}
	pos = (torch.rand(n, v, 3, generator=g) * 0.2).cuda()
	wd = torch.randn(n, v, 3, generator=g).cuda()
	wc = torch.randn(n, v, 3, generator=g).cuda()
	out = {}
	try:
		for tiles, knob in (('64', 0), ('32', 128)):
			_lib.set_tuning('ablate', knob)
			m.zero_grad(set_to_none=True)
			lv = {k: t.clone().requires_grad_(True) for k, t in lat.items()}
			res = m(pos, **lv)
			((res['disp'] * wd).sum() + (res['col'] * wc).sum()).backward()
			torch.cuda.synchronize()
			out[tiles] = (res['disp'].detach().clone(), res['col'].detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
						  {k: t.grad.clone() for k, t in lv.items()})
	finally:
		_lib.set_tuning('ablate', 0)
	a, b = out['64'], out['32']
	assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
	assert a[0].abs().max().item() > 0 and a[1].abs().max().item() > 0
	for k in b[2]:
		assert torch.equal(a[2][k], b[2][k]), k
	for k in b[3]:
		assert torch.equal(a[3][k], b[3][k]), k


def test_side_streams_are_bound_to_queues_beside_the_callers():
	"""HIP maps streams onto a few hardware queues in creation order; two streams on one queue run in order.  After a call that forks, the
	context's side streams Q, T1, T2 must each sit on a queue of their own that is not the caller's, and the reduce stream R on T2's -- also
	when the process created a pile of streams first (what torch.distributed / RCCL does before the first MLP call of a rank)."""
	import ctypes
	from find_amd import _lib, synthetic
	L = _lib.lib()
	clutter = [torch.cuda.Stream() for _ in range(5)]   # (kept alive: they hold their queues)
	for s in clutter:
		with torch.cuda.stream(s):
			torch.zeros(8, device='cuda').add_(1)
	torch.cuda.synchronize()
	h = ctypes.c_void_p()
	_lib.check(L.find_ctx_create(torch.cuda.current_device(), ctypes.byref(h)), 'find_ctx_create')
	try:
		model = synthetic.make_model(1002, train_size=2, val_size=1, device='cuda')
		lat = synthetic.latents(2, seed=0, device='cuda')
		prev = _lib._ctx.get(torch.cuda.current_device())
		_lib._ctx[torch.cuda.current_device()] = h   # the fresh context serves this model's calls
		try:
			lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
			res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
			((res['verts'] ** 2).sum() + (res['col'] ** 2).sum()).backward()
			torch.cuda.synchronize()
		finally:
			_lib._ctx[torch.cuda.current_device()] = prev
		g = (ctypes.c_int32 * 5)()
		_lib.check(L.find_ctx_stream_groups(h, _lib.current_stream(torch.device('cuda')), ctypes.cast(g, ctypes.c_void_p)), 'find_ctx_stream_groups')
		caller, q, t1, t2, r = list(g)
		assert caller == 0
		assert len({caller, q, t1, t2}) == 4, list(g)
		assert r == t2, list(g)
	finally:
		_lib.check(L.find_ctx_destroy(h), 'find_ctx_destroy')
	del clutter


def test_stream_beside_picks_a_stream_on_the_asked_hardware_queue():
	"""find_ctx_stream_beside (what ModelWithLoss's second stream is chosen with): among a dozen candidate streams the one returned runs
	beside the caller's stream and shares the hardware queue of the asked side stream -- checked against find_ctx_stream_groups with the
	candidate put in the caller's place --; bad arguments are refused; ModelWithLoss's own pick sits on Q's queue."""
	import ctypes
	from find_amd import _lib
	from find_amd.model_with_loss import _second_stream
	L = _lib.lib()
	dev = torch.device('cuda', torch.cuda.current_device())
	h = _lib.ctx(dev)
	main = torch.cuda.current_stream(dev)
	cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
	arr = (ctypes.c_void_p * len(cands))(*[c.cuda_stream for c in cands])
	idx = ctypes.c_int32(-7)
	for role in (0, 1, 2):
		_lib.check(L.find_ctx_stream_beside(h, ctypes.c_void_p(main.cuda_stream), arr, len(cands), role, ctypes.byref(idx)), 'find_ctx_stream_beside')
		assert 0 <= idx.value < len(cands), (role, idx.value)   # (twelve streams over four hardware queues: every queue is met)
		g = (ctypes.c_int32 * 5)()
		_lib.check(L.find_ctx_stream_groups(h, ctypes.c_void_p(cands[idx.value].cuda_stream), ctypes.cast(g, ctypes.c_void_p)), 'find_ctx_stream_groups')
		assert g[1 + role] == g[0], (role, list(g))                 # the candidate (group 0 here) shares side stream `role`'s queue
		g2 = (ctypes.c_int32 * 5)()
		_lib.check(L.find_ctx_stream_groups(h, ctypes.c_void_p(main.cuda_stream), ctypes.cast(g2, ctypes.c_void_p)), 'find_ctx_stream_groups')
		assert g2[1 + role] != g2[0]                                 # ... which is not the caller's
	assert L.find_ctx_stream_beside(h, ctypes.c_void_p(main.cuda_stream), arr, len(cands), 9, ctypes.byref(idx)) != 0
	assert L.find_ctx_stream_beside(h, ctypes.c_void_p(main.cuda_stream), None, len(cands), 0, ctypes.byref(idx)) != 0
	side = _second_stream(dev)
	g = (ctypes.c_int32 * 5)()
	_lib.check(L.find_ctx_stream_groups(h, ctypes.c_void_p(side.cuda_stream), ctypes.cast(g, ctypes.c_void_p)), 'find_ctx_stream_groups')
	assert g[1] == g[0], list(g)   # Q's queue
	torch.cuda.synchronize()


def test_unbound_side_streams_give_the_same_gradients():
	"""The fall-back layout of the side streams -- `bind_streams` = 0: the four streams as HIP created them, some of them on the caller's
	hardware queue (what the probe leaves when every candidate shares a queue) -- changes the schedule, never a number: a full-size
	backward through a context of that kind is bit-identical to the one through the default context."""
	import ctypes
	from find_amd import _lib, synthetic
	L = _lib.lib()
	dev = torch.cuda.current_device()
	model = synthetic.make_model(6890, train_size=4, val_size=1, device='cuda')
	lat = synthetic.latents(4, seed=3, device='cuda')

	def grads():
		model.zero_grad(set_to_none=True)
		lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
		res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
		((res['verts'] ** 2).sum() + (res['col'] ** 2).sum()).backward()
		torch.cuda.synchronize()
		out = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
		out.update({f'lat/{k}': v.grad.clone() for k, v in lv.items()})
		return out

	want = grads()
	h = ctypes.c_void_p()
	_lib.check(L.find_ctx_create(dev, ctypes.byref(h)), 'find_ctx_create')
	try:
		_lib.check(L.find_ctx_set(h, b'bind_streams', 0), 'find_ctx_set(bind_streams)')
		prev = _lib._ctx.get(dev)
		_lib._ctx[dev] = h
		try:
			got = grads()
		finally:
			_lib._ctx[dev] = prev
	finally:
		_lib.check(L.find_ctx_destroy(h), 'find_ctx_destroy')
	assert set(got) == set(want) and len(want) >= 30
	for n in want:
		assert torch.equal(got[n], want[n]), n


def test_deferred_weight_gradient_join_changes_no_gradient_and_is_joined_when_backward_returns(golden_main):
	"""The texture pass asks for its weight gradients to trail the backward (defer_wgrad_join=True -> find_ctx "defer_join"): the
	gradients read on the caller's stream right after backward() -- no device synchronisation in between -- equal those of the run that
	joins inside the call (bit for bit but for the latent-derived sums, below), are bit-identical from run to run, nothing is left pending on the context, and the second pass still folds into the first's gradients."""
	from find_amd import _lib
	from find_amd import functional as FN
	m = _model_from_golden(golden_main)
	g = torch.Generator().manual_seed(11)
	n_feet = 4
	lat = {k: torch.randn(n_feet, 100, generator=g).cuda() * 0.1 for k in ['shapevec', 'texvec', 'posevec']}
	pos_main = (torch.rand(1, 6890, 3, generator=g) * 0.2).cuda()          # template pass: shared trunk, displacement head read
	pos_tex = (torch.rand(n_feet, 1000, 3, generator=g) * 0.2).cuda()      # texture pass: per-foot samples, colour head only
	prev = FN.DEFER_WGRAD_JOIN
	out = {}
	try:
		for defer in (False, True, True):
			FN.DEFER_WGRAD_JOIN = defer
			m.zero_grad(set_to_none=True)
			FN._PENDING_WGRADS.clear()
			lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
			main = m(pos_main, **lv, want=('disp',))
			tex = m(pos_tex, **lv, want=('col',), defer_wgrad_join=True)
			((main['disp'] ** 2).sum() + (tex['col'] ** 2).sum()).backward()
			got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}   # (clones on the caller's stream, no synchronise before)
			got.update({f'lat/{k}': v.grad.clone() for k, v in lv.items()})
			assert _lib.get_tuning('pending', pos_main.device) == 0
			torch.cuda.synchronize()
			if defer in out:
				for n in got:
					assert torch.equal(got[n], out[defer][n]), (defer, n)   # run to run
			out[defer] = got
	finally:
		FN.DEFER_WGRAD_JOIN = prev
	assert set(out[True]) == set(out[False]) and len(out[True]) >= 26
	# Bit for bit, except what is built from the per-foot column sums of the colour head's first-layer gradient -- the latent columns of that
	# layer's weight gradient and the latent gradients: the deferred path takes those sums from the foot-sum kernel on the caller's stream, the
	# other from the weight-gradient slabs (another order of the same additions).
	for n in out[False]:
		a, b = out[True][n], out[False][n]
		if n.startswith('lat/'):
			assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item(), n
		elif n in ('mlp_col.0.weight', 'mlp_disp.0.weight'):
			assert torch.equal(a[:, :256], b[:, :256]), n
			assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item(), n
		else:
			assert torch.equal(a, b), n


def test_forward_and_backward_use_a_replaced_parameter_object(golden_main):
	"""the kernels read the module tree's CURRENT Parameters.  Replace one object (not its data) after a first call has
	cached the weight list: the next forward must equal the oracle's on the new weights and the gradient must land on the new object."""
	m = _model_from_golden(golden_main)
	g = {k: _t(golden_main[f'fwd/a/{k}']) for k in ['pos', 'shapevec', 'texvec', 'posevec']}
	with torch.no_grad():
		before = m(g['pos'], shapevec=g['shapevec'], texvec=g['texvec'], posevec=g['posevec'])
	gen = torch.Generator().manual_seed(3)
	old = m.mlp_col[2].weight
	new = torch.nn.Parameter((torch.randn(256, 256, generator=gen) / 16).cuda())
	m.mlp_col[2].weight = new
	res = m(g['pos'], shapevec=g['shapevec'], texvec=g['texvec'], posevec=g['posevec'])
	sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
	ref = mlp_ref.mlp_forward(sd, m.encoder[0]._B, g['pos'].cpu(), g['shapevec'].cpu(), g['texvec'].cpu(), g['posevec'].cpu())
	assert (res['col'].detach().cpu() - ref['col']).abs().max().item() < TOL
	assert (res['col'].detach() - before['col']).abs().max().item() > 1e-3   # (the replacement matters)
	(res['col'] ** 2).sum().backward()
	assert new.grad is not None and float(new.grad.abs().max()) > 0 and old.grad is None
