"""ctypes binding of libfind_hip.so (include/find_hip.h).  There is NO fallback: if the HIP library is
missing or a call fails, a RuntimeError is raised.

FIND_DIAG=1 in the environment loads libfind_hip_diag.so instead -- the laboratory build of the same sources (include/find_hip_diag.h:
fault reproducers, superseded kernels, timers, wrong-result ablation bits).  Only tools/ do that; tests, bench.py and
__graft_entry__ run the product."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
DIAG = os.environ.get('FIND_DIAG', '0') not in ('', '0')
LIB_PATH = os.path.join(_PKG, 'lib', 'libfind_hip_diag.so' if DIAG else 'libfind_hip.so')
MAX_LAYERS = 8
ABI_VERSION = 2

_lib = None


class MlpParams(Structure):
	_fields_ = [
		('width', c_int32), ('in_dim', c_int32), ('pe_size', c_int32), ('n_trunk', c_int32),
		('n_disp', c_int32), ('n_col', c_int32), ('lat_disp', c_int32), ('lat_col', c_int32),
		('B', c_void_p),
		('trunk_w', c_void_p * MAX_LAYERS), ('trunk_b', c_void_p * MAX_LAYERS),
		('disp_w', c_void_p * MAX_LAYERS), ('disp_b', c_void_p * MAX_LAYERS),
		('col_w', c_void_p * MAX_LAYERS), ('col_b', c_void_p * MAX_LAYERS),
		('avg_col', c_void_p),
		('precision', c_int32),   # 0 = context default, 1 = fp32, 2 = fp16 operands (find_hip.h)
	]


class MlpGrads(Structure):
	_fields_ = [
		('trunk_w', c_void_p * MAX_LAYERS), ('trunk_b', c_void_p * MAX_LAYERS),
		('disp_w', c_void_p * MAX_LAYERS), ('disp_b', c_void_p * MAX_LAYERS),
		('col_w', c_void_p * MAX_LAYERS), ('col_b', c_void_p * MAX_LAYERS),
		('lat_disp', c_void_p), ('lat_col', c_void_p),
	]


class RenderParams(Structure):
	_fields_ = [
		('image_h', c_int32), ('image_w', c_int32), ('fov_deg', c_float), ('znear', c_float), ('zfar', c_float),
		('sil_blur_radius', c_float), ('sil_sigma', c_float), ('sil_faces_per_pixel', c_int32),
		('rgb_sigma', c_float), ('rgb_gamma', c_float), ('background', c_float * 3), ('light_pos', c_float * 3),
		('ambient', c_float), ('diffuse', c_float), ('specular', c_float), ('shininess', c_float), ('z_clip', c_float),
	]


# name -> (restype, argtypes); mirrors include/find_hip.h one to one
_P = c_void_p
_I = c_int64
PROTOTYPES = {
	'find_abi_version': (c_int, []),
	'find_last_error': (c_char_p, []),
	'find_build_arch': (c_char_p, []),
	'find_ctx_create': (c_int, [c_int, POINTER(c_void_p)]),
	'find_ctx_destroy': (c_int, [_P]),
	'find_ctx_stream_groups': (c_int, [_P, _P, _P]),
	'find_ctx_stream_beside': (c_int, [_P, _P, _P, c_int32, c_int32, _P]),
	'find_ctx_set': (c_int, [_P, c_char_p, _I]),
	'find_ctx_get': (c_int, [_P, c_char_p, POINTER(c_int64)]),
	'find_ctx_join': (c_int, [_P, _P]),
	'find_render_switches': (c_int, [_I]),
	'find_mlp_ws_bytes': (c_int64, [POINTER(MlpParams), _I, _I, _I, c_int]),
	'find_mlp_fwd': (c_int, [_P, POINTER(MlpParams), _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, c_int, _P]),
	'find_mlp_bwd_scratch_bytes': (c_int64, [POINTER(MlpParams), _I, _I, _I]),
	'find_mlp_bwd': (c_int, [_P, POINTER(MlpParams), _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, POINTER(MlpGrads), _P]),
	'find_linear_relu_fwd': (c_int, [_P, _P, _P, _P, _I, _I, _P, _P]),
	'find_linear_wgrad_scratch_bytes': (c_int64, [_I]),
	'find_linear_wgrad': (c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P]),
	'find_latent_gather_fwd': (c_int, [_P, _I, _I, _P, _I, _P, _P]),
	'find_latent_gather_bwd': (c_int, [_P, _P, _I, _I, _I, _P, _P]),
	'find_latent_gather_many_fwd': (c_int, [_I, _P, _P, _P, _P, _I, _P, _P]),
	'find_latent_gather_many_bwd': (c_int, [_I, _P, _P, _P, _P, _I, _P, _P]),
	'find_weighted_terms_fwd': (c_int, [_I, _P, _P, _P, _P, _P]),
	'find_weighted_terms_bwd': (c_int, [_I, _P, _P, _P, _P, _P]),
	'find_uv_sample': (c_int, [_P, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P, _I, _I, _P, _P]),
	'find_render_frags': (c_int, [POINTER(RenderParams), _I, _I, _I, _I, _P, _P, _P, _P]),
	'find_adam_step': (c_int, [_I, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, _I, _P]),
	'find_adam_step_dev': (c_int, [_I, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, _P, _P]),
	'find_sgd_step': (c_int, [_I, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_int, c_int, _P]),
	'find_register_fwd': (c_int, [_P, _I, _P, _P, _I, _I, _P, _P]),
	'find_register_bwd_ws_bytes': (c_int64, [_I, _I]),
	'find_register_bwd': (c_int, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P]),
	'find_sample_points_fwd': (c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
	'find_sample_points_bwd': (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
	'find_face_areas': (c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
	'find_nn_fwd': (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
	'find_nn_bwd': (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
	'find_sample_surface_ws_bytes': (c_int64, [_I, _I]),
	'find_sample_surface_fwd': (c_int, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
	'find_sample_surface_again': (c_int, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
	'find_chamfer_ws_bytes': (c_int64, [_I, _I, _I]),
	'find_chamfer_fwd': (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
	'find_chamfer_bwd': (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
	'find_masked_mse_fwd': (c_int, [_P, _P, _I, _P, _P]),
	'find_masked_mse_bwd': (c_int, [_P, _P, _I, _P, _P, _P]),
	'find_image_mse_ws_bytes': (c_int64, []),
	'find_image_mse_fwd': (c_int, [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P]),
	'find_image_mse_bwd': (c_int, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
	'find_smooth_loss_fwd': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, c_float, c_float, _P, _P, _I, _P]),
	'find_smooth_loss_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, c_float, c_float, _P, _P, _I, _P, _P]),
	'find_smooth_ws_bytes': (c_int64, [_I, _I, _I]),
	'find_smooth_fwd': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
	'find_smooth_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
	'find_render_ws_bytes': (c_int64, [POINTER(RenderParams), _I, _I, _I, _I]),
	'find_render_fwd': (c_int, [POINTER(RenderParams), _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
	'find_render_bwd': (c_int, [POINTER(RenderParams), _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
	'find_render_flags': (c_int, [_P, _P, _P]),
}
# include/find_hip_diag.h: what libfind_hip_diag.so exports on top
DIAG_PROTOTYPES = {
	'find_debug_raster_ablate': (c_int, [_I]),
}


def lib():
	"""Load (once) and return the shared library; raise if it is missing or has the wrong ABI."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		raise RuntimeError(f'find_amd: HIP library not built: {LIB_PATH} is missing. Run `python -m find_amd.build` '
						   '(or __graft_entry__.build()). There is no CPU fallback.')
	# PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  It must be in the process BEFORE this
	# library is loaded so that both share that one runtime: loaded the other way round, the library binds /opt/rocm's copy, the
	# process ends up with device pointers of one runtime handed to the other and every launch fails ("no ROCm-capable device").
	import torch  # noqa: F401
	L = ctypes.CDLL(LIB_PATH)
	for name, (res, args) in list(PROTOTYPES.items()) + (list(DIAG_PROTOTYPES.items()) if DIAG else []):
		try:
			fn = getattr(L, name)
		except AttributeError as e:
			raise RuntimeError(f'find_amd: {LIB_PATH} does not export {name}; rebuild it') from e
		fn.restype = res
		fn.argtypes = args
	v = L.find_abi_version()
	if v != ABI_VERSION:
		raise RuntimeError(f'find_amd: ABI mismatch: library {v}, binding {ABI_VERSION}; rebuild with python -m find_amd.build')
	_lib = L
	# profiling aid: FIND_TUNING="key=value,key=value" becomes the default of every context created below (tools/, bench.py experiments)
	for kv in filter(None, os.environ.get('FIND_TUNING', '').split(',')):
		k, v = kv.split('=')
		if k.strip() == 'raster_ablate':   # process-wide, not a context knob: takes effect now
			_raster_switches(L, int(v))
		else:
			_TUNING_DEFAULTS[k.strip()] = int(v)
	return L


# ---------------------------------------------------------------------------------------------- per-device contexts
_ctx = {}               # device index -> find_ctx* (c_void_p); lives as long as the process
_TUNING_DEFAULTS = {}   # knob -> value applied to every context when it is created


def ctx(device=None):
	"""The find_ctx of a device (created on first use): internal streams / events of the MLP entry points, launch attributes,
	knobs.  `device`: torch.device, index or None (= the current device)."""
	import torch
	if device is None:
		idx = torch.cuda.current_device()
	elif isinstance(device, int):
		idx = device
	else:
		device = torch.device(device)
		idx = device.index if device.index is not None else torch.cuda.current_device()
	h = _ctx.get(idx)
	if h is None:
		L = lib()
		h = c_void_p()
		check(L.find_ctx_create(idx, ctypes.byref(h)), 'find_ctx_create')
		for k, v in _TUNING_DEFAULTS.items():
			if k == 'raster_ablate':
				_raster_switches(L, v)
			else:
				check(L.find_ctx_set(h, k.encode(), v), f'find_ctx_set({k})')
		_ctx[idx] = h
	return h


def set_tuning(key, value, device=None):
	"""Set a knob (find_hip.h: find_ctx_set) on the context of `device`, or -- device None -- on every existing context and as the
	default of contexts created later.  Returns nothing; raises on an unknown key."""
	L = lib()
	value = int(value)
	if key == 'raster_ablate':
		_raster_switches(L, value)
		return
	if device is not None:
		check(L.find_ctx_set(ctx(device), key.encode(), value), f'find_ctx_set({key})')
		return
	import torch
	if not _ctx and torch.cuda.is_available():
		ctx()
	for h in _ctx.values():
		check(L.find_ctx_set(h, key.encode(), value), f'find_ctx_set({key})')
	_TUNING_DEFAULTS[key] = value


def _raster_switches(L, bits):
	"""The process-wide switches of the rasteriser / the Chamfer search: the product takes the result-preserving bits (find_render_switches)
	and refuses the rest; the laboratory build takes every bit (find_debug_raster_ablate)."""
	if DIAG:
		check(L.find_debug_raster_ablate(bits), 'find_debug_raster_ablate')
	else:
		check(L.find_render_switches(bits), 'find_render_switches')


def get_tuning(key, device=None):
	v = c_int64()
	check(lib().find_ctx_get(ctx(device), key.encode(), ctypes.byref(v)), f'find_ctx_get({key})')
	return v.value


def check(rc, what):
	if rc != 0:
		msg = lib().find_last_error()
		raise RuntimeError(f'find_amd: {what} failed (code {rc}): {msg.decode() if msg else "?"}')


def ptr(t):
	"""Device pointer of a tensor (None -> NULL)."""
	return None if t is None else c_void_p(t.data_ptr())


def current_stream(device):
	"""hipStream_t of torch's current stream on `device`, as the C-ABI wants it.  (torch.cuda.current_stream() builds a Stream object every
	time -- 10 us, fifteen times per training step; the raw handle comes from the same C call without it.)"""
	import torch
	if isinstance(device, torch.device):
		idx = device.index if device.index is not None else torch.cuda.current_device()
	elif isinstance(device, int):
		idx = device
	else:
		idx = torch.device(device).index
		idx = torch.cuda.current_device() if idx is None else idx
	return c_void_p(torch._C._cuda_getCurrentRawStream(idx))
