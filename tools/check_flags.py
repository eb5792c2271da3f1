"""Count pixels with more than faces_per_pixel silhouette candidates at C3 / C4-share sizes (find_render_flags)."""
import os, sys, ctypes
os.environ.setdefault('FIND_DIAG', '1')   # laboratory build (include/find_hip_diag.h): this tool uses what the product library does not carry
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from find_amd import functional_render as FR, synthetic, _lib
from find_amd.functional import _ws, _c, _faces_i32
from find_amd._lib import ptr, check, current_stream
from find_amd.cameras import look_at_view_transform
_lib.set_tuning('raster_ablate', 64)
for size in (256, 512):
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(0)
	verts = (v[None] * (1 + 0.1 * torch.rand(16, 1, 3, generator=g))).cuda()
	rng = np.random.RandomState(7)
	R, T = look_at_view_transform(dist=np.full(4, 0.3), elev=rng.uniform(-90, 90, 4), azim=rng.uniform(-90, 90, 4), up=((1, 0, 0),))
	R, T = R.cuda(), T.cuda()
	params = FR.make_params(size)
	L = _lib.lib()
	faces = _faces_i32(f.cuda())
	N, V, M, F = 16, v.shape[0], 4, f.shape[0]
	ws = _ws(L.find_render_ws_bytes(ctypes.byref(params), N, M, V, F), verts.device)
	mask = torch.empty(N, M, size, size, device='cuda')
	check(L.find_render_fwd(ctypes.byref(params), ptr(verts), ptr(faces), 1, None, ptr(R), ptr(T), N, M, V, F, ptr(mask), None, None, None, ptr(ws), ws.numel(), current_stream(verts.device)), 'fwd')
	torch.cuda.synchronize()
	fl = ws[:256].view(torch.int32).cpu().tolist()
	print(size, 'flags (z-straddlers, unresolved overflow pixels):', FR.render_flags(ws), 'of', N * M * size * size, 'pixels; overflow pixels', fl[4], 'max candidates per pixel', fl[5])
	print('   waves in the K pass by longest list [64,128) [128,256) [256,512) [512,1024) >=1024:', fl[8:13], ' sum of longest lists', fl[20], ' over-lanes', fl[21],
		  ' K-pass wave time %.1f ms summed over waves (100 MHz ticks x16)' % (fl[22] * 16 / 100e6 * 1e3))
