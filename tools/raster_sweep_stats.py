"""Counters of the rasteriser's two sweeps (find_debug_raster_ablate bit 64) at the C3 / C4 render shapes: how much of a tile's list the second
sweep -- the band candidates of the pixels with more than K candidates -- walks again.  python tools/raster_sweep_stats.py [size]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from find_amd import _lib, functional_render as FR, synthetic
from find_amd._lib import check, current_stream, ptr
from find_amd.cameras import look_at_view_transform
from find_amd.functional import _faces_i32, _ws
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
v, f = synthetic.template(6890)
g = torch.Generator().manual_seed(0)
verts = (v[None] * (1 + 0.1 * torch.rand(16, 1, 3, generator=g))).cuda()
rng = np.random.RandomState(7)
R, T = look_at_view_transform(dist=np.full(4, 0.3), elev=rng.uniform(-90, 90, 4), azim=rng.uniform(-90, 90, 4), up=((1, 0, 0),))
R, T, fc = R.cuda(), T.cuda(), _faces_i32(f.cuda())
params = FR.make_params(size)
L = _lib.lib()
_lib.set_tuning('raster_ablate', 64)
ws = _ws(L.find_render_ws_bytes(ctypes.byref(params), 16, 4, 6890, fc.shape[0]), verts.device)
mask = torch.empty(16, 4, size, size, device='cuda')
check(L.find_render_fwd(ctypes.byref(params), ptr(verts), ptr(fc), 1, None, ptr(R), ptr(T), 16, 4, 6890, fc.shape[0], ptr(mask), None, None, None, ptr(ws), ws.numel(),
						current_stream(verts.device)), 'find_render_fwd')
torch.cuda.synchronize()
fl = ws[:256].view(torch.int32).cpu().tolist()
_lib.set_tuning('raster_ablate', 0)
print(f'{size}^2: first sweep {fl[24] * 64} lane-level tests = {fl[24]} (wave, face) evaluations, {fl[25] * 64} candidates; pixels in the second sweep {fl[4]}, largest band {fl[5]}, '
	  f'band candidates {fl[26]} ({fl[26] / max(fl[4], 1):.1f} per pixel); second sweep: {fl[28]} tiles (list entries {fl[31]}), {fl[29]} batches walked, {fl[30]} faces staged, '
	  f'{fl[27]} (wave, face) evaluations = {fl[27] / max(fl[24], 1):.2f} of the first sweep; unresolved {fl[1]}')
