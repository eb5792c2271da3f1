#!/usr/bin/env python3
"""Sub-path measurements of SURVEY.md §8(d) that bench.py's headline line does not carry: the rasteriser + silhouette /
Phong shaders (C3 and the per-rank share of C4), Chamfer at the training and eval sizes, the smoothness terms -- each timed
on the GPU (HIP events on the launch stream, inputs resident in HBM).  One JSON line per sub-path.
`python bench.py --subpaths` runs the same workloads and times the CPU oracle beside each of them (the oracle is only ever
touched from tests/, smoke() and bench.py's CPU-baseline leg); `python tools/bench_paths.py` prints the GPU side alone."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from find_amd import functional as FN            # noqa: E402
from find_amd import functional_render as FR     # noqa: E402
from find_amd import synthetic                   # noqa: E402
from find_amd.cameras import look_at_view_transform  # noqa: E402

HBM_PEAK_GBS = 8000.0


def gpu_ms(fn, warm_s=0.3, min_iters=10, min_s=0.1):
	"""Average GPU time of fn() in ms.  The GPU idles while the CPU oracle of the previous sub-path runs, so warm up by wall time
	(the clock needs a few hundred ms of work to come back up), then time enough iterations to cover min_s."""
	t0 = time.perf_counter()
	n = 0
	while n < 3 or time.perf_counter() - t0 < warm_s:
		fn()
		n += 1
		if n % 4 == 0:
			torch.cuda.synchronize()
	torch.cuda.synchronize()
	per = max((time.perf_counter() - t0) / n, 1e-5)
	iters = max(min_iters, int(min_s / per))
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters):
		fn()
	e1.record()
	e1.synchronize()
	return e0.elapsed_time(e1) / iters


def views(m, seed=7):
	rng = np.random.RandomState(seed)
	return look_at_view_transform(dist=np.full(m, 0.3), elev=rng.uniform(-90, 90, m), azim=rng.uniform(-90, 90, m), up=((1, 0, 0),))


# HBM-side bytes per forward launch of raster_kernel / per launch of sil_bwd_kernel at the C3 shape, rocprofv3 --pmc FETCH_SIZE (x2, the
# gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE in separate passes (tools/prof_raster.sh -> profiles/r05_raster_pmc.txt)
RASTER_TRAFFIC_C3 = {'raster_kernel': 2 * 649308.6 * 1024 + 616660.7 * 1024, 'sil_bwd_kernel': 2 * 118954.5 * 1024 + 83899.9 * 1024}   # profiles/r05_raster_pmc.txt


def raster_counts(verts, fc, Rc, Tc, params):
	"""(pixel x face tests issued, silhouette candidates) of one forward render: the rasteriser's diagnostic counters (ablation bit 64, which
	only the laboratory build has: run with FIND_DIAG=1; the product library yields no counts)."""
	import ctypes
	from find_amd import _lib
	if not _lib.DIAG:
		raster_counts.last_flags = [None] * 64
		return None, None, None
	from find_amd._lib import check, current_stream, ptr
	from find_amd.functional import _faces_i32, _ws
	L = _lib.lib()
	N, V = verts.shape[0], verts.shape[1]
	M, F = Rc.shape[0], fc.shape[0]
	faces = _faces_i32(fc)
	keep = int(os.environ.get('RASTER_ABLATE_KEEP', '0'))   # bits to keep while counting (128 = the wave-per-tile kernel)
	_lib.set_tuning('raster_ablate', 64 | keep)
	try:
		ws = _ws(L.find_render_ws_bytes(ctypes.byref(params), N, M, V, F), verts.device)
		mask = torch.empty(N, M, params.image_h, params.image_w, device=verts.device)
		check(L.find_render_fwd(ctypes.byref(params), ptr(verts), ptr(faces), 1, None, ptr(Rc), ptr(Tc), N, M, V, F, ptr(mask), None, None, None, ptr(ws),
								ws.numel(), current_stream(verts.device)), 'find_render_fwd')
		torch.cuda.synchronize()
		fl = ws[:256].view(torch.int32).cpu().tolist()
	finally:
		_lib.set_tuning('raster_ablate', keep)
	raster_counts.last_flags = fl
	return fl[24] * 64, fl[25] * 64, fl[4]


def bench_render(n_feet, n_views, size, want_image, cpu=None):
	v, f = synthetic.template(6890)
	g = torch.Generator().manual_seed(0)
	verts = (v[None] * (1 + 0.1 * torch.rand(n_feet, 1, 3, generator=g))).cuda()
	cols = torch.rand(n_feet, v.shape[0], 3, generator=g).cuda()
	R, T = views(n_views)
	Rc, Tc, fc = R.cuda(), T.cuda(), f.cuda()
	params = FR.make_params(size)
	gt = torch.rand(n_feet, n_views, size, size, generator=g).cuda()
	gti = torch.rand(n_feet, n_views, size, size, 3, generator=g).cuda() if want_image else None

	def fwd():
		return FR.render(verts, cols if want_image else None, fc, Rc, Tc, params, want_image=want_image)

	def fwdbwd():
		vg = verts.detach().requires_grad_(True)
		cg = cols.detach().requires_grad_(True) if want_image else None
		mask, image, _, _ = FR.render(vg, cg, fc, Rc, Tc, params, want_image=want_image)
		loss = ((mask - gt) ** 2).mean()
		if want_image:
			loss = loss + ((image - gti) ** 2).mean()
		loss.backward()

	ms_f, ms_fb = gpu_ms(fwd), gpu_ms(fwdbwd)
	tests, cands, over_px = raster_counts(verts, fc, Rc, Tc, params)
	px = n_feet * n_views * size * size
	images = n_feet * n_views
	F = f.shape[0]
	alg = px * (8 + (24 if want_image else 0)) + images * 36 * F
	out = dict(path=f'render+{"phong+" if want_image else ""}silhouette fwd+bwd', workload=f'{n_feet} feet x {n_views} views @{size}^2, V=6890 F={F}',
			   ms_fwd=ms_f, ms_fwd_bwd=ms_fb, vertices_views_per_s=n_feet * 6890 * n_views / (ms_fb * 1e-3), mpix_per_s_fwd=px / ms_f / 1e3,
			   bytes_algorithmic=alg, achieved_GBs_fwd=alg / (ms_f * 1e-3) / 1e9, hbm_frac_fwd=alg / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
			   pixel_face_tests=tests, tests_per_s_fwd=None if tests is None else tests / (ms_f * 1e-3), silhouette_candidates=cands, pixels_over_K=over_px,
			   candidate_list_bytes=None if cands is None else 8 * cands, lane_efficiency=None if cands is None else cands / max(tests, 1),
			   tiles_left_early=raster_counts.last_flags[26], tie_fixup_pixels=raster_counts.last_flags[7], pool_entries=raster_counts.last_flags[6],
			   hbm_traffic_bytes_fwd_launch=RASTER_TRAFFIC_C3['raster_kernel'] if (size == 256 and n_feet == 16 and n_views == 4 and not want_image) else None,
			   bound='VALU (pixel x face fragment math; lists in depth order let a wave leave when its pixels hold their K nearest), then HBM traffic of the '
					 'per-pixel candidate lists (8 B per candidate, written once, read ~3x by the K-nearest pass); HBM floor of the fused output %.1f us' % (alg / (HBM_PEAK_GBS * 1e9) * 1e6))
	if cpu:
		dt = cpu('render', verts=verts[:1].cpu().numpy(), faces=f.numpy(), colors=cols[:1].cpu().numpy(), R=R[:1].numpy(), T=T[:1].numpy(), size=size,
				 want_image=want_image)
		out['cpu_oracle'] = dict(ms_per_image_fwd=dt * 1e3, sample='1 foot x 1 view forward, oracle/raster_ref.c (OpenMP, all host threads)',
								 gpu_speedup_fwd=(dt * 1e3) / (ms_f / images))
	return out


def bench_chamfer(n_feet, p1, p2, label, cpu=None):
	g = torch.Generator().manual_seed(1)
	x = (torch.rand(n_feet, p1, 3, generator=g) * 0.2).cuda()
	y = (torch.rand(n_feet, p2, 3, generator=g) * 0.2).cuda()

	def fwdbwd():
		xg = x.detach().requires_grad_(True)
		loss, _ = FN.chamfer_distance(xg, y)
		loss.backward()

	ms = gpu_ms(fwdbwd)
	pairs = n_feet * p1 * p2
	out = dict(path='chamfer fwd+bwd', workload=f'{label}: {n_feet} feet, {p1} x {p2} points', ms=ms, gpairs_per_s=pairs / (ms * 1e-3) / 1e9,
			   gflops=8.0 * pairs / (ms * 1e-3) / 1e9, bytes_algorithmic=24 * (p1 + p2) * n_feet, bound='VALU (3-D distances); not MFMA-shaped')
	if cpu:
		dt = cpu('chamfer', x=x[:1].cpu(), y=y[:1].cpu())
		out['cpu_oracle'] = dict(ms_per_foot_fwd=dt * 1e3, sample='1 foot forward, oracle/geom_ref.py (torch-CPU cdist)', gpu_speedup=(dt * 1e3) / (ms / n_feet))
	return out


def bench_smooth(n_feet, cpu=None):
	v, f = synthetic.template(6890)
	topo = FN.MeshTopology(f.cuda(), v.shape[0])
	g = torch.Generator().manual_seed(2)
	verts = (v[None] + 0.001 * torch.randn(n_feet, v.shape[0], 3, generator=g)).cuda()

	def fwdbwd():
		vg = verts.detach().requires_grad_(True)
		e, l = FN.mesh_edge_and_laplacian(vg, topo)
		(e + l).backward()

	ms = gpu_ms(fwdbwd)
	out = dict(path='mesh_edge_loss + cot-Laplacian smoothing fwd+bwd', workload=f'{n_feet} feet, V=6890 F={f.shape[0]}', ms=ms,
			   vertices_per_s=n_feet * 6890 / (ms * 1e-3), bound='gather latency (CSR tables), HBM floor trivial')
	if cpu:
		dt = cpu('smooth', verts=verts[:1].cpu(), faces=f)
		out['cpu_oracle'] = dict(ms_per_foot_fwd=dt * 1e3, sample='1 foot forward, oracle/geom_ref.py', gpu_speedup=(dt * 1e3) / (ms / n_feet))
	return out


def run_all(cpu=None):
	"""cpu: None, or a callable (kind, **inputs) -> seconds that times the CPU oracle on that sample (supplied by bench.py)."""
	return [bench_render(16, 4, 256, False, cpu), bench_render(16, 4, 256, True, cpu), bench_render(16, 4, 512, True, cpu),
			bench_chamfer(16, 5000, 5000, 'train (losses.py:61)', cpu), bench_chamfer(16, 10000, 10000, 'eval (eval_3d.py:148)', cpu),
			bench_smooth(16, cpu)]


def main():
	if not torch.cuda.is_available():
		raise SystemExit('bench_paths.py needs an MI355X')
	for r in run_all(None):
		print(json.dumps(r), flush=True)


if __name__ == '__main__':
	main()
