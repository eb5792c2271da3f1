#!/usr/bin/env python3
"""Headline benchmark of the FIND hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric (BASELINE.json): deformed vertices x rendered views / second, forward+backward.
Workload at every N: BASELINE.json configs[1] per GPU -- a batch of 16 feet x 6890-vertex template through
Fourier PE + trunk + displacement/colour heads + similarity registration, loss = sum(verts^2)+sum(col^2), full
backward to every weight, latent row and registration row (views := 1: nothing is rendered in this config).
Weak scaling: every rank owns 16 distinct feet; gradients of the replicated parameters are averaged with one
RCCL all-reduce per step.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- the dominant kernel (256x256 Linear+ReLU fp32-MFMA GEMM over all 110 240 rows) timed live with
                  HIP events on the launch stream; achieved = 2*rows*256*256 flop / average duration.
  cpu_baseline -- oracle/mlp_ref.py (the reference's op sequence on torch-CPU) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch


def dtype_label():
	"""Arithmetic type of the path as configured: fp32 unless the opt-in fp16 matrix-pipe mode was switched on."""
	from find_amd import functional as FF
	if FF.get_mlp_precision() == 'fp16':
		return 'f16 operands / f32 accumulation in the 256->256 layers (fwd, dX, dW), f32 tensors and f32 elsewhere'
	return 'f32'


def emit(obj):
	"""Print the result as the LAST line of stdout: RCCL writes its version banner through C stdio, which is block-buffered on a pipe
	and would otherwise come out after this line when the process group is destroyed."""
	import ctypes
	try:
		ctypes.CDLL(None).fflush(None)
	except Exception:
		pass
	print(json.dumps(obj), flush=True)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FEET = 16
N_VERTS = 6890
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable by a streaming kernel)
# MAC counts per vertex evaluation: reference-equivalent fwd+bwd, and what this build executes when the
# trunk is shared by the 16 feet of a batch (trunk fwd+bwd once per template vertex instead of once per foot-vertex).
MAC_FWD_REF = 866304
MAC_FWDBWD_REF = 2465536
MAC_TRUNK_FWD = 515 * 256 + 4 * 256 * 256
MAC_HEADS_FWD = (456 + 356) * 256 + 4 * 256 * 256 + 2 * 3 * 256


# HBM-side traffic of one launch of the dominant kernel, measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes (tools/pmc_traffic.sh; profiles/r01_traffic_pmc_summary.txt): 2 x 62.02 MB fetched + 112.9 MB written (fp32 gemm4),
# 2 x 57.10 MB + 112.9 MB (fp16-mode gemm5).
GEMM_TRAFFIC_BYTES = 236.9e6
GEMM5_TRAFFIC_BYTES = 227.1e6


def executed_flops_per_step(n_feet, n_verts):
	"""fwd+bwd flops this build executes for one batch with a shared template (per-foot latent columns folded into a bias)."""
	trunk_fwd = MAC_TRUNK_FWD
	trunk_bwd = 2 * MAC_TRUNK_FWD - 515 * 256  # dW + dX, no dX through layer 0
	heads_in = 2 * 256 * 256                      # first layer of both heads: the latent columns are a bias, and the trunk rows are shared,
	heads_rest = 4 * 256 * 256 + 2 * 3 * 256      # so H W^T (forward), dW and dH (backward, after the sum over feet) run once per TEMPLATE vertex
	return 2.0 * (n_verts * (trunk_fwd + trunk_bwd + 3 * heads_in) + n_feet * n_verts * 3 * heads_rest)


def build_step(device, seed):
	from find_amd import synthetic
	model = synthetic.make_model(N_VERTS, train_size=N_FEET, val_size=2, device=device)
	lat = synthetic.latents(N_FEET, seed=seed, device=device)
	# latent rows live in the model's tables (LatentVector); the step gathers them like trainer.sample_latent_vectors
	with torch.no_grad():
		model.shapevec.data.copy_(lat['shapevec'])
		model.texvec.data.copy_(lat['texvec'])
		model.posevec.data.copy_(lat['posevec'])
		model.reg.data.copy_(lat['reg'])
	idx = torch.arange(N_FEET, device=device)
	params = [p for p in model.parameters() if p.requires_grad]

	def step():
		for p in params:
			p.grad = None
		batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx],
					 reg_train=model.reg[idx])
		res = model.get_meshes_from_batch(batch, is_train=True)
		loss = (res['verts'] ** 2).sum() + (res['col'] ** 2).sum()
		loss.backward()
		return loss

	return model, params, step


def time_dominant_kernel(device, iters=100, warm=150):
	"""Average duration of the dominant kernel (Linear 256->256 + ReLU over all head rows), HIP events on the launch stream."""
	import ctypes
	from find_amd import _lib
	L = _lib.lib()
	rows = N_FEET * N_VERTS
	g = torch.Generator().manual_seed(0)
	x = torch.randn(rows, 256, generator=g).to(device)
	w = (torch.randn(256, 256, generator=g) / 16).to(device)
	b = torch.randn(256, generator=g).to(device)
	y = torch.empty_like(x)
	stream = torch.cuda.current_stream(device)

	def launch():
		_lib.check(L.find_linear_relu_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), N_FEET, N_VERTS, _lib.ptr(y),
										  ctypes.c_void_p(stream.cuda_stream)), 'find_linear_relu_fwd')

	# the GPU idled while the inputs were generated on the host: launch long enough for the clock to come back up before timing
	for _ in range(warm):
		launch()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record(stream)
	for _ in range(iters):
		launch()
	e1.record(stream)
	e1.synchronize()
	ms = e0.elapsed_time(e1) / iters
	flops = 2.0 * rows * 256 * 256
	return ms, flops


def cpu_baseline(sample_feet=2, steps=2):
	"""oracle/mlp_ref.py = the reference's op sequence (no trunk sharing, latents concatenated per vertex) on host cores."""
	from find_amd import synthetic
	from oracle import mlp_ref
	try:
		avail = len(os.sched_getaffinity(0))
	except AttributeError:
		avail = os.cpu_count() or 1
	model = synthetic.make_model(N_VERTS, train_size=N_FEET, val_size=2, device='cpu')
	sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in model.state_dict().items()}
	B = model.encoder[0]._B
	tv = model.template_verts.data

	def one_step(n_feet):
		lat = synthetic.latents(n_feet, seed=0, device='cpu')
		lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
		t0 = time.perf_counter()
		res = mlp_ref.get_meshes_verts(sd, B, tv, lv['shapevec'], lv['reg'], lv['texvec'], lv['posevec'])
		loss = (res['verts'] ** 2).sum() + (res['col'] ** 2).sum()
		loss.backward()
		return time.perf_counter() - t0

	# torch-CPU does not scale to every hardware thread of a big host on GEMMs this small: pick the best thread count
	# from a short probe (1 foot), then time the sample with it
	probe = {}
	for t in sorted({min(avail, c) for c in (8, 16, 32, 64, 128)}):
		torch.set_num_threads(t)
		one_step(1)
		probe[t] = one_step(1)
	cores = min(probe, key=probe.get)
	torch.set_num_threads(cores)
	best = min(one_step(sample_feet) for _ in range(steps))
	return dict(value=sample_feet * N_VERTS / best, unit='vertices*views/s', cores=cores, kind='port',
				sample=f'{sample_feet} of {N_FEET} feet x {N_VERTS} verts, fwd+bwd, best of {steps}, torch-CPU with {cores} threads '
					   f'(best of a {sorted(probe)}-thread probe; host exposes {avail} hardware threads)')


def train3d(with_cpu, steps, warmup, n_feet=16):
	"""The reference's own training configuration (cfgs/train_3d.yaml: losses chamf + smooth + texture, nothing rendered in the
	timed loop, so views := 1): ModelWithLoss.forward on a batch of `n_feet` feet (6890-vertex template, 10 002-vertex GT scans,
	5000 / 1000 surface samples as losses.py:27,61), backward, and the step of the three optimisers (train.py:161-168).  One
	JSON line; the CPU leg runs the oracle's composition of the same step on a bounded sample of feet."""
	from find_amd import optim, synthetic
	from find_amd.model_with_loss import ModelWithLoss
	from find_amd.opts import Opts
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	if not torch.cuda.is_available():
		raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
	dev = torch.device('cuda', 0)
	opts = Opts(chamf_loss=True, smooth_loss=True, texture_loss=True)
	mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=n_feet, val_size=2,
						shapevec_size=100, texvec_size=100, posevec_size=100, template_mesh_loc=None)
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():
		mwl.model.mlp_disp[-1].weight.copy_(torch.randn(mwl.model.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		mwl.model.mlp_disp[-1].bias.copy_(torch.randn(3, generator=g) * 0.01)
	mwl = mwl.to(dev)
	v, f = synthetic.template(N_VERTS)
	mwl.model.set_template(v.to(dev), f.to(dev))
	lat = synthetic.latents(n_feet, seed=0, device=dev)
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			getattr(mwl.model, k).data.copy_(lat[k])
	gv, gf, gc = synthetic.gt_feet(n_feet, 10002, seed=0, device=dev)
	batch = dict(mesh=Meshes(gv, gf, TexturesVertex(gc.clamp(0.05, 0.95))), idx=torch.arange(n_feet, device=dev), name=[f'{i:04d}' for i in range(n_feet)])
	m = mwl.model
	optims = [optim.Adam(m.main_params, lr=5e-4), optim.SGD(m.reg_params, lr=1e-3, momentum=0.9), optim.Adam(m.latent_params, lr=1e-3)]

	def step():
		for o in optims:
			o.zero_grad(set_to_none=True)
		b = dict(batch)
		b.update(sample_latent_vectors(b, m.latent_vectors_train))
		loss, _ = mwl(b, 0, opts, chamf=True, smooth=True, texture=True)
		loss.backward()
		for o in optims:
			o.step()
		return loss

	for _ in range(warmup):
		step()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(steps):
		step()
	torch.cuda.synchronize()
	ms = (time.perf_counter() - t0) / steps * 1e3
	out = {'metric': 'deformed vertices x rendered views / sec (fwd+bwd)', 'value': n_feet * N_VERTS / (ms * 1e-3), 'unit': 'vertices*views/s', 'n_gpus': 1,
		   'steps': steps, 'warmup': warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_label(),
		   'data': 'synthetic',
		   'config': {'workload': f'train_3d.yaml: {n_feet} feet x {N_VERTS}-vertex template, GT 10002-vertex scans, losses chamf(5000 samples) + smooth + '
								  f'texture(1000 samples), backward, Adam/SGD/Adam steps; nothing rendered, views:=1', 'feet_per_gpu': n_feet}}
	if with_cpu:
		out['cpu_baseline'] = train3d_cpu(mwl, batch, gv, gf, gc)
	emit(out)


def train3d_cpu(mwl, batch, gv, gf, gc, sample_feet=1):
	"""Oracle composition of the same step (tests/test_gpu_pipeline.py::test_train_3d_loss_set_matches_oracle) on `sample_feet` feet."""
	from oracle import geom_ref, mlp_ref
	m = mwl.model
	nf = sample_feet
	sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	B, tv, tf = m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()
	lat = {k: getattr(m, k).data.detach().cpu()[:nf].clone().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec', 'reg')}
	gvc, gfc, gcc = gv.cpu()[:nf], gf.cpu(), gc.cpu()[:nf].clamp(0.05, 0.95)
	g = torch.Generator().manual_seed(0)

	def draws(verts, faces, n):
		areas = geom_ref.face_areas(verts, faces)
		return torch.multinomial(areas, n, replacement=True, generator=g), torch.rand(verts.shape[0], n, 2, generator=g)

	def one():
		t0 = time.perf_counter()
		res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
		fi, uv = draws(gvc, gfc, 5000)
		gt_s = geom_ref.sample_points(gvc, gfc, fi, uv)
		fi, uv = draws(res['verts'].detach(), tf, 5000)
		pr_s = geom_ref.sample_points(res['verts'], tf, fi, uv)
		l_ch = geom_ref.chamfer_distance(pr_s, gt_s)
		l_sm = geom_ref.mesh_smoothness(res['verts'], tf)
		fi, uv = draws(gvc, gfc, 1000)
		tx_p, tx_c = geom_ref.sample_points(gvc, gfc, fi, uv, attr=gcc)
		col = mlp_ref.mlp_forward(sd, B, tx_p, lat['shapevec'], lat['texvec'], lat['posevec'])['col']
		mask = (tx_c < 1).any(dim=-1, keepdim=True).expand(-1, -1, 3)
		l_tx = (torch.nn.functional.mse_loss(col, tx_c, reduction='none') * mask).mean()
		(l_ch * 10000. + l_sm * 1000. + l_tx).backward()
		return time.perf_counter() - t0

	try:
		avail = len(os.sched_getaffinity(0))
	except AttributeError:
		avail = os.cpu_count() or 1
	cores = min(avail, 16)
	torch.set_num_threads(cores)
	one()
	best = min(one() for _ in range(2))
	return dict(value=nf * N_VERTS / best, unit='vertices*views/s', cores=cores, kind='port',
				sample=f'{nf} of the feet, same step without the optimiser update, best of 2, oracle (torch-CPU / numpy) with {cores} threads')


def c3(with_cpu, steps, warmup, n_feet=16, n_views=4, size=256, c4=False):
	"""BASELINE.json configs[2] (and, with c4=True, the per-rank share of configs[3]: 16 of the 128 feet, 4 views @512^2, silhouette + pixel +
	Chamfer losses; under torch.distributed.run every rank takes 16 feet and the gradients are all-reduced, as in the headline run).
	configs[2]: a batch of 16 feet x 4 views @256^2 with the silhouette render loss, end to end -- MLP query,
	registration, GT and predicted renders (the GT is re-rendered every step, as the reference does), silhouette loss, backward
	through rasteriser and MLP, optimiser steps.  One JSON line; the CPU leg runs the oracle's composition on one foot x one view."""
	import numpy as np
	from find_amd import optim, synthetic
	from find_amd.model_with_loss import ModelWithLoss
	from find_amd.opts import Opts
	from find_amd.renderer import FootRenderer
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import sample_latent_vectors
	if not torch.cuda.is_available():
		raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
	import torch.distributed as dist
	from find_amd import distributed as fdist
	rank, world, local = fdist.init_from_env()
	torch.cuda.set_device(local)
	dev = torch.device('cuda', local)
	if c4:
		size = 512
	opts = Opts(sil_loss=True, pix_loss=c4, chamf_loss=c4, num_views=n_views)
	mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=n_feet, val_size=2,
						shapevec_size=100, texvec_size=100, posevec_size=100, template_mesh_loc=None)
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():
		mwl.model.mlp_disp[-1].weight.copy_(torch.randn(mwl.model.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		mwl.model.mlp_disp[-1].bias.copy_(torch.randn(3, generator=g) * 0.01)
	mwl = mwl.to(dev)
	mwl.rdr = FootRenderer(image_size=size, device=dev)
	v, f = synthetic.template(N_VERTS)
	mwl.model.set_template(v.to(dev), f.to(dev))
	lat = synthetic.latents(n_feet, seed=rank, device=dev)
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			getattr(mwl.model, k).data.copy_(lat[k])
	gv, gf, gc = synthetic.gt_feet(n_feet, 10002, seed=rank, device=dev)
	batch = dict(mesh=Meshes(gv, gf, TexturesVertex(gc.clamp(0.05, 0.95))), idx=torch.arange(n_feet, device=dev), name=[f'{i:04d}' for i in range(n_feet)])
	np.random.seed(7)
	R, T = mwl.rdr.sample_views(nviews=n_views, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)
	m = mwl.model
	optims = [optim.Adam(m.main_params, lr=5e-4), optim.SGD(m.reg_params, lr=1e-3, momentum=0.9), optim.Adam(m.latent_params, lr=1e-3)]
	bucket = None
	if world > 1:
		trainable = [p for p in m.parameters() if p.requires_grad]
		fdist.broadcast_parameters([p for p in m.parameters() if p.is_floating_point()])
		bucket = fdist.GradBucket(trainable)

	def step():
		for o in optims:
			o.zero_grad(set_to_none=True)
		b = dict(batch)
		b.update(sample_latent_vectors(b, m.latent_vectors_train))
		loss, _ = mwl(b, 0, opts, sil=True, pix=c4, chamf=c4, render_foot=True, views=(R, T))
		loss.backward()
		if bucket is not None:
			bucket.allreduce_()
		for o in optims:
			o.step()
		return loss

	for _ in range(warmup):
		step()
	if world > 1:
		dist.barrier()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(steps):
		step()
	torch.cuda.synchronize()
	if world > 1:
		dist.barrier()
	torch.cuda.synchronize()
	elapsed = time.perf_counter() - t0
	if world > 1:
		t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t.item())
	ms = elapsed / steps * 1e3
	if rank != 0:
		dist.barrier()
		dist.destroy_process_group()
		return
	n_feet_total = n_feet * world
	out = {'metric': 'deformed vertices x rendered views / sec (fwd+bwd)', 'value': n_feet_total * N_VERTS * n_views / (ms * 1e-3), 'unit': 'vertices*views/s',
		   'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
		   'dtype': dtype_label(), 'data': 'synthetic',
		   'config': {'workload': f'{"C4 rank share" if c4 else "C3"}: {n_feet} feet x {n_views} views @{size}^2 per GPU, {N_VERTS}-vertex template (13776 faces), 10002-vertex GT '
								  f'scans re-rendered every step, {"silhouette + pixel + Chamfer losses" if c4 else "silhouette loss"}, backward through rasteriser + MLP, '
								  f'optimiser steps', 'feet_per_gpu': n_feet, 'views': n_views, 'parallelism': f'dp{world}'}}
	if with_cpu and world == 1 and not c4:
		out['cpu_baseline'] = c3_cpu(mwl, gv, gf, R, T, size)
	emit(out)
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()


def c3_cpu(mwl, gv, gf, R, T, size):
	"""Oracle composition of the C3 step on one foot x one view: MLP (torch-CPU), C rasteriser for both meshes, torch restatement of the
	soft silhouette for the gradient, backward to the MLP weights."""
	from oracle import mlp_ref, render_ref
	m = mwl.model
	sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and k.split('.')[0] in ('base', 'mlp_disp', 'mlp_col'))
		  for k, v in m.state_dict().items()}
	B, tv, tf = m.encoder[0]._B, m.template_verts.data.cpu(), m.template_faces.data[0].cpu().long()
	lat = {k: getattr(m, k).data.detach().cpu()[:1].clone().requires_grad_(True) for k in ('shapevec', 'texvec', 'posevec', 'reg')}
	Rc, Tc = R[:1].cpu(), T[:1].cpu()
	rp = render_ref.default_params(size)
	try:
		avail = len(os.sched_getaffinity(0))
	except AttributeError:
		avail = os.cpu_count() or 1
	cores = min(avail, 16)
	torch.set_num_threads(cores)

	def one():
		t0 = time.perf_counter()
		res = mlp_ref.get_meshes_verts(sd, B, tv, lat['shapevec'], lat['reg'], lat['texvec'], lat['posevec'])
		gt = render_ref.render(gv[:1].cpu().numpy(), gf.cpu().numpy(), None, Rc.numpy(), Tc.numpy(), image_size=size, want_image=False)['mask']
		vproj = render_ref.project(rp, res['verts'].detach().numpy(), Rc.numpy(), Tc.numpy())
		p2f, _, _, _ = render_ref.rasterize(vproj, tf.numpy(), 1, size, size, 100, rp.sil_blur_radius)
		mask = render_ref.torch_mask(rp, res['verts'], tf, Rc, Tc, torch.from_numpy(p2f).long(), 1)
		((mask - torch.from_numpy(gt)) ** 2).mean().backward()
		return time.perf_counter() - t0

	best = min(one() for _ in range(2))
	return dict(value=N_VERTS / best, unit='vertices*views/s', cores=cores, kind='port',
				sample=f'1 foot x 1 view of the same step without the optimiser update, best of 2, oracle (torch-CPU MLP + oracle/raster_ref.c OpenMP + '
					   f'torch autograd through the K=100 fragments) with {cores} torch threads')


def subpaths(with_cpu):
	"""Sub-path lines (tools/bench_paths.py workloads).  The CPU leg times the oracle on one foot / one image of the same inputs."""
	sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
	import bench_paths

	def cpu(kind, **kw):
		from oracle import geom_ref, render_ref
		t0 = time.perf_counter()
		if kind == 'render':
			render_ref.render(kw['verts'], kw['faces'], kw['colors'], kw['R'], kw['T'], image_size=kw['size'], want_image=kw['want_image'])
		elif kind == 'chamfer':
			geom_ref.chamfer_distance(kw['x'], kw['y'])
		else:
			geom_ref.mesh_smoothness(kw['verts'], kw['faces'])
		return time.perf_counter() - t0

	if not torch.cuda.is_available():
		raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
	for r in bench_paths.run_all(cpu if with_cpu else None):
		emit(r)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type=int, default=1)
	ap.add_argument('--steps', type=int, default=30)
	ap.add_argument('--warmup', type=int, default=5)
	ap.add_argument('--no-cpu-baseline', action='store_true')
	ap.add_argument('--train3d', action='store_true', help='instead of the headline line: the reference training configuration (train_3d.yaml losses + optimiser steps)')
	ap.add_argument('--c3', action='store_true', help='instead of the headline line: BASELINE configs[2] end to end (16 feet x 4 views @256^2, silhouette render loss)')
	ap.add_argument('--c5', action='store_true', help='BASELINE configs[4] geometry: the headline workload on the 50 002-vertex dense template (fp32 unless --fp16)')
	ap.add_argument('--fp16', action='store_true', help="opt-in reduced precision (find_amd.functional.set_mlp_precision('fp16')): the 256->256 layers, forward and dX, on the fp16 matrix pipe with fp32 accumulation -- BASELINE configs[4] with --c5; NOT the parity path, never the default line")
	ap.add_argument('--c4', action='store_true', help='per-rank share of BASELINE configs[3]: 16 feet x 4 views @512^2, silhouette + pixel + Chamfer losses; works under torch.distributed.run')
	ap.add_argument('--dp-overhead', action='store_true', help='diagnostic: run the headline step on ONE GPU through the data-parallel code path (one-rank RCCL group, gradient bucket + all-reduce) to see what the N>1 bookkeeping costs per step')
	ap.add_argument('--subpaths', action='store_true', help='instead of the headline line: one JSON line per render / Chamfer / smoothness sub-path (SURVEY 8d), CPU oracle timed beside each')
	args = ap.parse_args()
	if args.c5:
		global N_VERTS
		N_VERTS = 50002
	if args.fp16:
		from find_amd import functional as FF
		FF.set_mlp_precision('fp16')
	if args.subpaths:
		return subpaths(not args.no_cpu_baseline)
	if args.train3d:
		return train3d(not args.no_cpu_baseline, args.steps, args.warmup)
	if args.c3 or args.c4:
		return c3(not args.no_cpu_baseline, args.steps, args.warmup, c4=args.c4)

	import torch.distributed as dist
	from find_amd import distributed as fdist
	rank, world, local = fdist.init_from_env()
	if world != args.gpus:
		raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}')
	if not torch.cuda.is_available():
		raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
	torch.cuda.set_device(local)
	device = torch.device('cuda', local)

	model, params, step = build_step(device, seed=rank)
	bucket = None
	if args.dp_overhead and world == 1:
		dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29533', rank=0, world_size=1)
		bucket = fdist.GradBucket(params)
	if world > 1:
		fdist.broadcast_parameters([p for p in model.parameters() if p.is_floating_point()])
		bucket = fdist.GradBucket(params)

	def full_step():
		step()
		if bucket is not None:
			bucket.allreduce_()

	for _ in range(args.warmup):
		full_step()
	if world > 1:
		dist.barrier()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(args.steps):
		full_step()
	torch.cuda.synchronize()
	if world > 1:
		dist.barrier()
	torch.cuda.synchronize()
	elapsed = time.perf_counter() - t0
	if world > 1:
		t = torch.tensor([elapsed], device=device, dtype=torch.float64)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		elapsed = float(t.item())

	if world > 1:
		# every rank pushes out what C stdio has buffered (RCCL's banner) before rank 0 prints the result line
		import ctypes
		ctypes.CDLL(None).fflush(None)
		sys.stdout.flush()
		dist.barrier()
	if rank == 0:
		ms_step = elapsed / args.steps * 1e3
		verts_per_step = world * N_FEET * N_VERTS
		value = verts_per_step * args.steps / elapsed
		kms, kflops = time_dominant_kernel(device)
		ach = kflops / (kms * 1e-3) / 1e12
		fl_exec = executed_flops_per_step(N_FEET, N_VERTS)
		fl_ref = 2.0 * MAC_FWDBWD_REF * N_FEET * N_VERTS
		out = {
			'metric': 'deformed vertices x rendered views / sec (fwd+bwd)', 'value': value, 'unit': 'vertices*views/s',
			'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
			'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_label(), 'data': 'synthetic',
			'config': {'workload': f'{("C5 geometry, " + ("fp16 MLP" if args.fp16 else "fp32")) if N_VERTS == 50002 else ("C2, fp16 MLP (opt-in mode)" if args.fp16 else "C2")}: {N_FEET} feet x {N_VERTS}-vertex template per GPU, PE+trunk+heads+registration fwd+bwd, views:=1',
					   'feet_per_gpu': N_FEET, 'template_verts': N_VERTS, 'parallelism': f'dp{world}' + (' through the one-rank bucket + RCCL path (diagnostic)' if bucket is not None and world == 1 else ''),
					   'flops_executed_per_step': fl_exec, 'flops_reference_equiv_per_step': fl_ref,
					   'step_tflops_executed': fl_exec / (ms_step * 1e-3) / 1e12,
					   'step_tflops_reference_equiv': fl_ref / (ms_step * 1e-3) / 1e12},
			'roofline': {'bound': 'mfma', 'kernel': 'find::mlp::gemm4_kernel<1, 4, 8> (Linear 256->256 + bias + ReLU over 110240 rows, fp32 MFMA)',
						 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_MFMA_TFLOPS,
						 'avg_kernel_ms': kms, 'flops_per_launch': kflops, 'traffic': GEMM_TRAFFIC_BYTES if N_VERTS == 6890 else None,
						 'traffic_note': 'HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), '
										 'profiles/r01_traffic_pmc_summary.txt; algorithmic 226.0e6'},
		}
		if args.fp16:
			# gemm5 is bound by its streams: algorithmic bytes = rows x 1 KB read + rows x 1 KB written + the 256-KB weight matrix
			nbytes = 2.0 * N_FEET * N_VERTS * 1024 + 256 * 1024
			gbs = nbytes / (kms * 1e-3) / 1e9
			out['roofline'] = {'bound': 'hbm', 'kernel': f'find::mlp::gemm5_kernel<1> (Linear 256->256 + bias + ReLU over {N_FEET * N_VERTS} rows, fp16 MFMA operands, fp32 tensors in HBM)',
							   'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS, 'avg_kernel_ms': kms,
							   'bytes_per_launch': nbytes, 'traffic': GEMM5_TRAFFIC_BYTES if N_VERTS == 6890 else None,
							   'mfma_tflops': kflops / (kms * 1e-3) / 1e12}
		if world == 1 and not args.no_cpu_baseline:
			out['cpu_baseline'] = cpu_baseline()
		emit(out)
	if world > 1:
		dist.barrier()
	if dist.is_initialized():
		dist.destroy_process_group()


if __name__ == '__main__':
	main()
