"""GPU side of the data-parallel step: the gradient arena -- backward kernels writing straight into the flat
all-reduce bucket -- and the RCCL collective itself on a one-rank `nccl` group (the multi-rank arithmetic is covered on CPU with
`gloo`, tests/test_distributed_cpu.py)."""
import os
import queue
import socket

import pytest
import torch


def _collect(q, procs, n, timeout=500):
	"""n results from the workers' queue -- or an immediate failure when a worker has died (not a 500-second wait for its answer)."""
	import time
	out, t0 = [], time.time()
	while len(out) < n:
		try:
			out.append(q.get(timeout=2))
		except queue.Empty:
			dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
			assert not dead, f'a worker exited with code {dead[0]}'
			assert time.time() - t0 < timeout, 'timed out waiting for the workers'
	return out

pytestmark = pytest.mark.gpu


def _model_and_step(n_feet=4, n_verts=1002, seed=0):
	from find_amd import synthetic
	dev = torch.device('cuda:0')
	model = synthetic.make_model(n_verts, train_size=n_feet, val_size=2, device=dev)
	lat = synthetic.latents(n_feet, seed=seed, device=dev)
	with torch.no_grad():
		model.shapevec.data.copy_(lat['shapevec'])
		model.texvec.data.copy_(lat['texvec'])
		model.posevec.data.copy_(lat['posevec'])
		model.reg.data.copy_(lat['reg'])
	idx = torch.arange(n_feet, device=dev)
	params = [p for p in model.parameters() if p.requires_grad]

	def backward_once(scale=1.0):
		batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx], reg_train=model.reg[idx])
		res = model.get_meshes_from_batch(batch, is_train=True)
		(scale * ((res['verts'] ** 2).sum() + (res['col'] ** 2).sum())).backward()

	return model, params, backward_once


def _grads(params):
	return [None if p.grad is None else p.grad.detach().clone() for p in params]


def test_backward_writes_into_the_bucket_and_matches_plain_gradients():
	from find_amd import distributed as fd
	model, params, backward_once = _model_and_step()
	backward_once()
	plain = _grads(params)
	for p in params:
		p.grad = None
	bucket = fd.GradBucket(params)
	try:
		assert bucket.arena
		backward_once()
		aliased = 0
		for p, v, g0 in zip(params, bucket.views, plain):
			assert (p.grad is None) == (g0 is None)
			if p.grad is None:
				continue
			assert torch.equal(p.grad, g0)
			aliased += int(p.grad.data_ptr() == v.data_ptr())
		# every MLP weight and every latent table gets its gradient from a HIP backward: all of them must live in the bucket
		assert aliased == sum(g is not None for g in plain), f'{aliased} gradients in the bucket'
		# the flat buffer holds exactly those values (padding between slots stays zero)
		total = sum(float(g.double().sum()) for g in plain if g is not None)
		assert abs(float(bucket.flat.double().sum()) - total) <= 1e-6 * max(1.0, abs(total))
		# second backward before the step ends: slots are handed out once, autograd accumulates in place
		backward_once(0.5)
		for p, g0 in zip(params, plain):
			if g0 is not None:
				assert torch.allclose(p.grad, 1.5 * g0, rtol=1e-5, atol=1e-6 * float(g0.abs().max()))
		# next step without clearing .grad: the live gradients must not be overwritten by the kernels
		bucket.allreduce_()   # no process group: ends the step only
		backward_once(0.5)
		for p, g0 in zip(params, plain):
			if g0 is not None:
				assert torch.allclose(p.grad, 2.0 * g0, rtol=1e-5, atol=1e-6 * float(g0.abs().max()))
	finally:
		bucket.close()


@pytest.mark.timeout(300)
def test_rccl_one_rank_allreduce_in_place():
	import torch.distributed as dist
	from find_amd import distributed as fd
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	torch.cuda.set_device(0)
	dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
	try:
		model, params, backward_once = _model_and_step(seed=1)
		extra = torch.nn.Parameter(torch.ones(5, device='cuda'))  # never receives a gradient
		bucket = fd.GradBucket(params + [extra])
		try:
			backward_once()
			want = _grads(params)
			bucket.allreduce_()   # RCCL average over one rank, in place on the bucket
			torch.cuda.synchronize()
			for p, g0 in zip(params, want):  # (tables the step never read -- the validation latents -- come back as zeros)
				assert torch.equal(p.grad, g0 if g0 is not None else torch.zeros_like(p))
			assert torch.equal(extra.grad, torch.zeros(5, device='cuda'))
			# a gradient that did not come out of the arena takes the copy path
			for p in params:
				p.grad = None
			extra.grad = torch.full((5,), 3.0, device='cuda')
			backward_once()
			bucket.allreduce_()
			torch.cuda.synchronize()
			assert torch.equal(extra.grad, torch.full((5,), 3.0, device='cuda'))
			for p, g0 in zip(params, want):
				assert torch.equal(p.grad, g0 if g0 is not None else torch.zeros_like(p))
		finally:
			bucket.close()
	finally:
		dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ two ranks, the real model
def _dp_worker(rank, world, port, n_verts, q):
	"""One data-parallel rank running the REAL HIP model (both ranks share cuda:0; `gloo` carries the CUDA bucket): 8 of the 16 feet,
	find_mlp_bwd with its five internal side streams writing straight into the GradBucket arena, one flat all-reduce."""
	import os
	os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
	import torch.distributed as dist
	from find_amd import distributed as fd
	from find_amd import synthetic
	torch.cuda.set_device(0)
	r, w, _ = fd.init_from_env(backend='gloo')
	dev = torch.device('cuda:0')
	n_total = 16
	model = synthetic.make_model(n_verts, train_size=n_total, val_size=2, device=dev)
	lat = synthetic.latents(n_total, seed=5, device=dev)
	with torch.no_grad():
		if rank != 0:   # ranks start apart and must agree after the broadcast
			for p in model.parameters():
				if p.requires_grad:
					p.add_(0.01)
		else:
			for k in ('shapevec', 'texvec', 'posevec', 'reg'):
				getattr(model, k).data.copy_(lat[k])
	fd.broadcast_parameters([p for p in model.parameters() if p.is_floating_point()])
	params = [p for p in model.parameters() if p.requires_grad]
	# the MLP weights' part of the bucket leaves INSIDE the backward (GradBucket.arm_early: when autograd accumulates the first trunk weight,
	# behind the MLP's last weight-gradient kernel), the latent tables' part behind it: same averaged gradients as one collective
	mlp_w = [p for seq in (model.base, model.mlp_disp, model.mlp_col) for p in seq.parameters()]
	bucket = fd.GradBucket(params, early=mlp_w)
	bucket.arm_early(model.base[0].weight)
	assert bucket.arena and 3.4e6 < bucket.n_early * 4 < bucket.numel * 4
	lo, hi = fd.shard_range(n_total, rank, world)
	idx = torch.arange(lo, hi, device=dev)
	batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx], reg_train=model.reg[idx])
	res = model.get_meshes_from_batch(batch, is_train=True)
	loss = ((res['verts'] ** 2).sum() + (res['col'] ** 2).sum()) / (hi - lo)   # a batch mean, as every FIND loss
	loss.backward()
	in_arena = sum(int(p.grad is not None and p.grad.data_ptr() == v.data_ptr()) for p, v in zip(bucket.params, bucket.views))
	assert bucket.early_issued == 1, 'the weights\' part of the bucket did not go out inside the backward'
	bucket.allreduce_()
	torch.cuda.synchronize()
	grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).cpu()
	q.put((rank, grads.numpy().copy(), in_arena, (lo, hi)))
	dist.barrier()
	bucket.close()
	dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_of_the_find_model_equal_one_process_on_16_feet():
	"""SURVEY 8e on the FIND model itself: two ranks x 8 feet, gradients averaged through GradBucket (arena mode) == one process x 16
	feet.  All FIND losses are batch means, so the mean of the two shard means is the global mean."""
	import torch.multiprocessing as mp
	from find_amd import synthetic
	n_verts, world = 6890, 2
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	procs = [ctx.Process(target=_dp_worker, args=(r, world, port, n_verts, q)) for r in range(world)]
	for p in procs:
		p.start()
	res = sorted(_collect(q, procs, world), key=lambda t: t[0])
	for p in procs:
		p.join(timeout=120)
		assert p.exitcode == 0
	(_, ga, na, sa), (_, gb, nb, sb) = res
	assert sa == (0, 8) and sb == (8, 16)
	ga, gb = torch.from_numpy(ga), torch.from_numpy(gb)
	assert torch.equal(ga, gb), 'both ranks hold the same averaged gradient'
	# single process, all 16 feet
	dev = torch.device('cuda:0')
	model = synthetic.make_model(n_verts, train_size=16, val_size=2, device=dev)
	lat = synthetic.latents(16, seed=5, device=dev)
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			getattr(model, k).data.copy_(lat[k])
	params = [p for p in model.parameters() if p.requires_grad]
	idx = torch.arange(16, device=dev)
	batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx], reg_train=model.reg[idx])
	out = model.get_meshes_from_batch(batch, is_train=True)
	(((out['verts'] ** 2).sum() + (out['col'] ** 2).sum()) / 16).backward()
	# every gradient the HIP backward produces went through the arena on both ranks (MLP weights + the four train tables)
	n_hip = sum(p.grad is not None for p in params)
	assert na == n_hip and nb == n_hip, (na, nb, n_hip)
	o = 0
	worst = 0.0
	for p in params:
		g = ga[o:o + p.numel()].view_as(p)
		o += p.numel()
		want = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
		scale = max(1e-6, want.abs().max().item())
		err = (g - want).abs().max().item() / scale
		worst = max(worst, err)
		# fp32 sums in a different order (two partial batches, then their mean) against one pass over 16 feet
		if err >= 1e-5:   # say where: a handful of elements of one weight gradient was the signature of the co-residence fault (mlp.hip: waves above 256 registers)
			d = (g - want).abs()
			bad = (d > 1e-5 * scale).nonzero()
			raise AssertionError(f'{tuple(p.shape)}: relative error {err:.3e}; {bad.shape[0]} element(s) off, first at {bad[:8].tolist()}, '
								 f'values {[(float(g[tuple(i)]), float(want[tuple(i)])) for i in bad[:4]]}')
	print(f'2-rank DP vs 1 process: worst relative gradient difference {worst:.2e}')


@pytest.mark.timeout(900)
def test_one_rank_rccl_step_keeps_the_stream_layout_and_the_step_time():
	"""What a rank of `bench.py --gpus N` runs, on one GPU (tools/dp_one_rank.py, a fresh process per mode because HIP maps streams onto
	its hardware queues in creation order and RCCL creates its own first): after init_process_group('nccl') the MLP context's side streams
	Q, T1, T2 still sit off the caller's hardware queue, and the headline step through broadcast + gradient bucket + all-reduce costs what
	the plain step costs (round 2 found 3.47 against 3.25 ms here before the layout was probed).  Bound 8 %: the data-parallel step carries
	0.12 - 0.13 ms of host work the plain one has not (two collectives through the process group, the arena's book-keeping: tools/r6_dp_phases.py)
	and the second collective sits between the backward and the optimiser; round 6 measured x1.04 - 1.06 box to box once the plain step
	had dropped to 1.58 ms (5 % was the bound while the plain step took 1.73), and two processes on one box differ by 1-2 % on their own."""
	import os
	import re
	import subprocess
	import sys
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
	def measure():
		res = {}
		for mode in ('plain', 'dp'):
			r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dp_one_rank.py'), '60', mode], capture_output=True, text=True, env=env, timeout=400, cwd=root)
			assert r.returncode == 0, (mode, r.stdout[-1500:], r.stderr[-1500:])
			groups = [int(x) for x in re.search(r'\[caller, Q, T1, T2, R\] = \[([^\]]*)\]', r.stdout).group(1).split(',')]
			ms = [float(x) for x in re.findall(rf'{mode}: ([0-9.]+) ms/step', r.stdout)]
			assert len(ms) == 2, r.stdout
			res[mode] = (groups, min(ms))
		return res

	# A layout regression is there in every run (3.47 against 3.25 ms, every time); a box whose host was busy for one of the two processes is
	# not (seen once in a full-suite run: x1.47, x1.01-1.02 in the runs before and after): up to three attempts, every one printed.
	for attempt in range(3):
		res = measure()
		for mode, (g, _) in res.items():
			assert g[1] != g[0] and g[2] != g[0] and g[3] != g[0], f'{mode}: a side stream shares the hardware queue of the caller: {g}'
		ratio = res['dp'][1] / res['plain'][1]
		print(f"one-rank RCCL step {res['dp'][1]:.3f} ms against {res['plain'][1]:.3f} ms plain: x{ratio:.3f}; queue groups {res['dp'][0]} / {res['plain'][0]}")
		if ratio < 1.08:
			break
	assert ratio < 1.08, (res, ratio)


def _train3d_grads(lo, hi, n_total, n_verts, gt_verts, bucket_early):
	"""Gradients of ONE train_3d.yaml network-stage step (chamf + smooth + texture: TWO MLP passes, the texture pass's weight gradients
	deferred on the side streams and folded into the main pass's) on the feet [lo, hi) of n_total, with fixed sampler draws."""
	import sys
	sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
	from test_gpu_train3d import FixedDraws
	from find_amd import distributed as fd
	from find_amd import synthetic
	from find_amd.model_with_loss import ModelWithLoss
	from find_amd.opts import Opts
	from find_amd.structures import Meshes, TexturesVertex
	from find_amd.train_utils import backward_on_this_thread, sample_latent_vectors
	dev = torch.device('cuda:0')
	opts = Opts(chamf_loss=True, smooth_loss=True, texture_loss=True, use_pose_code=True)
	mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=n_total, val_size=2,
						shapevec_size=100, texvec_size=100, posevec_size=100, template_mesh_loc=None)
	g = torch.Generator().manual_seed(1234)
	with torch.no_grad():
		mwl.model.mlp_disp[-1].weight.copy_(torch.randn(mwl.model.mlp_disp[-1].weight.shape, generator=g) * 0.01)
		mwl.model.mlp_disp[-1].bias.copy_(torch.randn(3, generator=g) * 0.01)
	mwl = mwl.to(dev)
	m = mwl.model
	v, f = synthetic.template(n_verts)
	m.set_template(v.to(dev), f.to(dev))
	lat = synthetic.latents(n_total, seed=5, device=dev)
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			getattr(m, k).data.copy_(lat[k])
	gv, gf, gc = synthetic.gt_feet(n_total, gt_verts, seed=3, device=dev)
	gc = gc.clamp(0.05, 0.95)
	F_gt, F_t = gf.shape[0], f.shape[0]
	gd = torch.Generator().manual_seed(9)
	full = [(torch.randint(0, F_gt, (n_total, 5000), generator=gd), torch.rand(n_total, 5000, 2, generator=gd)),
			(torch.randint(0, F_t, (n_total, 5000), generator=gd), torch.rand(n_total, 5000, 2, generator=gd)),
			(torch.randint(0, F_gt, (n_total, 1000), generator=gd), torch.rand(n_total, 1000, 2, generator=gd))]
	draws = [(a[lo:hi].to(torch.int32).to(dev), b[lo:hi].to(dev)) for a, b in full]
	params = [p for p in m.parameters() if p.requires_grad]
	bucket = None
	if bucket_early:
		mlp_w = [p for seq in (m.base, m.mlp_disp, m.mlp_col) for p in seq.parameters()]
		bucket = fd.GradBucket(params, early=mlp_w)
		bucket.arm_early(m.base[0].weight)
	batch = dict(mesh=Meshes(gv[lo:hi].contiguous(), gf, TexturesVertex(gc[lo:hi].contiguous())), idx=torch.arange(lo, hi, device=dev),
				 name=[f'{i:04d}' for i in range(lo, hi)])
	batch.update(sample_latent_vectors(batch, m.latent_vectors_train))
	with FixedDraws(draws), backward_on_this_thread():
		loss, _ = mwl(batch, 0, opts, chamf=True, smooth=True, texture=True)
		loss.backward()
	if bucket is not None:
		assert bucket.early_issued == 1, 'the MLP weights\' part of the bucket did not leave inside the backward'
		bucket.allreduce_()
	torch.cuda.synchronize()
	grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).cpu()
	if bucket is not None:
		bucket.close()
	return grads, [tuple(p.shape) for p in params]


def _train3d_worker(rank, world, port, n_verts, gt_verts, q):
	import os
	os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
	import torch.distributed as dist
	from find_amd import distributed as fd
	torch.cuda.set_device(0)
	fd.init_from_env(backend='gloo')
	lo, hi = fd.shard_range(16, rank, world)
	grads, _ = _train3d_grads(lo, hi, 16, n_verts, gt_verts, bucket_early=True)
	q.put((rank, grads.numpy().copy()))
	dist.barrier()
	dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_of_the_two_pass_training_step_equal_one_process():
	"""The step the headline times, data-parallel: chamf + smooth + texture on 2 x 8 feet, the MLP weights' part of the gradient bucket
	all-reduced INSIDE the backward (GradBucket.arm_early), against one process on the 16 feet with the same sampler draws.  The step
	runs the MLP twice; the texture pass's weight gradients are still running on the side streams when its backward node returns and the
	main pass's backward adds its own to them -- the early collective must come behind both (an all-reduce that left after the first
	would average half a gradient and have the other half added on top: a factor-of-two error in every MLP weight)."""
	import torch.multiprocessing as mp
	n_verts, gt_verts, world = 1002, 1002, 2
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	procs = [ctx.Process(target=_train3d_worker, args=(r, world, port, n_verts, gt_verts, q)) for r in range(world)]
	for p in procs:
		p.start()
	res = sorted(_collect(q, procs, world), key=lambda t: t[0])
	for p in procs:
		p.join(timeout=120)
		assert p.exitcode == 0
	ga, gb = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
	assert torch.equal(ga, gb), 'both ranks hold the same averaged gradient'
	want, shapes = _train3d_grads(0, 16, 16, n_verts, gt_verts, bucket_early=False)
	o = 0
	for sh in shapes:
		n = int(torch.tensor(sh).prod()) if sh else 1
		g, w = ga[o:o + n], want[o:o + n]
		o += n
		scale = max(1e-8, w.abs().max().item())
		err = (g - w).abs().max().item() / scale
		# fp32 sums in a different order (two half batches, then their mean); the texture loss's mean runs over the masked samples of the batch
		assert err < 2e-4, (sh, err)
