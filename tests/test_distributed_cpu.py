"""N>1 path on CPU: world_size-2 `gloo` process groups exercising find_amd.distributed (contiguous foot shards, one flat
gradient all-reduce per step, parameter broadcast).  The GPU run uses the same code with the `nccl` (RCCL) backend."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	p = s.getsockname()[1]
	s.close()
	return p


def _worker(rank, world, port, q):
	os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
	torch.set_num_threads(1)
	from find_amd import distributed as fd
	r, w, _ = fd.init_from_env(backend='gloo')
	assert (r, w) == (rank, world)
	torch.manual_seed(100 + rank)  # ranks start from different weights ...
	net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
	table = torch.nn.Parameter(torch.zeros(8, 4))  # a latent table: each rank touches only the rows of its own feet
	unused = torch.nn.Parameter(torch.ones(3))     # never receives a gradient on any rank
	params = list(net.parameters()) + [table, unused]
	fd.broadcast_parameters(params, src=0)          # ... and agree after the broadcast
	w0 = torch.cat([p.detach().reshape(-1) for p in params])
	# global batch of 8 "feet", sharded contiguously
	g = torch.Generator().manual_seed(0)
	X = torch.randn(8, 6, generator=g)
	Y = torch.randn(8, 3, generator=g)
	lo, hi = fd.shard_range(8, rank, world)
	bucket = fd.GradBucket(params)
	loss = ((net(X[lo:hi]) + table[lo:hi].sum(1, keepdim=True) - Y[lo:hi]) ** 2).mean()  # batch-mean loss, as every FIND loss
	loss.backward()
	if rank == 0:
		bucket.allreduce_()
	else:   # the issue-now / wait-before-the-step form must be the same collective
		bucket.allreduce_(async_op=True)
		assert bucket._pending is not None
		bucket.wait()
	assert bucket._pending is None
	bucket.wait()   # (nothing pending: a no-op)
	grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
	q.put((rank, w0.numpy().copy(), grads.numpy().copy(), (lo, hi)))  # numpy: plain pickles (tensors would travel as shm handles)
	dist.barrier()
	dist.destroy_process_group()


def test_gloo_two_ranks_match_single_process():
	world = 2
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	port = _free_port()
	procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
	for p in procs:
		p.start()
	res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
	for p in procs:
		p.join(timeout=60)
		assert p.exitcode == 0
	(_, w0a, ga, sa), (_, w0b, gb, sb) = res
	w0a, ga, w0b, gb = [torch.from_numpy(a) for a in (w0a, ga, w0b, gb)]
	assert sa == (0, 4) and sb == (4, 8)
	assert torch.equal(w0a, w0b), 'broadcast_parameters must leave every rank with rank 0 weights'
	assert torch.allclose(ga, gb, atol=0, rtol=0), 'all ranks hold the same averaged gradient'
	# single-process reference on the full batch: mean of equal-size shard means == global mean
	torch.manual_seed(100)
	net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
	table = torch.nn.Parameter(torch.zeros(8, 4))
	unused = torch.nn.Parameter(torch.ones(3))
	params = list(net.parameters()) + [table, unused]
	assert torch.equal(torch.cat([p.detach().reshape(-1) for p in params]), w0a)
	g = torch.Generator().manual_seed(0)
	X = torch.randn(8, 6, generator=g)
	Y = torch.randn(8, 3, generator=g)
	loss = ((net(X) + table.sum(1, keepdim=True) - Y) ** 2).mean()
	loss.backward()
	ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
	assert torch.allclose(ga, ref, atol=1e-6), (ga - ref).abs().max()


def test_shard_range_covers_everything():
	from find_amd.distributed import shard_range
	for n in [1, 7, 16, 128, 129]:
		for w in [1, 2, 3, 8]:
			spans = [shard_range(n, r, w) for r in range(w)]
			assert spans[0][0] == 0 and spans[-1][1] == n
			assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
			sizes = [b - a for a, b in spans]
			assert max(sizes) - min(sizes) <= 1


def test_single_process_bucket_is_a_noop():
	from find_amd.distributed import GradBucket
	p = torch.nn.Parameter(torch.ones(3))
	(p * 2).sum().backward()
	GradBucket([p]).allreduce_()
	assert torch.equal(p.grad, torch.full((3,), 2.0))


# ------------------------------------------------------------------------------------------------ early prefix (GradBucket.arm_early)
class _ArenaLinear(torch.autograd.Function):
	"""x @ w.T + b whose backward asks find_amd.functional for its gradient buffers, as the HIP wrappers do (the bucket's arena protocol)."""

	@staticmethod
	def forward(ctx, x, w, b):
		ctx.save_for_backward(x, w, b)
		return x @ w.t() + b

	@staticmethod
	def backward(ctx, g):
		from find_amd import functional as FN
		x, w, b = ctx.saved_tensors
		gw, gb = FN._grads_like([w, b])
		gw.copy_(g.t() @ x)
		gb.copy_(g.sum(0))
		return g @ w, gw, gb


def _early_model(seed):
	torch.manual_seed(seed)
	w1, b1 = torch.nn.Parameter(torch.randn(5, 6) * 0.3), torch.nn.Parameter(torch.randn(5) * 0.1)
	w2, b2 = torch.nn.Parameter(torch.randn(3, 5) * 0.3), torch.nn.Parameter(torch.randn(3) * 0.1)
	table = torch.nn.Parameter(torch.randn(8, 3) * 0.1)   # a latent table: not part of the early prefix
	return [w1, b1, w2, b2], table


def _early_loss(ws, table, X, Y, lo, hi):
	w1, b1, w2, b2 = ws
	h = torch.relu(_ArenaLinear.apply(X[lo:hi], w1, b1))
	return ((_ArenaLinear.apply(h, w2, b2) + table[lo:hi] - Y[lo:hi]) ** 2).mean()


def _early_worker(rank, world, port, q):
	os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
	torch.set_num_threads(1)
	from find_amd import distributed as fd
	fd.init_from_env(backend='gloo')
	ws, table = _early_model(7)
	params = ws + [table]
	g = torch.Generator().manual_seed(0)
	X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
	lo, hi = fd.shard_range(8, rank, world)
	bucket = fd.GradBucket(params, arena=True, early=ws)
	bucket.arm_early(ws[0])
	assert bucket.params[:4] == ws and bucket.n_early == sum((p.numel() + 3) & ~3 for p in ws)
	out = {}

	def grads():
		return torch.cat([p.grad.reshape(-1) for p in params]).numpy().copy()

	# A: one backward per step -- the prefix leaves inside the backward, the table's part in allreduce_()
	_early_loss(ws, table, X, Y, lo, hi).backward()
	assert bucket.early_issued == 1 and bucket._early is not None
	bucket.allreduce_(async_op=True)
	bucket.wait()
	assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params[:4], bucket.views[:4]))   # (the table's gradient is autograd's own tensor: copy path)
	out['A'] = grads()
	# B: a second backward() before allreduce_() (micro-batches) would accumulate into gradients that are being all-reduced: refused, loudly
	for p in params:
		p.grad = None
	mid = (lo + hi) // 2
	(_early_loss(ws, table, X, Y, lo, mid) * 0.5).backward()
	assert bucket.early_issued == 2
	try:
		(_early_loss(ws, table, X, Y, mid, hi) * 0.5).backward()
		raise AssertionError('the second backward() of a step with an early prefix in flight was not refused')
	except RuntimeError as e:
		assert 'arm_early' in str(e)
	# C: ... and that step never calls allreduce_(): its early collective goes stale; the next backward drops it, behind its completion
	for p in params:
		p.grad = None
	_early_loss(ws, table, X, Y, lo, hi).backward()
	assert bucket.early_dropped == 1 and bucket.early_issued == 3
	bucket.allreduce_()
	out['C'] = grads()
	# D: the step after that takes the early path again
	for p in params:
		p.grad = None
	_early_loss(ws, table, X, Y, lo, hi).backward()
	assert bucket.early_issued == 4
	bucket.allreduce_()
	assert bucket.early_dropped == 1
	out['D'] = grads()
	# E: a gradient that is not its arena slot when allreduce_() looks (here: replaced by a clone of the LOCAL gradient, taken before the
	# collective could touch it -- gloo's work runs on the calling thread's queue: wait first, then restore) makes the prefix travel again
	for p in params:
		p.grad = None
	bucket._hook.remove(); bucket._hook = None      # (take the local gradients without the early collective ...)
	_early_loss(ws, table, X, Y, lo, hi).backward()
	local = [p.grad.clone() for p in ws]
	for p in params:
		p.grad = None
	bucket.taken = [False] * len(bucket.params); bucket._task = None
	bucket.arm_early(ws[0])                          # (... then the real thing)
	_early_loss(ws, table, X, Y, lo, hi).backward()
	assert bucket.early_issued == 5
	bucket._early[0].wait()
	for p, g in zip(ws, local):
		p.grad = g                                   # every early .grad now lives outside the arena, as after a pass that could not fold
	bucket.allreduce_()
	assert bucket.early_dropped == 2
	out['E'] = grads()
	q.put((rank, out))
	dist.barrier()
	dist.destroy_process_group()


def test_early_prefix_equals_one_collective_equals_single_process():
	"""GradBucket(early=...) + arm_early on gloo, world 2: prefix inside the backward + remainder == the full-batch gradient of one process;
	of the cases ADVICE r5 names, a second backward before allreduce_ is refused, a step without allreduce_ has its early collective
	dropped, a .grad that left the arena makes the prefix travel again -- and the average is right after each."""
	world = 2
	ctx = mp.get_context('spawn')
	q = ctx.Queue()
	port = _free_port()
	procs = [ctx.Process(target=_early_worker, args=(r, world, port, q)) for r in range(world)]
	for p in procs:
		p.start()
	res = sorted([q.get(timeout=60) for _ in range(world)], key=lambda t: t[0])
	for p in procs:
		p.join(timeout=60)
		assert p.exitcode == 0
	ws, table = _early_model(7)
	g = torch.Generator().manual_seed(0)
	X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
	_early_loss(ws, table, X, Y, 0, 8).backward()   # (no bucket in this process: _grads_like hands out plain tensors)
	ref = torch.cat([p.grad.reshape(-1) for p in ws + [table]])
	for case in 'ACDE':
		a, b = torch.from_numpy(res[0][1][case]), torch.from_numpy(res[1][1][case])
		assert torch.equal(a, b), f'case {case}: the ranks disagree'
		assert torch.allclose(a, ref, atol=1e-6), (case, (a - ref).abs().max())
