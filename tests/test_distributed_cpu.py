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
