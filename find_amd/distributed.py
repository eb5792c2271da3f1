"""Data parallelism over foot instances: one process per GPU, full model + latent tables replicated,
one flat fp32 gradient bucket all-reduced (sum, then / world) per step over RCCL/xGMI (`nccl` backend on ROCm) --
or `gloo` on CPU for tests.  The reference has no distributed path at all (run_expmt.py:80-99 only fans out
independent experiments), so this is new surface, kept deliberately small."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
	"""Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (as set by torch.distributed.run).
	Returns (rank, world, local_rank).  A single-process run needs no process group."""
	world = int(os.environ.get('WORLD_SIZE', '1'))
	rank = int(os.environ.get('RANK', '0'))
	local = int(os.environ.get('LOCAL_RANK', '0'))
	if world > 1 and not dist.is_initialized():
		os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
		os.environ.setdefault('MASTER_PORT', '29500')
		if backend is None:
			backend = 'nccl' if torch.cuda.is_available() else 'gloo'
		if backend == 'nccl':
			torch.cuda.set_device(local)
		dist.init_process_group(backend=backend, rank=rank, world_size=world)
	return rank, world, local


def shard_range(n_items, rank, world):
	"""Contiguous, near-equal shard [lo, hi) of n_items for this rank (feet are independent through the path)."""
	base, rem = divmod(n_items, world)
	lo = rank * base + min(rank, rem)
	return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
	"""Flat gradient bucket for a fixed parameter list.  `allreduce_()` averages .grad over ranks in ONE collective
	(MLP grads || latent-table grads, ~3.6 MB fp32).  Parameters without a gradient contribute zeros, so every rank
	issues an identically-shaped collective even when its shard touched different latent rows."""

	def __init__(self, params, group=None):
		self.params = [p for p in params if p.requires_grad]
		self.group = group
		self.numel = sum(p.numel() for p in self.params)
		p0 = self.params[0]
		self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
		self.views, o = [], 0
		for p in self.params:
			self.views.append(self.flat[o:o + p.numel()].view_as(p))
			o += p.numel()

	def allreduce_(self):
		if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
			return
		world = dist.get_world_size(self.group)
		have = [p.grad is not None for p in self.params]
		if not all(have):
			self.flat.zero_()
		torch._foreach_copy_([v for v, h in zip(self.views, have) if h], [p.grad for p, h in zip(self.params, have) if h])
		dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
		self.flat.div_(world)
		# copy back in one multi-tensor launch (a per-parameter loop is ~30 tiny kernels per step)
		torch._foreach_copy_([p.grad for p, h in zip(self.params, have) if h], [v for v, h in zip(self.views, have) if h])
		for p, v, h in zip(self.params, self.views, have):
			if not h:
				p.grad = v.clone()


def broadcast_parameters(params, src=0, group=None):
	"""Make every rank start from rank `src`'s values (one flat broadcast)."""
	if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
		return
	params = list(params)
	flat = torch.cat([p.data.reshape(-1).float() for p in params])
	dist.broadcast(flat, src=src, group=group)
	o = 0
	for p in params:
		p.data.copy_(flat[o:o + p.numel()].view_as(p))
		o += p.numel()
