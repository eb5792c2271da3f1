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
			# FIND_DIST_BACKEND=gloo: diagnostic runs of the multi-rank code path on fewer GPUs than ranks (gloo stages device tensors through the host)
			backend = os.environ.get('FIND_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
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
	issues an identically-shaped collective even when its shard touched different latent rows.

	On a GPU the bucket registers its slots as the gradient arena of find_amd.functional: the backward kernels write each
	parameter's gradient straight into its slot and autograd adopts that view as `.grad`, so a step costs the collective alone
	(RCCL averages in place) -- no gather into the bucket and no scatter back.  A gradient that did not come out of the arena
	(second backward of the same parameter in one step, CPU tensors, foreign autograd nodes) takes the copy path."""

	def __init__(self, params, group=None, arena=None):
		self.params = [p for p in params if p.requires_grad]
		self.group = group
		p0 = self.params[0]
		self.offsets, o = [], 0
		for p in self.params:
			self.offsets.append(o)
			o += (p.numel() + 3) & ~3  # 16-byte aligned slots; the padding stays zero
		self.numel = o
		self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
		self.views = [self._view(i) for i in range(len(self.params))]
		self.taken = [False] * len(self.params)
		self.arena = p0.is_cuda if arena is None else arena
		if self.arena:
			from . import functional
			functional.register_grad_arena(self, self.params)

	def _view(self, i):
		p = self.params[i]
		return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)

	def take(self, i, shape, device):
		"""Arena request from a backward kernel wrapper: a FRESH view of slot i (autograd adopts a gradient tensor only when nothing
		else references it), or None when the slot was handed out already this step or still backs a live .grad."""
		p = self.params[i]
		if self.taken[i] or tuple(shape) != tuple(p.shape) or device != self.flat.device:
			return None
		g = p.grad
		if g is not None and g.data_ptr() == self.views[i].data_ptr():
			return None
		self.taken[i] = True
		return self._view(i)

	def close(self):
		if self.arena:
			from . import functional
			functional.unregister_grad_arena(self)
			self.arena = False

	def allreduce_(self, async_op=False):
		"""Average the gradients over the ranks: ONE collective on the flat bucket.  async_op=True issues it and returns -- the caller goes on
		enqueueing work that does not read the gradients and calls wait() before the optimiser step (RCCL runs the collective on its own
		stream behind everything the current stream held at the call; wait() makes the current stream wait for it, the host never blocks)."""
		self.taken = [False] * len(self.params)
		self._pending = None
		if not (dist.is_available() and dist.is_initialized()):
			return
		world = dist.get_world_size(self.group)
		missing, src, dst = [], [], []
		for p, v in zip(self.params, self.views):
			if p.grad is None:
				missing.append((p, v))
			elif p.grad.data_ptr() != v.data_ptr():
				src.append(p.grad)
				dst.append(v)
		if missing:
			torch._foreach_zero_([v for _, v in missing])
		if src:
			torch._foreach_copy_(dst, src)
		avg = dist.get_backend(self.group) == 'nccl'   # RCCL averages in place; gloo has no AVG: sum, then divide
		work = dist.all_reduce(self.flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
		self._pending = (work, None if avg else world, src, dst, missing)
		if not async_op:
			self.wait()

	def wait(self):
		"""Finish the collective allreduce_(async_op=True) issued: afterwards every parameter's .grad holds the average."""
		pend, self._pending = getattr(self, '_pending', None), None
		if pend is None:
			return
		work, div, src, dst, missing = pend
		work.wait()
		if div is not None:
			self.flat.div_(div)
		if src:  # copy back in one multi-tensor launch (a per-parameter loop is ~30 tiny kernels per step)
			torch._foreach_copy_(src, dst)
		for p, v in missing:
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
