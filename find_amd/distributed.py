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

	def __init__(self, params, group=None, arena=None, early=None):
		"""early: parameters whose gradients are complete before the backward pass ends (FIND: the MLP's weights -- what follows their
		last weight-gradient kernel is the loss-side tail: registration, latent scatter).  They are laid out FIRST in the flat buffer and
		`arm_early()` hangs the collective over that prefix on the moment autograd accumulates the trigger parameter, so that it runs
		under the rest of the backward; `allreduce_()` then only has the remainder to send."""
		params = [p for p in params if p.requires_grad]
		early = [p for p in (early or []) if p.requires_grad]
		ids = {id(p) for p in early}
		self.params = early + [p for p in params if id(p) not in ids]
		self.n_early_params = len(early)
		self.group = group
		p0 = self.params[0]
		self.offsets, o = [], 0
		for p in self.params:
			self.offsets.append(o)
			o += (p.numel() + 3) & ~3  # 16-byte aligned slots; the padding stays zero
		self.numel = o
		self.n_early = self.offsets[self.n_early_params] if self.n_early_params < len(self.params) else o   # floats of the early prefix
		if not early:
			self.n_early = 0
		self._early = None      # (work, divisor) of the prefix collective issued inside this step's backward
		self._task = None       # autograd graph task the `taken` flags belong to
		self._hook = None
		self.early_issued = 0   # steps in which the prefix went out early (diagnostics / tests)
		self.early_dropped = 0  # early collectives whose result had to be dropped (diagnostics / tests)
		self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
		self._flat_early, self._flat_rest = self.flat[:self.n_early], self.flat[self.n_early:]
		# (per-slot constants of take(): it runs once per gradient tensor and backward pass -- 40 times per FIND step -- on the host's critical path)
		self._shapes = [tuple(p.shape) for p in self.params]
		self._strides = [tuple(p.contiguous().stride()) if p.dim() else () for p in self.params]
		self._ptrs = [self.flat.data_ptr() + 4 * o for o in self.offsets]
		from . import functional as _fn
		self._task_id = _fn._graph_task_id
		self.views = [self._view(i) for i in range(len(self.params))]
		self.taken = [False] * len(self.params)
		self.arena = p0.is_cuda if arena is None else arena
		if self.arena:
			from . import functional
			functional.register_grad_arena(self, self.params)

	def _comm(self):
		"""(process group, its all-reduce options, whether the backend averages) -- looked up once: the c10d python wrapper around a collective
		(argument checks, logging decorator, backend look-up) costs the host 15 us per call, and the data-parallel step is bound by its host."""
		c = getattr(self, '_comm_cache', None)
		if c is None:
			from torch.distributed import distributed_c10d as c10d
			pg = self.group if self.group is not None else c10d._get_default_group()
			avg = dist.get_backend(self.group) == 'nccl'
			opts = c10d.AllreduceOptions()
			opts.reduceOp = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
			c = self._comm_cache = (pg, opts, avg)
		return c

	def _all_reduce_async(self, tensor):
		pg, opts, avg = self._comm()
		return pg.allreduce([tensor], opts), avg

	def _view(self, i):
		return self.flat.as_strided(self._shapes[i], self._strides[i], self.offsets[i])   # (one op: slice + view were two)

	def take_many(self, slots, tensors):
		"""take() for the slots of `tensors` at once (a backward kernel wrapper asks for a whole layer list: 26 calls of take() were 40 us of the
		data-parallel step's host time)."""
		first = True
		out = []
		flat, shapes, strides, offsets, ptrs, params = self.flat, self._shapes, self._strides, self.offsets, self._ptrs, self.params
		for i, t in zip(slots, tensors):
			if first:
				v = self.take(i, t.shape, t.device)   # (the per-pass book-keeping runs here)
				first = False
				out.append(v)
				continue
			taken = self.taken
			if taken[i] or t.shape != shapes[i] or t.device != flat.device:
				out.append(None)
				continue
			g = params[i].grad
			if g is not None and g.data_ptr() == ptrs[i]:
				out.append(None)
				continue
			taken[i] = True
			out.append(flat.as_strided(shapes[i], strides[i], offsets[i]))
		return out

	def take(self, i, shape, device):
		"""Arena request from a backward kernel wrapper: a FRESH view of slot i (autograd adopts a gradient tensor only when nothing
		else references it), or None when the slot was handed out already in this backward pass or still backs a live .grad."""
		task = self._task_id()
		if task != self._task:
			# a new backward() call: the flags of the last one are history (a step that ended without allreduce_() must not leave them set --
			# the next step's hook would take a stale arena for a complete one); a slot that still backs a live .grad stays refused below
			self._task = task
			self.taken = [False] * len(self.params)
			if self._early is not None:
				# ... and so is an early collective nobody consumed.  It averaged the arena IN PLACE.  If the gradients it averaged are gone
				# (zero_grad: a step that never called allreduce_()), it is dropped -- behind its completion: this pass's kernels write the
				# same slots.  If they are still the parameters' .grad, this is a second backward() of the same step (micro-batches): it
				# would accumulate into gradients that are half-way through a collective -- refused, loudly
				live = [q for q, v in zip(self.params[:self.n_early_params], self.views[:self.n_early_params]) if q.grad is not None and q.grad.data_ptr() == v.data_ptr()]
				if live:
					raise RuntimeError('find_amd.distributed.GradBucket: backward() called again before allreduce_() while the early part of the bucket '
									   '(arm_early) is being all-reduced: gradient accumulation over several backward passes needs a bucket without `early`')
				self._early[0].wait()
				self._early = None
				self.early_dropped += 1
		if self.taken[i] or shape != self._shapes[i] or device != self.flat.device:
			return None
		g = self.params[i].grad
		if g is not None and g.data_ptr() == self._ptrs[i]:
			return None
		self.taken[i] = True
		return self.flat.as_strided(self._shapes[i], self._strides[i], self.offsets[i])

	def arm_early(self, trigger):
		"""Issue the all-reduce of the early prefix when autograd has accumulated `trigger` -- a parameter every backward node that writes an
		early gradient feeds (FIND: `base[0].weight`, which both MLP passes of a step reach: its AccumulateGrad node runs after the last of
		them has returned, i.e. behind its last weight-gradient kernel on the stream).  The collective averages the arena prefix in place: it
		goes out only if every early slot was handed to a backward kernel in this pass and the trigger's .grad IS its slot (else the prefix
		travels with the rest, as before).  allreduce_() checks that every early .grad still is its slot -- a pass that could not fold its
		gradients into the arena leaves them elsewhere: the prefix then travels again, from the real gradients --; a step that never reaches
		allreduce_() has its early collective dropped by the next backward, and a second backward() of the same step is refused (take())."""
		if self.n_early == 0 or self._hook is not None:
			return
		trigger._find_hooks_are_stream_safe = True   # (find_amd.functional: this hook joins deferred weight-gradient work itself before it reads the arena, so deferred joins stay allowed)
		t_index = next(i for i, p in enumerate(self.params) if p is trigger)

		def fire(_p):
			if not (dist.is_available() and dist.is_initialized()):
				return
			if self._early is not None or not all(self.taken[:self.n_early_params]):
				return
			g = trigger.grad
			if g is None or g.data_ptr() != self.views[t_index].data_ptr():
				return   # (autograd summed into a tensor of its own: a second MLP pass that could not fold into the arena)
			if self.arena:
				from . import functional
				# weight-gradient work a deferred find_mlp_bwd left on the context's side streams (functional: defer_join) writes the arena
				# too: the current stream -- and with it the collective issued behind it -- waits for it first.  (In FIND's step the main
				# pass's backward has joined it before this hook runs; a graph in which the deferred pass is the trigger's only feeder has not.)
				if functional._DEFERRED:
					functional._join_deferred()
				# ... and so does the fold of an MLP pass that ran on a stream of its own (model_with_loss.TEXTURE_STREAM): it adds into the
				# parked slots on THAT stream and reports no gradient, so autograd has not ordered this hook behind it
				for s in functional._CROSS_STREAMS:
					torch.cuda.current_stream(s.device).wait_stream(s)
			work, avg = self._all_reduce_async(self._flat_early)
			self._early = (work, None if avg else dist.get_world_size(self.group))
			self.early_issued += 1
		self._hook = trigger.register_post_accumulate_grad_hook(fire)

	def close(self):
		if self._hook is not None:
			self._hook.remove()
			self._hook = None
		if self.arena:
			from . import functional
			functional.unregister_grad_arena(self)
			self.arena = False

	def allreduce_(self, async_op=False):
		"""Average the gradients over the ranks: ONE collective on the flat bucket.  async_op=True issues it and returns -- the caller goes on
		enqueueing work that does not read the gradients and calls wait() before the optimiser step (RCCL runs the collective on its own
		stream behind everything the current stream held at the call; wait() makes the current stream wait for it, the host never blocks)."""
		self.taken = [False] * len(self.params)
		self._task = None
		self._pending = None
		early, self._early = self._early, None
		if not (dist.is_available() and dist.is_initialized()):
			return
		world = dist.get_world_size(self.group)
		if early is not None:
			# the prefix is on its way -- valid only if every early gradient still IS its arena slot (a pass that did not fold into the
			# arena, a hook that replaced .grad: the real gradients then sit elsewhere and are copied into the slots below)
			ok = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
					 for p, v in zip(self.params[:self.n_early_params], self.views[:self.n_early_params]))
			if not ok:
				early[0].wait()   # (finish it first: the copies below overwrite what it averaged)
				early = None
				self.early_dropped += 1
		missing, src, dst = [], [], []
		lo = self.n_early_params if early is not None else 0   # (the early prefix is on its way already: every one of its slots came out of the arena)
		for p, v in list(zip(self.params, self.views))[lo:]:
			if p.grad is None:
				missing.append((p, v))
			elif p.grad.data_ptr() != v.data_ptr():
				src.append(p.grad)
				dst.append(v)
		if missing:
			torch._foreach_zero_([v for _, v in missing])
		if src:
			torch._foreach_copy_(dst, src)
		rest = self._flat_rest if early is not None else self.flat
		work, avg = self._all_reduce_async(rest) if rest.numel() else (None, self._comm()[2])   # RCCL averages in place; gloo has no AVG: sum, then divide
		self._pending = (work, None if avg else world, src, dst, missing, early, rest)
		if not async_op:
			self.wait()

	def wait(self):
		"""Finish the collective allreduce_(async_op=True) issued: afterwards every parameter's .grad holds the average."""
		pend, self._pending = getattr(self, '_pending', None), None
		if pend is None:
			return
		work, div, src, dst, missing, early, rest = pend
		if early is not None:
			early[0].wait()
			if early[1] is not None:
				self.flat[:self.n_early].div_(early[1])
		if work is not None:
			work.wait()
			if div is not None:
				rest.div_(div)
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
