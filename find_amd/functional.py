"""torch.autograd bindings of the HIP hot path (thin: argument checks, workspace allocation, C-ABI call).

Every function requires CUDA(HIP) fp32 tensors and raises otherwise -- there is no CPU or eager fallback."""
import ctypes

import torch

from . import _lib
from ._lib import MlpGrads, MlpParams, check, current_stream, ptr


def _require_gpu(*tensors):
	for t in tensors:
		if t is None:
			continue
		if not t.is_cuda:
			raise RuntimeError('find_amd: the HIP path needs tensors on a ROCm device (got a CPU tensor); '
							   'there is no CPU fallback -- use oracle/ only for testing')
		if t.dtype != torch.float32 and t.is_floating_point():
			raise RuntimeError(f'find_amd: fp32 tensors required, got {t.dtype}')


def _c(t):
	return None if t is None else t.contiguous()


def _ws(nbytes, device):
	return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------------- gradient arena
# distributed.GradBucket registers the parameters it reduces here: the backward kernels then write those gradients straight into
# the bucket's flat buffer (autograd adopts the returned views as .grad), so the all-reduce needs no gather / scatter copies.
_GRAD_ARENA = {}   # (data_ptr, numel) of a parameter -> (bucket, slot)


def register_grad_arena(bucket, params):
	for i, prm in enumerate(params):
		_GRAD_ARENA[(prm.data_ptr(), prm.numel())] = (bucket, i)


def unregister_grad_arena(bucket):
	for k in [k for k, (b, _) in _GRAD_ARENA.items() if b is bucket]:
		del _GRAD_ARENA[k]


def _grad_for(key, shape, device):
	"""Uninitialised gradient buffer for the input with arena key (data_ptr, numel): its slot of a registered bucket when that slot
	is free this step, else a fresh tensor."""
	if _GRAD_ARENA:
		ent = _GRAD_ARENA.get(key)
		if ent is not None:
			v = ent[0].take(ent[1], shape, device)
			if v is not None:
				return v
	return torch.empty(shape, device=device, dtype=torch.float32)


def _grads_like(tensors):
	"""Gradient buffers for a list of inputs: arena slots where registered, otherwise views of ONE fresh allocation."""
	out = [None] * len(tensors)
	rest = []
	if _GRAD_ARENA:
		ents = [_GRAD_ARENA.get((t.data_ptr(), t.numel())) for t in tensors]
		b0 = next((e[0] for e in ents if e is not None), None)
		if b0 is not None and all(e is not None and e[0] is b0 for e in ents):   # (the usual case: one bucket holds them all)
			return _fill_rest(b0.take_many([e[1] for e in ents], tensors), tensors)
		for i, (t, ent) in enumerate(zip(tensors, ents)):
			out[i] = ent[0].take(ent[1], t.shape, t.device) if ent is not None else None
	return _fill_rest(out, tensors)


def _fill_rest(out, tensors):
	"""Views of ONE fresh allocation for the entries of `out` that are still None."""
	rest = [i for i, v in enumerate(out) if v is None]
	if rest:
		# 16-byte aligned slots (the slab reduce stores float4)
		offs, n = [], 0
		for i in rest:
			offs.append(n)
			n += (tensors[i].numel() + 3) & ~3
		flat = torch.empty(n, dtype=tensors[rest[0]].dtype, device=tensors[rest[0]].device)
		for i, o in zip(rest, offs):
			shape = tensors[i].shape
			out[i] = flat.as_strided(shape, _contiguous_strides(shape), o)   # (one op per tensor: slice + view were two; ._base is `flat` either way)
	return out


_STRIDES = {}


def _contiguous_strides(shape):
	st = _STRIDES.get(shape)
	if st is None:
		acc, rev = 1, []
		for d in reversed(shape):
			rev.append(acc)
			acc *= d
		st = _STRIDES[shape] = tuple(reversed(rev))
	return st


# ----------------------------------------------------------------------------------------------- MLP
class MLPSpec:
	"""Static description of the network (layer counts / sizes) shared by forward and backward."""

	def __init__(self, n_trunk, n_disp, n_col, pe_size, lat_disp, lat_col, in_dim=3, width=256, precision=None):
		self.n_trunk, self.n_disp, self.n_col = n_trunk, n_disp, n_col
		self.pe_size, self.lat_disp, self.lat_col = pe_size, lat_disp, lat_col
		self.in_dim, self.width = in_dim, width
		self.precision = precision   # None: the process default (set_mlp_precision); 'fp32' / 'bf16x3' / 'fp16': this model only

	@property
	def n_weights(self):
		return 2 * (self.n_trunk + self.n_disp + 1 + self.n_col + 1)


_PRECISION_CODE = {None: 0, 'fp32': 1, 'fp16': 2, 'bf16x3': 3}


def _fill_params(spec, B, avg_col, weights):
	p = MlpParams()
	p.precision = _PRECISION_CODE[spec.precision if spec.precision is not None else _MLP_PRECISION]
	p.width, p.in_dim, p.pe_size = spec.width, spec.in_dim, spec.pe_size
	p.n_trunk, p.n_disp, p.n_col = spec.n_trunk, spec.n_disp, spec.n_col
	p.lat_disp, p.lat_col = spec.lat_disp, spec.lat_col
	p.B = None if B is None else B.data_ptr()
	p.avg_col = None if avg_col is None else avg_col.data_ptr()
	it = iter(weights)
	for i in range(spec.n_trunk):
		p.trunk_w[i] = next(it).data_ptr()
		p.trunk_b[i] = next(it).data_ptr()
	for i in range(spec.n_disp + 1):
		p.disp_w[i] = next(it).data_ptr()
		p.disp_b[i] = next(it).data_ptr()
	for i in range(spec.n_col + 1):
		p.col_w[i] = next(it).data_ptr()
		p.col_b[i] = next(it).data_ptr()
	return p


class _MLP(torch.autograd.Function):
	"""disp, col = MLP(pos; latents, weights)   -- find_mlp_fwd / find_mlp_bwd."""

	@staticmethod
	def forward(ctx, spec, pos, lat_disp, lat_col, B, avg_col, *weights):
		spec, heads, defer = spec if isinstance(spec, tuple) else (spec, 3, False)   # heads: bit 0 = displacement, bit 1 = colour
		if len(weights) != spec.n_weights:
			raise RuntimeError(f'find_amd.mlp: expected {spec.n_weights} weight tensors, got {len(weights)}')
		_require_gpu(pos, lat_disp, lat_col, B, avg_col, *weights)
		if pos.requires_grad:
			raise RuntimeError('find_amd.mlp: gradient w.r.t. positions is not part of the FIND path (template / sampled '
							   'GT points carry no grad, model.py:285, losses.py:39-51)')
		L = _lib.lib()
		pos, lat_disp, lat_col, B, avg_col = _c(pos), _c(lat_disp), _c(lat_col), _c(B), _c(avg_col)
		leaves = all(w.is_leaf for w in weights)   # (the fold in backward() parks gradients at AccumulateGrad nodes: leaves only)
		weights = tuple(w.contiguous() for w in weights)
		if pos.dim() != 3 or pos.shape[-1] != spec.in_dim:
			raise RuntimeError(f'find_amd.mlp: pos must be (B|1, V, {spec.in_dim}), got {tuple(pos.shape)}')
		pos_batch, V, _ = pos.shape
		n_feet = pos_batch
		batches = set()
		for name, lat, width, bit in (('lat_disp', lat_disp, spec.lat_disp, 1), ('lat_col', lat_col, spec.lat_col, 2)):
			if lat is None:
				if width != 0 and heads & bit:   # (the latents of a head this call does not evaluate may be left out)
					raise RuntimeError(f'find_amd.mlp: {name} is missing but the head expects {width} latent columns')
				continue
			if lat.dim() != 2 or lat.shape[1] != width:
				raise RuntimeError(f'find_amd.mlp: {name} must be (n_feet, {width}), got {tuple(lat.shape)}')
			batches.add(int(lat.shape[0]))
		if len(batches) > 1:   # (the reference fails in torch.cat / expand; here it would be an out-of-bounds device read)
			raise RuntimeError(f'find_amd.mlp: lat_disp and lat_col have different batch sizes {sorted(batches)}')
		if batches:
			n_feet = batches.pop()
		if pos_batch != 1 and pos_batch != n_feet:
			raise RuntimeError(f'find_amd.mlp: pos batch {pos_batch} does not match latent batch {n_feet}')
		p = _fill_params(spec, B, avg_col, weights)
		save = any(ctx.needs_input_grad)
		nbytes = L.find_mlp_ws_bytes(ctypes.byref(p), pos_batch, n_feet, V, int(save))
		if nbytes < 0:
			check(-1, 'find_mlp_ws_bytes')
		ws = _ws(nbytes, pos.device)
		disp = torch.empty(n_feet, V, 3, device=pos.device, dtype=torch.float32) if heads & 1 else None
		col = torch.empty(n_feet, V, 3, device=pos.device, dtype=torch.float32) if heads & 2 else None
		check(L.find_mlp_fwd(_lib.ctx(pos.device), ctypes.byref(p), ptr(pos), pos_batch, n_feet, V, ptr(lat_disp), ptr(lat_col), ptr(disp), ptr(col),
							 ptr(ws), ws.numel(), int(save), current_stream(pos.device)), 'find_mlp_fwd')
		if save:
			ctx.spec = spec
			ctx.precision = p.precision
			ctx.dims = (pos_batch, n_feet, V)
			ctx.ws = ws
			ctx.leaves = leaves
			ctx.defer = bool(defer)
			# an output nothing reads hands None to backward, not a tensor of zeros: the main pass of a train_3d.yaml step never reads its
			# colours (no pixel loss), and a materialised zero gradient sent the whole colour head through its backward on zeros
			ctx.set_materialize_grads(False)
			ctx.save_for_backward(pos, lat_disp, lat_col, B, avg_col, *weights)
		return disp, col

	@staticmethod
	def backward(ctx, g_disp, g_col):
		L = _lib.lib()
		spec = ctx.spec
		pos, lat_disp, lat_col, B, avg_col, *weights = ctx.saved_tensors
		pos_batch, n_feet, V = ctx.dims
		g_disp, g_col = _c(g_disp), _c(g_col)
		p = _fill_params(spec, B, avg_col, weights)
		p.precision = ctx.precision   # the arithmetic the forward ran in, whatever the default is by now
		nt, nd = 2 * spec.n_trunk, 2 * (spec.n_disp + 1)
		# Which gradients exist.  A head nothing read (no upstream gradient) contributes nothing: its weights and latents get None, as
		# in the reference, where such a head never enters the graph (the texture pass reads 'col' only, losses.py:45-51; the main pass of
		# train_3d.yaml never reads its colours) -- an exact-zero tensor instead would still advance Adam's moments and apply weight decay.
		# No weight asks for a gradient at all: frozen network, latent gradients only (find_hip.h: find_mlp_grads).
		frozen = not any(ctx.needs_input_grad[6:])
		live = [False] * len(weights)
		if not frozen and (g_disp is not None or g_col is not None):
			live[:nt] = [True] * nt
			if g_disp is not None:
				live[nt:nt + nd] = [True] * nd
			if g_col is not None:
				live[nt + nd:] = [True] * (len(weights) - nt - nd)
		idx = [i for i, on in enumerate(live) if on]
		grads = [None] * len(weights)
		for i, g in zip(idx, _grads_like([weights[i] for i in idx]) if idx else []):
			grads[i] = g
		# A second backward through the same weights inside one backward() call (FIND's texture pass and its main pass) would have
		# autograd add 26 pairs of gradient tensors one launch each (InputBuffer accumulation).  The first call's gradients are still
		# parked at the parameters' AccumulateGrad nodes at that point -- those nodes run only after every MLP node that feeds them --, so
		# this call adds its own to them with one multi-tensor launch and reports "no contribution" for those weights instead.  That
		# rests on the parameters being leaves whose gradients nothing else touches before the engine accumulates them: leaves only, no
		# double backward (create_graph), same graph task, same stream.
		task = _graph_task_id()
		key = tuple(w.data_ptr() for w in weights)
		pending = _PENDING_WGRADS.get(key)
		can_park = ctx.leaves and not torch.is_grad_enabled() and task >= 0 and all(ctx.needs_input_grad[6:])
		# (on the stream that produced the parked gradients, or -- the texture term on a stream of its own beside the main pass,
		# model_with_loss.TEXTURE_STREAM -- on another one: the sum then waits for the event recorded behind the first call, and the
		# stream that called backward() waits for this one when the pass ends, _join_cross)
		cur = torch.cuda.current_stream(pos.device)
		fold = pending is not None and can_park and pending[0] == task
		cross = fold and pending[2] != cur
		g_lat_disp = torch.empty_like(lat_disp) if (lat_disp is not None and g_disp is not None and ctx.needs_input_grad[2]) else None
		g_lat_col = torch.empty_like(lat_col) if (lat_col is not None and g_col is not None and ctx.needs_input_grad[3]) else None
		if not idx and g_lat_disp is None and g_lat_col is None:
			return (None,) * (6 + len(weights))
		G = MlpGrads()
		it = iter(grads)
		for i in range(spec.n_trunk):
			G.trunk_w[i] = ptr(next(it))
			G.trunk_b[i] = ptr(next(it))
		for i in range(spec.n_disp + 1):
			G.disp_w[i] = ptr(next(it))
			G.disp_b[i] = ptr(next(it))
		for i in range(spec.n_col + 1):
			G.col_w[i] = ptr(next(it))
			G.col_b[i] = ptr(next(it))
		G.lat_disp = ptr(g_lat_disp)
		G.lat_col = ptr(g_lat_col)
		sb = L.find_mlp_bwd_scratch_bytes(ctypes.byref(p), pos_batch, n_feet, V)
		scratch = _ws(sb, pos.device)
		# Deferred join (find_hip.h: "defer_join"): this call's weight gradients may keep running on the context's side streams while
		# autograd goes on to the next node -- the texture pass's are 0.28 ms that nothing reads before the main pass folds its own into
		# them.  Safe only while nobody else can touch them first: leaf weights without an existing .grad (AccumulateGrad then just
		# adopts the tensor: no kernel), no double backward, the first MLP backward of the task; every buffer the side streams still use
		# is kept alive until the join, which happens at the end of the next find_mlp_bwd or -- at the latest -- when backward() ends.
		h = _lib.ctx(pos.device)
		# (... and nobody is TOLD about them first: a tensor hook or a post-accumulate-grad hook on a weight would read its gradient as
		# soon as this node returns -- the texture pass alone in a graph, hooks of a wrapper such as DDP --)
		defer = (DEFER_WGRAD_JOIN and ctx.defer and can_park and not fold and (pending is None or pending[0] != task) and idx
				 and all(w.grad is None and not w._backward_hooks
						 and (not getattr(w, '_post_accumulate_grad_hooks', None) or getattr(w, '_find_hooks_are_stream_safe', False)) for w in weights))
		if defer:
			check(L.find_ctx_set(h, b'defer_join', 1), 'find_ctx_set(defer_join)')
		check(L.find_mlp_bwd(h, ctypes.byref(p), ptr(pos), pos_batch, n_feet, V, ptr(lat_disp), ptr(lat_col), ptr(g_disp), ptr(g_col),
							 ptr(ctx.ws), ctx.ws.numel(), ptr(scratch), scratch.numel(), ctypes.byref(G),
							 current_stream(pos.device)), 'find_mlp_bwd')
		if defer and _lib.get_tuning('pending', pos.device):
			# (the gradients by their BASE buffers: a second reference to a gradient tensor itself would make AccumulateGrad copy it -- on
			# this stream, while the side streams are still writing it -- instead of adopting it)
			bases = {id(g._base): g._base for g in grads if g is not None and g._base is not None}
			_hold_until_join(pos.device, [scratch, ctx.ws, g_disp, g_col, pos, lat_disp, lat_col, *weights, *bases.values()])
		elif _DEFERRED:
			_join_deferred()   # (this call's own join has waited for what an earlier one left running: its buffers may go)
		if fold:
			# add into the parked gradients where the first call parked one; a weight it had nothing for gets this call's gradient as its own
			slots = pending[1]
			both = [i for i in idx if slots[i] is not None]
			if cross:
				cur.wait_event(pending[3])
				_sync_at_end_of_backward(cur)
			if both:
				# Both calls carved their gradients out of ONE flat buffer each, in the same order (_grads_like): where the tensors both
				# have lie at the same offsets of the two buffers -- the trunk's, in front of either head's -- the sum is ONE contiguous add
				# (the multi-tensor form it replaces ran 12 workgroups for 18 us at the end of the step and cost the host 35 us).  (The
				# alignment gaps between slots are added too: nobody reads them.)
				b0, g0 = slots[both[0]][0], grads[both[0]]._base
				lo, hi = slots[both[0]][1], slots[both[-1]][1] + grads[both[-1]].numel()
				shift = grads[both[0]].storage_offset() - lo   # (a data-parallel run parks the first call's gradients in the bucket's arena: another base, the same spacing)
				if (FOLD_FLAT and g0 is not None and g0.dim() == 1 and b0.dim() == 1 and all(slots[i][0] is b0 and grads[i]._base is g0 and slots[i][1] + shift == grads[i].storage_offset() for i in both)
						and all(slots[both[k + 1]][1] - (slots[both[k]][1] + ((grads[both[k]].numel() + 3) & ~3)) == 0 for k in range(len(both) - 1))):
					b0[lo:hi].add_(g0[lo + shift:hi + shift])
				else:
					torch._foreach_add_([slots[i][0][slots[i][1]:slots[i][1] + grads[i].numel()].view(grads[i].shape) for i in both], [grads[i] for i in both])
			out = [None if slots[i] is not None else grads[i] for i in range(len(weights))]
			return (None, None, g_lat_disp, g_lat_col, None, None, *out)
		_PENDING_WGRADS.clear()
		if can_park and idx and all(grads[i]._base is not None and grads[i]._base.dim() == 1 for i in idx):
			# (where the gradients live, not the tensors themselves: autograd adopts a gradient as .grad only while nothing else holds it)
			done = torch.cuda.Event()
			done.record(cur)   # (behind the join of this call's side streams: the gradients are complete where this event sits)
			_PENDING_WGRADS[key] = (task, [(g._base, g.storage_offset()) if g is not None else None for g in grads], cur, done)
		return (None, None, g_lat_disp, g_lat_col, None, None, *grads)


import os as _os
DEFER_WGRAD_JOIN = _os.environ.get('FIND_DEFER_WGRADS', '1') != '0'   # switch for A/B runs
FOLD_FLAT = _os.environ.get('FIND_FOLD_FLAT', '1') != '0'           # switch for A/B runs: the second pass's trunk gradients added with one contiguous add
_DEFERRED = {}   # device -> tensors the side streams of a deferred find_mlp_bwd may still be using


_DEFERRED_TASK = [None]   # the autograd graph task whose end-of-pass callback will join what _DEFERRED holds


def _hold_until_join(device, tensors):
	task = _graph_task_id()
	if _DEFERRED and _DEFERRED_TASK[0] != task:
		# leftovers of a pass that never reached its end-of-pass callback (an exception between the deferred call and the end of that
		# backward()): join them here -- the callback is queued PER GRAPH TASK, a pass that defers always ends with a join
		_join_deferred()
	first = not _DEFERRED
	_DEFERRED.setdefault(device, []).extend(t for t in tensors if t is not None)
	if first:
		_DEFERRED_TASK[0] = task
		# the latest moment: the end of this backward pass, on the thread and stream that run it
		torch.autograd.Variable._execution_engine.queue_callback(_join_deferred)


def _join_deferred():
	"""Make the current stream of every device with deferred weight-gradient work wait for it, then let the buffers go."""
	L = _lib.lib()
	for device in list(_DEFERRED):
		check(min(L.find_ctx_join(_lib.ctx(device), current_stream(device)), 0), 'find_ctx_join')
	_DEFERRED.clear()


_CROSS_STREAMS = []   # streams other than the backward() caller's that hold the last word on a parked gradient of this pass


def _sync_at_end_of_backward(stream):
	if not _CROSS_STREAMS:
		torch.autograd.Variable._execution_engine.queue_callback(_join_cross)   # runs on the caller's stream when the pass ends
	if stream not in _CROSS_STREAMS:
		_CROSS_STREAMS.append(stream)


def _join_cross():
	try:
		for s in _CROSS_STREAMS:
			torch.cuda.current_stream(s.device).wait_stream(s)
	finally:
		_CROSS_STREAMS.clear()


_PENDING_WGRADS = {}   # weight data_ptrs -> (autograd graph-task id, [(flat buffer, offset)] of the gradients the first backward of that task handed to
                        # the engine, the stream they were produced on)


def _graph_task_id():
	f = getattr(torch._C, '_current_graph_task_id', None)
	return int(f()) if f is not None else -1


def mlp(spec, pos, lat_disp, lat_col, B, avg_col, weights, want=('disp', 'col'), defer_wgrad_join=False):
	"""Fused Fourier-PE + trunk + heads.  pos (1|N, V, 3); lat_disp (N, Ld)|None; lat_col (N, Lc)|None;
	weights: flat list [trunk w,b ..., disp w,b ..., col w,b ...] in reference state_dict order.
	Returns disp (N,V,3), col (N,V,3)   (reference: NeuralDisplacementField.forward, model.py:393-453); a head that `want` does not
	name is not evaluated and comes back as None (its parameters then get no gradient from this call, as a head nothing reads).
	defer_wgrad_join: this call's backward may return before its weight-gradient kernels have finished (see _MLP.backward); for a small
	per-foot call whose weight gradients nothing reads before a later MLP backward or the end of the pass: FIND's texture term."""
	heads = (1 if 'disp' in want else 0) | (2 if 'col' in want else 0)
	if heads == 0:
		raise ValueError("find_amd.mlp: want must name 'disp', 'col' or both")
	return _MLP.apply((spec, heads, defer_wgrad_join), pos, lat_disp, lat_col, B, avg_col, *weights)


# ----------------------------------------------------------------------------------------------- registration
class _LatentGather(torch.autograd.Function):
	"""rows = table[idx]   (LatentVector.__getitem__, model.py:131-152); deterministic scatter backward."""

	@staticmethod
	def forward(ctx, table, idx):
		_require_gpu(table, idx)
		L = _lib.lib()
		if table.dim() != 2 or idx.dim() != 1 or idx.dtype != torch.int64:
			raise RuntimeError(f'find_amd.latent_gather: table (rows, dim) and int64 idx (n) expected, got {tuple(table.shape)} / {tuple(idx.shape)} {idx.dtype}')
		table, idx = _c(table), _c(idx)
		out = torch.empty(idx.shape[0], table.shape[1], device=table.device, dtype=torch.float32)
		check(L.find_latent_gather_fwd(ptr(table), table.shape[0], table.shape[1], ptr(idx), idx.shape[0], ptr(out), current_stream(table.device)),
			  'find_latent_gather_fwd')
		ctx.save_for_backward(idx)
		ctx.shape = tuple(table.shape)
		ctx.table_key = (table.data_ptr(), table.numel())
		return out

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		idx, = ctx.saved_tensors
		g = _c(g)
		d_table = _grad_for(ctx.table_key, ctx.shape, g.device)
		check(L.find_latent_gather_bwd(ptr(g), ptr(idx), idx.shape[0], ctx.shape[0], ctx.shape[1], ptr(d_table), current_stream(g.device)),
			  'find_latent_gather_bwd')
		return d_table, None


def latent_gather(table, idx):
	return _LatentGather.apply(table, idx)


def _ptr_array(tensors):
	"""Host array of device pointers (NULL for None)."""
	return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _LatentGatherMany(torch.autograd.Function):
	"""rows_t = table_t[idx_t] for up to 8 tables in one launch each way (find_latent_gather_many_fwd / _bwd)."""

	@staticmethod
	def forward(ctx, n, *args):
		tables, idxs = args[:n], args[n:]
		_require_gpu(*tables)
		L = _lib.lib()
		tables = [_c(t) for t in tables]
		idxs = [_c(i) for i in idxs]
		n_idx = idxs[0].shape[0]
		for t, i in zip(tables, idxs):
			if t.dim() != 2 or i.dim() != 1 or i.dtype != torch.int64 or i.shape[0] != n_idx or not i.is_cuda:
				raise RuntimeError(f'find_amd.latent_gather_many: tables (rows, dim) and device int64 idx ({n_idx}) expected, got {tuple(t.shape)} / {tuple(i.shape)} {i.dtype}')
		dev = tables[0].device
		outs = [torch.empty(n_idx, t.shape[1], device=dev, dtype=torch.float32) for t in tables]
		rows = (ctypes.c_int64 * n)(*[t.shape[0] for t in tables])
		dims = (ctypes.c_int64 * n)(*[t.shape[1] for t in tables])
		check(L.find_latent_gather_many_fwd(n, _ptr_array(tables), rows, dims, _ptr_array(idxs), n_idx, _ptr_array(outs), current_stream(dev)),
			  'find_latent_gather_many_fwd')
		ctx.save_for_backward(*idxs)
		ctx.shapes = [tuple(t.shape) for t in tables]
		ctx.keys = [(t.data_ptr(), t.numel()) for t in tables]
		return tuple(outs)

	@staticmethod
	def backward(ctx, *gs):
		L = _lib.lib()
		idxs = ctx.saved_tensors
		n = len(idxs)
		dev = idxs[0].device
		gs = [_c(g) for g in gs]
		d_tables = [_grad_for(k, sh, dev) for k, sh in zip(ctx.keys, ctx.shapes)]
		rows = (ctypes.c_int64 * n)(*[sh[0] for sh in ctx.shapes])
		dims = (ctypes.c_int64 * n)(*[sh[1] for sh in ctx.shapes])
		check(L.find_latent_gather_many_bwd(n, _ptr_array(gs), rows, dims, _ptr_array(idxs), idxs[0].shape[0], _ptr_array(d_tables), current_stream(dev)),
			  'find_latent_gather_many_bwd')
		return (None, *d_tables, *([None] * n))


def latent_gather_many(tables, idxs):
	"""[table[idx] for table, idx in zip(tables, idxs)] in one launch (and one launch for all the scatter gradients)."""
	if not 1 <= len(tables) <= 8 or len(tables) != len(idxs):
		raise ValueError('find_amd.latent_gather_many: 1 .. 8 tables, one index tensor each')
	return list(_LatentGatherMany.apply(len(tables), *tables, *idxs))


class _WeightedTerms(torch.autograd.Function):
	"""(sum_i w_i * term_i, [w_i * term_i]) for 0-dim device terms: ModelWithLoss's loss weighting (reference model.py:1157-1163)."""

	@staticmethod
	def forward(ctx, weights, *terms):
		_require_gpu(*terms)
		L = _lib.lib()
		n = len(terms)
		terms = [_c(t) for t in terms]
		dev = terms[0].device
		out = torch.empty(n + 1, device=dev, dtype=torch.float32)
		w = (ctypes.c_float * n)(*weights)
		check(L.find_weighted_terms_fwd(n, _ptr_array(terms), w, ptr(out), ctypes.c_void_p(out.data_ptr() + 4 * n), current_stream(dev)), 'find_weighted_terms_fwd')
		ctx.weights = tuple(float(x) for x in weights)
		ctx.set_materialize_grads(False)   # an unused output's gradient arrives as None, not as a zero tensor (one fill launch each)
		return out[n], out[:n]

	@staticmethod
	def backward(ctx, g_total, g_scaled):
		L = _lib.lib()
		n = len(ctx.weights)
		ref = g_total if g_total is not None else g_scaled
		if ref is None:
			return (None,) * (n + 1)
		d = torch.empty(n, device=ref.device, dtype=torch.float32)
		w = (ctypes.c_float * n)(*ctx.weights)
		check(L.find_weighted_terms_bwd(n, w, ptr(_c(g_total)), ptr(_c(g_scaled)), ptr(d), current_stream(ref.device)), 'find_weighted_terms_bwd')
		return (None, *[d[i] for i in range(n)])


def weighted_terms(terms, weights):
	"""total = sum_i weights[i] * terms[i] (in order) and the list of the weighted terms, for 0-dim device tensors: one launch."""
	if not 1 <= len(terms) <= 8 or len(terms) != len(weights):
		raise ValueError('find_amd.weighted_terms: 1 .. 8 terms, one weight each')
	total, scaled = _WeightedTerms.apply(tuple(float(w) for w in weights), *terms)
	return total, [scaled[i] for i in range(len(terms))]


class _Register(torch.autograd.Function):
	"""X = ((verts + disp) * S) @ R(euler XYZ) + t     (model.py:481-491)."""

	@staticmethod
	def forward(ctx, verts, disp, reg):
		_require_gpu(verts, disp, reg)
		if verts.requires_grad:
			raise RuntimeError('find_amd.register_points: template vertices carry no gradient in FIND (model.py:285)')
		L = _lib.lib()
		verts, disp, reg = _c(verts), _c(disp), _c(reg)
		n_feet, V, _ = disp.shape
		vb = verts.shape[0]
		if vb not in (1, n_feet) or reg.shape != (n_feet, 9):
			raise RuntimeError(f'find_amd.register_points: bad shapes verts {tuple(verts.shape)} disp {tuple(disp.shape)} reg {tuple(reg.shape)}')
		out = torch.empty_like(disp)
		check(L.find_register_fwd(ptr(verts), vb, ptr(disp), ptr(reg), n_feet, V, ptr(out), current_stream(disp.device)), 'find_register_fwd')
		ctx.save_for_backward(verts, disp, reg)
		return out

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		verts, disp, reg = ctx.saved_tensors
		n_feet, V, _ = disp.shape
		g = _c(g)
		d_disp = torch.empty_like(disp)
		d_reg = torch.empty_like(reg)
		ws = _ws(L.find_register_bwd_ws_bytes(n_feet, V), disp.device)
		check(L.find_register_bwd(ptr(verts), verts.shape[0], ptr(disp), ptr(reg), ptr(g), n_feet, V, ptr(d_disp), ptr(d_reg),
								  ptr(ws), ws.numel(), current_stream(disp.device)), 'find_register_bwd')
		return None, d_disp, d_reg


def register_points(verts, disp, reg):
	return _Register.apply(verts, disp, reg)


# ----------------------------------------------------------------------------------------------- surface sampling
def _faces_i32(faces):
	f = faces if faces.dtype == torch.int32 else faces.to(torch.int32)
	return f.contiguous()


def face_areas(verts, faces):
	"""0.5*|(v1-v0)x(v2-v0)| per face, no gradient (the sampler's multinomial weights; PyTorch3D computes them under
	no_grad too).  verts (N,V,3); faces (F,3) shared or (N,F,3) (-1 padded)."""
	_require_gpu(verts)
	L = _lib.lib()
	verts = _c(verts.detach())
	faces = _faces_i32(faces)
	N, V, _ = verts.shape
	fb = 1 if faces.dim() == 2 else faces.shape[0]
	F = faces.shape[-2]
	out = torch.empty(N, F, device=verts.device, dtype=torch.float32)
	check(L.find_face_areas(ptr(verts), ptr(faces), fb, N, V, F, ptr(out), current_stream(verts.device)), 'find_face_areas')
	return out


class _SamplePoints(torch.autograd.Function):
	@staticmethod
	def forward(ctx, verts, faces, face_idx, uv, attr):
		_require_gpu(verts, uv, attr)
		L = _lib.lib()
		verts, uv, attr = _c(verts), _c(uv), _c(attr)
		faces = _faces_i32(faces)
		face_idx = face_idx.to(torch.int32).contiguous()
		N, V, _ = verts.shape
		S = face_idx.shape[1]
		fb = 1 if faces.dim() == 2 else faces.shape[0]
		F = faces.shape[-2]
		out = torch.empty(N, S, 3, device=verts.device, dtype=torch.float32)
		aout = torch.empty_like(out) if attr is not None else None
		check(L.find_sample_points_fwd(ptr(verts), ptr(faces), fb, ptr(face_idx), ptr(uv), N, V, F, S, ptr(out), ptr(attr), ptr(aout),
									   current_stream(verts.device)), 'find_sample_points_fwd')
		ctx.save_for_backward(faces, face_idx, uv)
		ctx.dims = (N, V, F, S, fb)
		ctx.has_attr = attr is not None
		ctx.set_materialize_grads(False)
		if attr is None:
			return out
		return out, aout

	@staticmethod
	def backward(ctx, g_out, g_attr=None):
		L = _lib.lib()
		faces, face_idx, uv = ctx.saved_tensors
		N, V, F, S, fb = ctx.dims

		def scatter(g):
			if g is None:
				return None
			d = torch.zeros(N, V, 3, device=g.device, dtype=torch.float32)
			check(L.find_sample_points_bwd(ptr(faces), fb, ptr(face_idx), ptr(uv), ptr(_c(g)), N, V, F, S, ptr(d), current_stream(g.device)),
				  'find_sample_points_bwd')
			return d

		d_verts = scatter(g_out) if ctx.needs_input_grad[0] else None
		d_attr = scatter(g_attr) if (ctx.has_attr and ctx.needs_input_grad[4]) else None
		return d_verts, None, None, None, d_attr


def sample_points(verts, faces, face_idx, uv, attr=None):
	"""Gather half of sample_points_from_meshes with the draws (face_idx (N,S), uv (N,S,2)) given.
	Returns points (N,S,3) [and attr sampled with the same barycentric weights]."""
	return _SamplePoints.apply(verts, faces, face_idx, uv, attr)


class _NoCtx:
	"""Stands in for an autograd context when a Function's forward is called directly (no input wants a gradient): takes the attributes
	and calls a forward makes, keeps nothing."""
	needs_input_grad = (False,) * 64

	def save_for_backward(self, *tensors):
		pass

	def mark_non_differentiable(self, *tensors):
		pass

	def set_materialize_grads(self, value):
		pass


_NO_CTX = _NoCtx()
CACHE_AREA_SUMS = _os.environ.get('FIND_CACHE_AREA_SUMS', '1') != '0'
_AREA_SUMS = {}         # key -> (running area sums of the meshes, event behind the launch that wrote them)
_AREA_SUM_OWNERS = {}   # key -> the tensors the key names


class _SampleSurface(torch.autograd.Function):
	"""sample_points_from_meshes with the face choice on the device: rnd (N,S,3) uniform draws [face, u, v] -> points (N,S,3)
	[, attr samples], plus the chosen faces and (u,v) as non-differentiable outputs (find_sample_surface_fwd)."""

	@staticmethod
	def forward(ctx, verts, faces, rnd, attr):
		_require_gpu(verts, rnd, attr)
		L = _lib.lib()
		verts_in, faces_in = verts, faces
		verts, rnd, attr = _c(verts), _c(rnd), _c(attr)
		faces = _faces_i32(faces)
		N, V, _ = verts.shape
		S = rnd.shape[1]
		if rnd.shape != (N, S, 3):
			raise RuntimeError(f'find_amd.sample_surface: rnd must be ({N}, S, 3), got {tuple(rnd.shape)}')
		fb = 1 if faces.dim() == 2 else faces.shape[0]
		F = faces.shape[-2]
		dev = verts.device
		out = torch.empty(N, S, 3, device=dev, dtype=torch.float32)
		aout = torch.empty_like(out) if attr is not None else None
		face_idx = torch.empty(N, S, device=dev, dtype=torch.int32)
		uv = torch.empty(N, S, 2, device=dev, dtype=torch.float32)
		# A mesh that carries no gradient (a GT scan: sampled twice per training step, the same tensors step after step) keeps its running
		# area sum: keyed by the tensors' storage AND version counters (an in-place torch op on either invalidates the entry), a handful of
		# entries, ordered behind the stream that built them by an event.  (four launches per step.)
		key = None
		if CACHE_AREA_SUMS and not verts.requires_grad and not torch.cuda.is_current_stream_capturing():
			key = (verts_in.data_ptr(), verts_in._version, faces_in.data_ptr(), faces_in._version, N, V, F, fb, dev.index)
			hit = _AREA_SUMS.get(key)
			if hit is not None:
				ws, ev, made_on = hit
				cur = current_stream(dev)
				if cur.value != made_on:   # (a stream sees its own earlier work)
					torch.cuda.current_stream(dev).wait_event(ev)
				check(L.find_sample_surface_again(ptr(verts), ptr(faces), fb, ptr(rnd), N, V, F, S, ptr(face_idx), ptr(uv), ptr(out), ptr(attr), ptr(aout),
												  ptr(ws), ws.numel(), current_stream(dev)), 'find_sample_surface_again')
				key = False   # (the entry keeps `ws` alive: no record_stream)
		if key is not False:
			ws = _ws(L.find_sample_surface_ws_bytes(N, F), dev)
			check(L.find_sample_surface_fwd(ptr(verts.detach()), ptr(faces), fb, ptr(rnd), N, V, F, S, ptr(face_idx), ptr(uv), ptr(out), ptr(attr), ptr(aout),
											ptr(ws), ws.numel(), current_stream(dev)), 'find_sample_surface_fwd')
			if key is not None:
				ev = torch.cuda.Event()
				ev.record(torch.cuda.current_stream(dev))
				if len(_AREA_SUMS) >= 8:
					_AREA_SUMS.pop(next(iter(_AREA_SUMS)))
				_AREA_SUMS[key] = (ws, ev, current_stream(dev).value)
				_AREA_SUM_OWNERS[key] = (verts_in, faces_in)   # (the keyed tensors stay alive with the entry: a freed and re-used address cannot alias it)
				for k in [k for k in _AREA_SUM_OWNERS if k not in _AREA_SUMS]:
					del _AREA_SUM_OWNERS[k]
		ctx.save_for_backward(faces, face_idx, uv)
		ctx.dims = (N, V, F, S, fb)
		ctx.has_attr = attr is not None
		ctx.mark_non_differentiable(face_idx, uv)
		ctx.set_materialize_grads(False)   # (autograd would hand zero tensors for face_idx / uv to backward: two fill launches per call)
		if attr is None:
			return out, face_idx, uv
		return out, aout, face_idx, uv

	@staticmethod
	def backward(ctx, g_out, *rest):
		L = _lib.lib()
		faces, face_idx, uv = ctx.saved_tensors
		N, V, F, S, fb = ctx.dims
		g_attr = rest[0] if ctx.has_attr else None

		def scatter(g):
			if g is None:
				return None
			d = torch.zeros(N, V, 3, device=g.device, dtype=torch.float32)
			check(L.find_sample_points_bwd(ptr(faces), fb, ptr(face_idx), ptr(uv), ptr(_c(g)), N, V, F, S, ptr(d), current_stream(g.device)),
				  'find_sample_points_bwd')
			return d

		d_verts = scatter(g_out) if ctx.needs_input_grad[0] else None
		d_attr = scatter(g_attr) if (ctx.has_attr and ctx.needs_input_grad[3]) else None
		return d_verts, None, None, d_attr


def sample_surface(verts, faces, rnd, attr=None):
	"""All of sample_points_from_meshes on the device given the uniform draws rnd (N,S,3) = [face draw, u, v]: faces ~ multinomial(area).
	Returns (points (N,S,3), attr samples or None, face_idx (N,S) int32, uv (N,S,2))."""
	if torch.is_grad_enabled() and (verts.requires_grad or (attr is not None and attr.requires_grad)):
		r = _SampleSurface.apply(verts, faces, rnd, attr)
	else:
		# nothing to differentiate (a GT scan, an evaluation): the same code without the autograd Function around it (its bookkeeping is
		# 20 us of host time per call, twice per training step, on a step whose host time is what a slow host makes the step time)
		r = _SampleSurface.forward(_NO_CTX, verts, faces, rnd, attr)
	if attr is None:
		return r[0], None, r[1], r[2]
	return r


# ----------------------------------------------------------------------------------------------- Chamfer / KNN
class _NN(torch.autograd.Function):
	"""dist[n,i] = min_j |x_i - y_j|^2 (K=1 knn_points); gradient flows to both x and the selected y."""

	@staticmethod
	def forward(ctx, x, y, x_len, y_len):
		_require_gpu(x, y)
		L = _lib.lib()
		x, y = _c(x), _c(y)
		N, P1, _ = x.shape
		P2 = y.shape[1]
		if y.shape[0] != N:
			raise RuntimeError('find_amd.nn: batch mismatch')
		xl = None if x_len is None else x_len.to(device=x.device, dtype=torch.int32).contiguous()
		yl = None if y_len is None else y_len.to(device=x.device, dtype=torch.int32).contiguous()
		dist = torch.empty(N, P1, device=x.device, dtype=torch.float32)
		idx = torch.empty(N, P1, device=x.device, dtype=torch.int32)
		check(L.find_nn_fwd(ptr(x), ptr(xl), ptr(y), ptr(yl), N, P1, P2, ptr(dist), ptr(idx), current_stream(x.device)), 'find_nn_fwd')
		ctx.save_for_backward(x, y, idx, xl if xl is not None else torch.empty(0, device=x.device, dtype=torch.int32))
		ctx.has_xl = xl is not None
		ctx.mark_non_differentiable(idx)
		return dist, idx

	@staticmethod
	def backward(ctx, g_dist, _g_idx):
		L = _lib.lib()
		x, y, idx, xl = ctx.saved_tensors
		N, P1, _ = x.shape
		P2 = y.shape[1]
		d_x = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
		d_y = torch.zeros_like(y) if ctx.needs_input_grad[1] else None
		if d_x is not None or d_y is not None:
			check(L.find_nn_bwd(ptr(x), ptr(xl) if ctx.has_xl else None, ptr(y), ptr(idx), ptr(_c(g_dist)), N, P1, P2, ptr(d_x), ptr(d_y),
								current_stream(x.device)), 'find_nn_bwd')
		return d_x, d_y, None, None


def knn1(x, y, x_len=None, y_len=None):
	"""Nearest neighbour in y of every x: (squared distance (N,P1), index (N,P1) int32)."""
	return _NN.apply(x, y, x_len, y_len)


class _Chamfer(torch.autograd.Function):
	"""pytorch3d.loss.chamfer_distance (defaults) as one scalar: both nearest-neighbour directions in one launch + a deterministic
	reduction (find_chamfer_fwd); the backward is one kernel over both directions (find_chamfer_bwd)."""

	@staticmethod
	def forward(ctx, x, y, x_len, y_len):
		_require_gpu(x, y)
		L = _lib.lib()
		x, y = _c(x), _c(y)
		N, P1, _ = x.shape
		P2 = y.shape[1]
		if y.shape[0] != N:
			raise RuntimeError('find_amd.chamfer_distance: batch mismatch')
		dev = x.device
		xl = None if x_len is None else x_len.to(device=dev, dtype=torch.int32).contiguous()
		yl = None if y_len is None else y_len.to(device=dev, dtype=torch.int32).contiguous()
		ws = _ws(L.find_chamfer_ws_bytes(N, P1, P2), dev)
		loss = torch.empty((), device=dev, dtype=torch.float32)
		check(L.find_chamfer_fwd(ptr(x), ptr(xl), ptr(y), ptr(yl), N, P1, P2, ptr(loss), ptr(ws), ws.numel(), current_stream(dev)), 'find_chamfer_fwd')
		ctx.save_for_backward(x, y, ws)
		ctx.lens = (xl, yl)
		return loss

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		x, y, ws = ctx.saved_tensors
		xl, yl = ctx.lens
		N, P1, _ = x.shape
		P2 = y.shape[1]
		need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
		if not (need_x or need_y):
			return None, None, None, None
		buf = torch.zeros((N * P1 * 3 if need_x else 0) + (N * P2 * 3 if need_y else 0), device=x.device, dtype=torch.float32)
		d_x = buf[:N * P1 * 3].view(N, P1, 3) if need_x else None
		d_y = buf[buf.numel() - N * P2 * 3:].view(N, P2, 3) if need_y else None
		check(L.find_chamfer_bwd(ptr(x), ptr(xl), ptr(y), ptr(yl), N, P1, P2, ptr(_c(g)), ptr(ws), ws.numel(), ptr(d_x), ptr(d_y), current_stream(x.device)),
			  'find_chamfer_bwd')
		return d_x, d_y, None, None


def chamfer_distance(x, y, x_lengths=None, y_lengths=None):
	"""pytorch3d.loss.chamfer_distance with its defaults (bidirectional, squared L2, point mean, batch mean).
	Returns (loss, None) like PyTorch3D (the second value is the normals term, never used by FIND: losses.py:77,85,88)."""
	return _Chamfer.apply(x, y, x_lengths, y_lengths), None


class _MaskedMSE(torch.autograd.Function):
	@staticmethod
	def forward(ctx, pred, target):
		_require_gpu(pred, target)
		L = _lib.lib()
		pred, target = _c(pred), _c(target.detach())
		if pred.shape != target.shape or pred.shape[-1] != 3:
			raise RuntimeError(f'find_amd.masked_mse: pred {tuple(pred.shape)} and target {tuple(target.shape)} must both be (..., 3)')
		loss = torch.empty((), device=pred.device, dtype=torch.float32)
		check(L.find_masked_mse_fwd(ptr(pred), ptr(target), pred.numel() // 3, ptr(loss), current_stream(pred.device)), 'find_masked_mse_fwd')
		ctx.save_for_backward(pred, target)
		return loss

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		pred, target = ctx.saved_tensors
		d = torch.empty_like(pred)
		check(L.find_masked_mse_bwd(ptr(pred), ptr(target), pred.numel() // 3, ptr(_c(g)), ptr(d), current_stream(pred.device)), 'find_masked_mse_bwd')
		return d, None


def masked_mse(pred, target):
	"""mean over all elements of any(target < 1, -1) * (pred - target)^2: the texture loss's masked L2 (reference losses.py:43-57)."""
	return _MaskedMSE.apply(pred, target)


# ----------------------------------------------------------------------------------------------- smoothness
class _ImageMSE(torch.autograd.Function):
	"""mean((a * am - b * bm)^2): the pixel loss on images inside their silhouettes / the silhouette loss on the masks themselves
	(find_image_mse_fwd / _bwd); gradients to a and am only (b, bm are the GT render)."""

	@staticmethod
	def forward(ctx, a, am, b, bm):
		_require_gpu(a, am, b, bm)
		L = _lib.lib()
		a, am, b, bm = _c(a), _c(am), _c(b.detach()), _c(None if bm is None else bm.detach())
		if a.shape != b.shape:
			raise RuntimeError(f'find_amd.image_mse: shapes differ: {tuple(a.shape)} / {tuple(b.shape)}')
		C = 1 if am is None and bm is None else int(a.shape[-1])
		n_pix = a.numel() // C
		for m in (am, bm):
			if m is not None and m.numel() != n_pix:
				raise RuntimeError(f'find_amd.image_mse: a mask of {m.numel()} values for {n_pix} pixels')
		loss = torch.empty((), device=a.device, dtype=torch.float32)
		ws = _ws(L.find_image_mse_ws_bytes(), a.device)
		check(L.find_image_mse_fwd(ptr(a), ptr(am), ptr(b), ptr(bm), n_pix, C, ptr(loss), ptr(ws), ws.numel(), current_stream(a.device)), 'find_image_mse_fwd')
		ctx.save_for_backward(a, am, b, bm)
		ctx.dims = (n_pix, C)
		return loss

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		a, am, b, bm = ctx.saved_tensors
		n_pix, C = ctx.dims
		d_a = torch.empty_like(a) if ctx.needs_input_grad[0] else None
		d_am = torch.empty_like(am) if (am is not None and ctx.needs_input_grad[1]) else None
		if d_a is None and d_am is None:
			return None, None, None, None
		check(L.find_image_mse_bwd(ptr(a), ptr(am), ptr(b), ptr(bm), n_pix, C, ptr(_c(g)), ptr(d_a), ptr(d_am), current_stream(a.device)), 'find_image_mse_bwd')
		return d_a, d_am, None, None


def image_mse(a, b, a_mask=None, b_mask=None):
	"""F.mse_loss(a * a_mask[..., None], b * b_mask[..., None]) (masks optional) in one pass each way; b / b_mask carry no gradient."""
	return _ImageMSE.apply(a, a_mask, b, b_mask)


class MeshTopology:
	"""Static per-template tables for the smoothness kernels: unique undirected edges, vertex->incident-corner CSR and
	vertex->neighbour CSR (host-built once, cached per faces tensor)."""
	_cache = {}

	def __init__(self, faces, n_verts):
		import numpy as np
		f = faces.detach().cpu().numpy().astype(np.int64).reshape(-1, 3)
		self.n_verts, self.n_faces = int(n_verts), int(f.shape[0])
		e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
		e.sort(axis=1)
		e = np.unique(e, axis=0)
		self.n_edges = int(e.shape[0])
		# vertex -> corners
		vid = f.reshape(-1)
		order = np.argsort(vid, kind='stable')
		vf_items = order.astype(np.int32)  # item = face*3 + corner
		vf_off = np.zeros(self.n_verts + 1, np.int32)
		np.add.at(vf_off, vid + 1, 1)
		vf_off = np.cumsum(vf_off).astype(np.int32)
		# vertex -> neighbours
		a = np.concatenate([e[:, 0], e[:, 1]])
		b = np.concatenate([e[:, 1], e[:, 0]])
		o2 = np.argsort(a, kind='stable')
		nbr_idx = b[o2].astype(np.int32)
		nbr_off = np.zeros(self.n_verts + 1, np.int32)
		np.add.at(nbr_off, a + 1, 1)
		nbr_off = np.cumsum(nbr_off).astype(np.int32)
		dev = faces.device
		self.faces = torch.from_numpy(f.astype(np.int32)).to(dev)
		self.edges = torch.from_numpy(e.astype(np.int32)).to(dev)
		self.vf_off = torch.from_numpy(vf_off).to(dev)
		self.vf_items = torch.from_numpy(vf_items).to(dev)
		self.nbr_off = torch.from_numpy(nbr_off).to(dev)
		self.nbr_idx = torch.from_numpy(nbr_idx).to(dev)

	@classmethod
	def get(cls, faces, n_verts):
		"""Tables of a faces tensor, built once per tensor OBJECT: the entry keeps a weak reference to the tensor it was built from, so a
		freed tensor whose address (or id) is reused by different connectivity never matches, and an in-place edit (_version) rebuilds."""
		import weakref
		key = (id(faces), int(n_verts))
		ent = cls._cache.get(key)
		if ent is not None:
			ref, version, topo = ent
			if ref() is faces and version == int(faces._version):
				return topo
		if len(cls._cache) > 16:
			cls._cache.clear()
		topo = cls(faces, n_verts)
		cls._cache[key] = (weakref.ref(faces), int(faces._version), topo)
		return topo


class _Smooth(torch.autograd.Function):
	@staticmethod
	def forward(ctx, verts, topo):
		_require_gpu(verts)
		L = _lib.lib()
		verts = _c(verts)
		N, V, _ = verts.shape
		if V != topo.n_verts:
			raise RuntimeError(f'find_amd.smooth: verts have {V} vertices, topology {topo.n_verts}')
		ws = _ws(L.find_smooth_ws_bytes(N, V, topo.n_faces), verts.device)
		out = torch.empty(2, device=verts.device, dtype=torch.float32)
		check(L.find_smooth_fwd(ptr(verts), ptr(topo.faces), ptr(topo.vf_off), ptr(topo.vf_items), ptr(topo.nbr_off), ptr(topo.nbr_idx),
								N, V, topo.n_faces, topo.n_edges, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 4),
								ptr(ws), ws.numel(), current_stream(verts.device)), 'find_smooth_fwd')
		ctx.topo, ctx.ws = topo, ws
		ctx.save_for_backward(verts)
		return out[0], out[1]

	@staticmethod
	def backward(ctx, g_edge, g_lap):
		L = _lib.lib()
		(verts,) = ctx.saved_tensors
		topo = ctx.topo
		N, V, _ = verts.shape
		g = torch.zeros(2, device=verts.device, dtype=torch.float32)
		if g_edge is not None:
			g[0] = g_edge
		if g_lap is not None:
			g[1] = g_lap
		d = torch.empty_like(verts)
		check(L.find_smooth_bwd(ptr(verts), ptr(topo.faces), ptr(topo.vf_off), ptr(topo.vf_items), ptr(topo.nbr_off), ptr(topo.nbr_idx),
								N, V, topo.n_faces, topo.n_edges, ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(g.data_ptr() + 4),
								ptr(ctx.ws), ctx.ws.numel(), ptr(d), current_stream(verts.device)), 'find_smooth_bwd')
		return d, None


def mesh_edge_and_laplacian(verts, topo):
	"""(mesh_edge_loss(target 0), mesh_laplacian_smoothing('cot')) for a batch sharing one topology."""
	return _Smooth.apply(verts, topo)


class _SmoothLoss(torch.autograd.Function):
	@staticmethod
	def forward(ctx, verts, topo, w_edge, w_lap):
		_require_gpu(verts)
		L = _lib.lib()
		verts = _c(verts)
		N, V, _ = verts.shape
		if V != topo.n_verts:
			raise RuntimeError(f'find_amd.smooth: verts have {V} vertices, topology {topo.n_verts}')
		ws = _ws(L.find_smooth_ws_bytes(N, V, topo.n_faces), verts.device)
		out = torch.empty((), device=verts.device, dtype=torch.float32)
		check(L.find_smooth_loss_fwd(ptr(verts), ptr(topo.faces), ptr(topo.vf_off), ptr(topo.vf_items), ptr(topo.nbr_off), ptr(topo.nbr_idx),
									 N, V, topo.n_faces, topo.n_edges, w_edge, w_lap, ptr(out), ptr(ws), ws.numel(), current_stream(verts.device)),
			  'find_smooth_loss_fwd')
		ctx.topo, ctx.ws, ctx.w = topo, ws, (w_edge, w_lap)
		ctx.save_for_backward(verts)
		return out

	@staticmethod
	def backward(ctx, g):
		L = _lib.lib()
		(verts,) = ctx.saved_tensors
		topo = ctx.topo
		N, V, _ = verts.shape
		d = torch.empty_like(verts)
		check(L.find_smooth_loss_bwd(ptr(verts), ptr(topo.faces), ptr(topo.vf_off), ptr(topo.vf_items), ptr(topo.nbr_off), ptr(topo.nbr_idx),
									 N, V, topo.n_faces, topo.n_edges, ctx.w[0], ctx.w[1], ptr(_c(g)), ptr(ctx.ws), ctx.ws.numel(), ptr(d),
									 current_stream(verts.device)), 'find_smooth_loss_bwd')
		return d, None, None, None


def mesh_smoothness_loss(verts, topo, w_edge=10.0, w_lap=0.1):
	"""w_edge * mesh_edge_loss + w_lap * mesh_laplacian_smoothing('cot') as one scalar (MeshSmoothnessLoss, reference losses.py:93-99)."""
	return _SmoothLoss.apply(verts, topo, float(w_edge), float(w_lap))


# Default arithmetic of the 256 -> 256 layers: 'bf16x3' (fp32-faithful on the bf16 matrix pipe; set_mlp_precision below says what that is);
# FIND_MLP_PRECISION=fp32 selects the fp32 MFMA kernels (A/B runs, the bench's comparison record)
_MLP_PRECISION = _os.environ.get('FIND_MLP_PRECISION', 'bf16x3')
if _MLP_PRECISION not in ('fp32', 'fp16', 'bf16x3'):
	raise ValueError(f"FIND_MLP_PRECISION: 'fp32', 'bf16x3' or 'fp16', got {_MLP_PRECISION!r}")
# the isolated-kernel entry points (find_linear_*) read the CONTEXT's knob: every context is created with the process default (the
# knob used to stay at fp32 until someone called set_mlp_precision, so get_mlp_precision() and those entry points could disagree)
_lib._TUNING_DEFAULTS.setdefault('mlp_f16', {'fp32': 0, 'fp16': 1, 'bf16x3': 2}[_MLP_PRECISION])


def set_mlp_precision(precision):
	"""Default arithmetic of the MLP's 256 -> 256 layers (forward, the dX chain and the weight gradients) for every model whose MLPSpec
	names no precision of its own (model.set_mlp_precision / MLPSpec.precision); also the mode of the isolated-kernel entry points.
	'fp32': fp32 MFMA (v_mfma_f32_32x32x2_f32) -- the reference's arithmetic (no AMP anywhere in FIND).
	'bf16x3': fp32-FAITHFUL arithmetic on the bf16 matrix pipe (csrc/mlp_gemm6.h, mlp_dw6.h): every fp32 operand is split EXACTLY into three
	bf16 pieces and the six products of relative size >= 2^-18 are accumulated in fp32; what is left out is <= 2^-26 of a product, a quarter
	of one fp32 rounding, so results are as close to the float64 product as the fp32 MFMA's (tests/test_gpu_mlp_bf16x3.py) at 6/16 of its
	matrix-pipe time.  Tensors stay fp32; covers the large launches of the 256 -> 256 layers (forward, dX, weight gradients).
	'fp16': both MFMA operands rounded to fp16, fp32 accumulation and fp32 tensors in memory (BASELINE.json configs[4], "fp16 MLP
	with MFMA tiles"); layer outputs then differ from fp32 by ~1e-3 relative.  Covers the forward Linear layers, the dX chain and the
	weight gradients of the 256 -> 256 layers (gemm5_kernel, dw3_kernel) where a launch has at least 1024 32-row units -- smaller
	launches, e.g. the shared trunk's template rows, are faster on the fp32 kernels and stay there; the Fourier layer, the 3-wide output layers, the two-segment
	trunk-output gradient, bias / latent gradients and everything outside the MLP stay fp32.  Operands are rounded, not scaled:
	activations or gradients beyond fp16's range (|x| > 65504) become inf and magnitudes under 6e-8 vanish -- FIND's activations are
	O(1) and its dZ O(1e-6 .. 1e-1), but a loss scaled far outside that is the caller's responsibility.  The precision travels with
	each call (find_mlp_params.precision): a backward always runs in the arithmetic of its forward.  Returns the previous setting."""
	global _MLP_PRECISION
	if precision not in ('fp32', 'fp16', 'bf16x3'):
		raise ValueError(f"set_mlp_precision: 'fp32', 'bf16x3' or 'fp16', got {precision!r}")
	_lib.set_tuning('mlp_f16', {'fp32': 0, 'fp16': 1, 'bf16x3': 2}[precision])   # the isolated-kernel entry points (find_linear_*) read the context's knob
	prev, _MLP_PRECISION = _MLP_PRECISION, precision
	return prev


def get_mlp_precision():
	return _MLP_PRECISION
