"""A whole FIND training step as ONE HIP graph.

At the reference's own batch size (batch_size_train = 1, src/train/opts.py:40) a train_3d.yaml step is a few hundred small
kernels -- sampling, two MLP passes, Chamfer, smoothness, their backward passes, the fused optimiser step -- and the host needs
longer to enqueue them than the GPU needs to run them.  GraphedStep captures the step once per batch shape (torch.cuda.CUDAGraph
on top of hipGraph; the side-stream fork / joins inside libfind_hip.so are captured with it) and afterwards replays it: the host's
work per step is copying the batch into the graph's static input buffers and one hipGraphLaunch.

What a step is (src/train/trainer.py:97-123): sample the batch's latent rows, ModelWithLoss.forward(batch, epoch, opts, **flags),
loss.backward(), step every optimiser.  The label -> row lookup of label-addressed tables (model.py:137-149) stays on the host, as in
the reference; only the resulting row indices travel to the device.

Constraints (checked or documented): every batch of one shape signature shares a graph; optimisers must be capture-safe
(find_amd.optim.Adam(capturable=True), find_amd.optim.SGD); flags that make ModelWithLoss.forward read device values on the host
(opts.restrict_3d_n_train) cannot be captured.  The warm-up iterations that precede a capture are DRY: parameters and optimiser
state are put back before the capture, so N calls are N training steps, the first batch of a shape included (only the random
generators -- the sampler's device generator, numpy's for the camera poses -- have advanced)."""
import collections

import torch

from . import functional as FN

from .structures import Meshes, TexturesUV, TexturesVertex


def _mesh_tensors(m):
	"""The device tensors a Meshes object carries, in a fixed order."""
	out = [m._verts]
	out.append(m._faces_shared if m._faces_shared is not None else m._faces)
	t = m.textures
	if isinstance(t, TexturesVertex):
		out.append(t._feat)
	elif isinstance(t, TexturesUV):
		out += [t._maps, t._faces_uvs, t._verts_uvs]
	elif t is not None:
		raise NotImplementedError(f'GraphedStep: unsupported texture type {type(t).__name__}')
	return out


def _mesh_padding(m):
	"""What the ragged dimension of each tensor of _mesh_tensors is padded with (PyTorch3D's conventions: vertices / features 0, face lists
	-1 -- every kernel of the path skips a face whose first index is negative); None = not ragged (texture maps keep their size)."""
	out = [0.0, -1]
	t = m.textures
	if isinstance(t, TexturesVertex):
		out.append(0.0)
	elif isinstance(t, TexturesUV):
		out += [None, -1, 0.0]
	return out


def _ragged_dim(t):
	return 0 if t.dim() == 2 else 1   # a face list shared by the batch is (F, 3); everything else (N, count, ...)


def bucket_size(n):
	"""Smallest of 256, 384, 512, 768, 1024, 1536, ... (half-octave steps: at most a third of a bucket is padding) that holds n."""
	b = 256
	while True:
		if b >= n:
			return b
		if b * 3 // 2 >= n:
			return b * 3 // 2
		b *= 2


def _bucket_shape(t, pad):
	if pad is None:
		return tuple(t.shape)
	sh = list(t.shape)
	d = _ragged_dim(t)
	sh[d] = bucket_size(sh[d])
	return tuple(sh)


def _padded_mesh(m):
	"""A copy of `m` whose vertex / face / feature tensors have bucket sizes (the static input of a captured step)."""
	new = []
	for t, pad in zip(_mesh_tensors(m), _mesh_padding(m)):
		sh = _bucket_shape(t, pad)
		if sh == tuple(t.shape):
			new.append(t.clone())
			continue
		p = t.new_full(sh, pad)
		p.narrow(_ragged_dim(t), 0, t.shape[_ragged_dim(t)]).copy_(t)
		new.append(p)
	tex = None
	if isinstance(m.textures, TexturesVertex):
		tex = TexturesVertex(new[2])
	elif isinstance(m.textures, TexturesUV):
		tex = TexturesUV(new[2], new[3], new[4])
	return Meshes(new[0], new[1], tex)


class _Captured:
	pass


class GraphedStep:
	def __init__(self, model_with_loss, opts, optimizers, latent_vectors=None, warmup=2, pre_step=None, stream=None, bucket=True, max_graphs=8, **flags):
		"""model_with_loss: find_amd.model_with_loss.ModelWithLoss; optimizers: list stepped after backward; latent_vectors: the
		LatentVector list to sample per batch (default: model.latent_vectors_train, or _val when flags has is_train=False);
		pre_step: optional callable run between backward and the optimiser steps (e.g. a gradient all-reduce); flags: the loss
		switches handed to ModelWithLoss.forward.
		bucket: scans of different sizes share a graph -- real Foot3D scans are ragged (src/data/dataset.py:263-297: "10-20 K faces"), and keyed on
		exact shapes every scan of a batch-1 epoch would be its own warm-up + capture + static buffers.  The batch's meshes are padded to
		bucket sizes (bucket_size: half-octave steps; faces with -1, which every kernel of the path skips, vertices and features with 0: no
		face refers to them), so a whole dataset replays a handful of graphs and gives the numbers of the unpadded eager step (a padded face
		has area 0 and is never sampled; the samplers' draws do not depend on the sizes).  max_graphs: captured graphs kept (least
		recently used goes first); n_captures counts them (find_amd.trainer.Trainer reports it per epoch).
		stream: the (non-default) stream warm-up and capture run on.  autograd ties every parameter's AccumulateGrad node to the stream it
		was created on and keeps the node while ANY graph that reaches the parameter is alive; a backward captured on another stream
		then records a dependency on that foreign stream, which hipStreamEndCapture (ROCm 7.x) answers with a segmentation fault.  A
		caller that also runs eager steps on these parameters passes the stream it runs them on (find_amd.trainer.Trainer does);
		without one, a capture refuses to start while such a graph is alive (a loss, or sampled latent rows, kept from an eager step)."""
		from . import optim
		self.mwl, self.opts, self.flags = model_with_loss, opts, dict(flags)
		self.optimizers = list(optimizers)
		for o in self.optimizers:
			if isinstance(o, optim.Adam) and not all(g.get('capturable', False) for g in o.param_groups):
				raise RuntimeError('GraphedStep: find_amd.optim.Adam must be built with capturable=True (its step count has to live on the device)')
			if not isinstance(o, (optim.Adam, optim.SGD)):
				raise RuntimeError(f'GraphedStep: {type(o).__name__} is not known to be capture-safe; use find_amd.optim.Adam(capturable=True) / SGD')
		if getattr(opts, 'restrict_3d_n_train', None) is not None:
			raise RuntimeError('GraphedStep: opts.restrict_3d_n_train makes the step read batch["idx"] on the host; not capturable')
		if self.flags.get('save_renders'):
			raise RuntimeError('GraphedStep: save_renders reads the images and batch["idx"] on the host; run that step eagerly '
							   '(find_amd.trainer.Trainer does: the reference asks for it only on checkpoint epochs, train.py:58-66)')
		# Camera poses.  ModelWithLoss draws them on the host (numpy's global generator, model.py:1060-1071) -- inside a capture that would
		# freeze the first step's poses into every replay.  They are a static INPUT of the graph instead: drawn per call in _load(), as the
		# eager step draws them, or the caller's own device tensors (refreshed from the same objects every call).
		self._draw_views = bool(self.flags.get('render_foot')) and self.flags.get('views') is None
		if self.flags.get('views') is not None and not all(torch.is_tensor(t) and t.is_cuda for t in self.flags['views']):
			raise RuntimeError('GraphedStep: views=(R, T) must be device tensors (a host tensor would be copied from pageable memory inside the capture)')
		m = model_with_loss.model
		if latent_vectors is None:
			latent_vectors = m.latent_vectors_train if self.flags.get('is_train', True) else m.latent_vectors_val
		self.latent_vectors = list(latent_vectors or [])
		if warmup < 1:
			# lazily built state must exist before a capture: topology tables (host-built), kernel attributes, side streams, optimiser
			# state -- and SGD's first step (momentum buffer := gradient) must not be the one that gets replayed
			raise ValueError('GraphedStep: at least one eager warm-up step is needed before a capture')
		self.warmup = warmup
		self.pre_step = pre_step
		self._graphs = collections.OrderedDict()
		self._pool = None
		self.bucket, self.max_graphs, self.n_captures = bool(bucket), int(max_graphs), 0
		self.stream, self._own_stream = stream, stream is None

	# ------------------------------------------------------------------ batch -> static buffers
	def _signature(self, batch):
		sig = []
		for k in sorted(batch):
			v = batch[k]
			if torch.is_tensor(v):
				sig.append((k, tuple(v.shape), str(v.dtype)))
			elif isinstance(v, Meshes):
				if self.bucket:   # bucket sizes, not the scan's own: the counts travel as -1 padding
					sig.append((k, tuple((_bucket_shape(t, pad), str(t.dtype)) for t, pad in zip(_mesh_tensors(v), _mesh_padding(v))), len(v),
								v._faces_shared is not None, type(v.textures).__name__))
				else:
					sig.append((k, tuple((tuple(t.shape), str(t.dtype)) for t in _mesh_tensors(v)), tuple(v._num_verts), tuple(v._num_faces),
								type(v.textures).__name__))
			elif isinstance(v, (list, tuple)):
				sig.append((k, len(v)))
			else:
				sig.append((k, type(v).__name__))
		return tuple(sig)

	def _latent_indices(self, batch):
		"""Host side of trainer.sample_latent_vectors: one row index per batch item and table (list.index for labelled tables)."""
		rows = []
		for vec in self.latent_vectors:
			if vec.labels is not None:
				assert vec.key in batch, f'Trying to sample from latent vector {vec.key} using keys, but not found in dataset'
				rows.append([vec._index_of(o) for o in batch[vec.key]])
			else:
				rows.append(None)  # addressed by batch['idx'], already a device tensor
		return rows

	def _capture(self, batch, epoch):
		dev = self.mwl.model.template_verts.device
		st = _Captured()
		# static copies of everything the step reads from the batch
		st.batch = {}
		for k, v in batch.items():
			if torch.is_tensor(v) and not v.is_cuda:
				raise RuntimeError(f'GraphedStep: batch[{k!r}] is a CPU tensor; move the batch to the device first (trainer.batch_to_device): '
								   'a host tensor would be read from pageable memory inside the capture and never refreshed')
			if torch.is_tensor(v) and v.is_cuda:
				st.batch[k] = v.clone()
			elif isinstance(v, Meshes):
				st.batch[k] = _padded_mesh(v) if self.bucket else v.clone()
			else:
				st.batch[k] = v
		st.filled = {}   # (batch key, tensor index) -> how far the static tensor holds real data (what a smaller scan must overwrite with padding)
		n_lab = sum(1 for vec in self.latent_vectors if vec.labels is not None)
		bsz = len(batch['idx']) if 'idx' in batch else len(next(iter(batch.values())))
		st.idx_host = torch.zeros(max(n_lab, 1), bsz, dtype=torch.int64).pin_memory()
		st.idx_dev = torch.zeros(max(n_lab, 1), bsz, dtype=torch.int64, device=dev)
		flags = dict(self.flags)
		st.views_host = st.views_dev = None
		if self._draw_views:
			R, T = self.mwl._views(self.opts)
			st.views_host = torch.cat([R.reshape(-1, 9), T.reshape(-1, 3)], dim=1).float().pin_memory()   # (M, 12)
			st.views_dev = st.views_host.to(dev)
			flags['views'] = (st.views_dev[:, :9].reshape(-1, 3, 3), st.views_dev[:, 9:])
			st.views_fresh = True   # the poses just drawn serve the first step
		# hyper-parameters the optimiser kernels receive BY VALUE are frozen into the graph: remember them, refuse a replay after a change
		st.hyper = self._hyper()

		def body():
			b = dict(st.batch)
			j = 0
			sel = []
			for vec in self.latent_vectors:
				if vec.labels is not None:
					sel.append(st.idx_dev[j])
					j += 1
				else:
					sel.append(b['idx'])
			idx = [vec.device_index(s) for vec, s in zip(self.latent_vectors, sel)]
			if len(idx) > 1 and all(i is not None for i in idx):   # every table in one launch, as train_utils.sample_latent_vectors
				rows = FN.latent_gather_many([vec.data for vec in self.latent_vectors], idx)
			else:
				rows = [vec[s] for vec, s in zip(self.latent_vectors, sel)]
			for vec, r in zip(self.latent_vectors, rows):
				b[vec.name] = r
			out = self.mwl(b, epoch, self.opts, **flags)
			loss, losses = out[0], out[1]
			loss.backward()
			if self.pre_step is not None:
				self.pre_step()
			for o in self.optimizers:
				o.step()
			return loss, losses

		self._load(st, batch)
		if st.views_host is not None:
			st.views_fresh = True   # (that _load consumed the flag without drawing: the captured poses serve the first REPLAY, so N calls make N draws -- ADVICE r3)
		snap = self._snapshot()
		if self.stream is None:
			self.stream = torch.cuda.Stream(device=dev)
		if self._own_stream:
			self._refuse_live_graphs()
		side = self.stream
		side.wait_stream(torch.cuda.current_stream(dev))
		with torch.cuda.stream(side):
			for _ in range(self.warmup):
				for o in self.optimizers:
					o.zero_grad(set_to_none=True)
				body()
		torch.cuda.current_stream(dev).wait_stream(side)
		self._restore(snap)
		torch.cuda.synchronize(dev)
		for o in self.optimizers:
			o.zero_grad(set_to_none=True)
		if self._pool is None:
			self._pool = torch.cuda.graph_pool_handle()
		st.graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(st.graph, pool=self._pool, stream=self.stream):
			st.loss, st.losses = body()
		return st

	def _refuse_live_graphs(self):
		"""Raise if an autograd graph that reaches one of the optimised parameters is still alive (see __init__: stream).  A parameter's
		AccumulateGrad node exists only while such a graph does: fetch it twice, tagging it the first time -- a node that is still
		tagged on the second fetch outlived our own reference, so something else holds it."""
		import gc
		def alive():
			out = []
			for o in self.optimizers:
				for g in o.param_groups:
					for p in g['params']:
						if not p.requires_grad:
							continue
						p.view_as(p).grad_fn.next_functions[0][0].metadata['find_probe'] = True
						if p.view_as(p).grad_fn.next_functions[0][0].metadata.pop('find_probe', False):
							out.append(tuple(p.shape))
			return out
		if alive():
			gc.collect()
			shapes = alive()
			if shapes:
				raise RuntimeError(f'GraphedStep: an autograd graph from an earlier step still references {len(shapes)} of the optimised parameters (shapes '
								   f'{shapes[:4]} ...): something keeps a loss, an output or sampled latent rows of that step.  Drop those references, '
								   'or run the eager steps and this capture on one stream (GraphedStep(stream=...)); capturing now would crash in hipStreamEndCapture.')

	def _snapshot(self):
		"""Values of every optimised parameter and of its optimiser state, to undo the warm-up steps."""
		snap = []
		for o in self.optimizers:
			for g in o.param_groups:
				for p in g['params']:
					st = o.state.get(p, {})
					snap.append((o, g, p, p.detach().clone(), {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}))
		return snap

	@torch.no_grad()
	def _restore(self, snap):
		"""In place (the capture that follows records these tensors' addresses).  State the warm-up created is reset to its initial value:
		Adam's moments and step count to zero; SGD's momentum buffer to zero, which makes the next step torch's "first step" (buffer :=
		gradient) exactly when dampening is 0 -- FIND's setting (train.py:162) -- and is refused otherwise."""
		for o, g, p, value, state in snap:
			p.copy_(value)
			for k, v in o.state.get(p, {}).items():
				if not torch.is_tensor(v):
					continue
				if k in state:
					v.copy_(state[k])
				else:
					if k == 'momentum_buffer' and g.get('dampening', 0) != 0:
						raise RuntimeError('GraphedStep: SGD with dampening != 0 cannot take a dry warm-up step (its first step differs from the later ones)')
					v.zero_()
		for o in self.optimizers:   # host mirrors of the step counts (find_amd.optim.Adam keeps one per bucket beside the device counter)
			for c in getattr(o, '_cache', {}).values():
				for bk in c.get('buckets', []):
					if 'step_dev' in bk:
						bk['step'] = int(bk['step_dev'].item())

	def _hyper(self):
		keys = ('lr', 'betas', 'eps', 'weight_decay', 'momentum', 'dampening', 'nesterov')
		return [tuple((k, g[k]) for k in keys if k in g) for o in self.optimizers for g in o.param_groups]

	def _load(self, st, batch):
		if getattr(st, 'hyper', None) is not None and st.hyper != self._hyper():
			raise RuntimeError('GraphedStep: an optimiser hyper-parameter (lr, betas, ...) changed after the capture; they are launch arguments '
							   'frozen into the graph -- build a new GraphedStep (FIND trains with constant rates, train.py:161-168)')
		if st.views_host is not None:
			if st.views_fresh:
				st.views_fresh = False
			else:
				R, T = self.mwl._views(self.opts)
				if getattr(st, 'views_event', None) is not None:
					st.views_event.synchronize()
				st.views_host.copy_(torch.cat([R.reshape(-1, 9), T.reshape(-1, 3)], dim=1))
				st.views_dev.copy_(st.views_host, non_blocking=True)
				st.views_event = torch.cuda.Event()
				st.views_event.record()
		for k, v in batch.items():
			s = st.batch[k]
			if torch.is_tensor(v) and not v.is_cuda:
				raise RuntimeError(f'GraphedStep: batch[{k!r}] is a CPU tensor; move the batch to the device first')
			if torch.is_tensor(v) and v.is_cuda:
				if s.data_ptr() != v.data_ptr():
					s.copy_(v, non_blocking=True)
			elif isinstance(v, Meshes):
				src, dst = _mesh_tensors(v), _mesh_tensors(s)
				same = [a.shape == b.shape for a, b in zip(src, dst)]
				if all(same):
					if any(a.data_ptr() != b.data_ptr() for a, b in zip(src, dst)):
						torch._foreach_copy_(dst, src)
					# (a scan that fills its bucket exactly: the static tensors are full NOW -- ADVICE r4: with `filled` left at the previous,
					# smaller scan's size the next mid-sized scan skipped the tail fill and kept this scan's last faces)
					for i, b in enumerate(dst):
						if (k, i) in st.filled:
							st.filled[(k, i)] = b.shape[_ragged_dim(b)]
					continue
				for i, (a, b, pad) in enumerate(zip(src, dst, _mesh_padding(v))):
					if a.shape == b.shape:
						b.copy_(a, non_blocking=True)
						if (k, i) in st.filled:
							st.filled[(k, i)] = b.shape[_ragged_dim(b)]
						continue
					d = _ragged_dim(a)
					n, was = a.shape[d], st.filled.get((k, i), b.shape[d])
					b.narrow(d, 0, n).copy_(a, non_blocking=True)
					if was > n:   # the previous, larger scan's tail
						b.narrow(d, n, was - n).fill_(pad)
					st.filled[(k, i)] = n
		rows = [r for r in self._latent_indices(batch) if r is not None]
		if rows:
			new = torch.tensor(rows, dtype=torch.int64)
			if not torch.equal(new, st.idx_host):
				# (the pinned staging buffer may still be read by the previous step's copy: wait for that one first)
				if getattr(st, 'idx_event', None) is not None:
					st.idx_event.synchronize()
				st.idx_host.copy_(new)
				st.idx_dev.copy_(st.idx_host, non_blocking=True)
				st.idx_event = torch.cuda.Event()
				st.idx_event.record()

	def __call__(self, batch, epoch=0):
		"""Run one step on `batch`.  Returns (loss, losses): static tensors, valid once the caller's stream has run the replay.
		The batch is loaded and the graph launched on the step's own stream, never on the legacy default stream: there a graph launched
		right behind the copy of a new scan into the static buffers did not wait for it (ROCm 7.0: a replay every ~0.65 ms with the
		network frozen ended in a GPU memory fault within a few dozen steps; with a synchronisation between copy and launch, or on any
		other stream, never) -- the caller's stream waits for the step's stream and the other way round, which costs two event waits."""
		sig = self._signature(batch)
		st = self._graphs.get(sig)
		if st is None:
			st = self._graphs[sig] = self._capture(batch, epoch)
			self.n_captures += 1
			while len(self._graphs) > self.max_graphs:   # least recently used first: its graph and static buffers go back to the pool
				self._graphs.popitem(last=False)
		else:
			self._graphs.move_to_end(sig)
		dev = st.idx_dev.device
		cur = torch.cuda.current_stream(dev)
		if cur == self.stream:
			self._load(st, batch)
			st.graph.replay()
			return st.loss, st.losses
		self.stream.wait_stream(cur)
		with torch.cuda.stream(self.stream):
			self._load(st, batch)
			st.graph.replay()
		cur.wait_stream(self.stream)
		return st.loss, st.losses
