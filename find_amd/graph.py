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
(opts.restrict_3d_n_train) cannot be captured.  The warm-up iterations that precede a capture are real training steps."""
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


class _Captured:
	pass


class GraphedStep:
	def __init__(self, model_with_loss, opts, optimizers, latent_vectors=None, warmup=2, pre_step=None, **flags):
		"""model_with_loss: find_amd.model_with_loss.ModelWithLoss; optimizers: list stepped after backward; latent_vectors: the
		LatentVector list to sample per batch (default: model.latent_vectors_train, or _val when flags has is_train=False);
		pre_step: optional callable run between backward and the optimiser steps (e.g. a gradient all-reduce); flags: the loss
		switches handed to ModelWithLoss.forward."""
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
		self._graphs = {}
		self._pool = None

	# ------------------------------------------------------------------ batch -> static buffers
	@staticmethod
	def _signature(batch):
		sig = []
		for k in sorted(batch):
			v = batch[k]
			if torch.is_tensor(v):
				sig.append((k, tuple(v.shape), str(v.dtype)))
			elif isinstance(v, Meshes):
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
		st.pairs = []  # (static tensor, key path) to refresh per call
		for k, v in batch.items():
			if torch.is_tensor(v) and v.is_cuda:
				st.batch[k] = v.clone()
			elif isinstance(v, Meshes):
				st.batch[k] = v.clone()
			else:
				st.batch[k] = v
		n_lab = sum(1 for vec in self.latent_vectors if vec.labels is not None)
		bsz = len(batch['idx']) if 'idx' in batch else len(next(iter(batch.values())))
		st.idx_host = torch.zeros(max(n_lab, 1), bsz, dtype=torch.int64).pin_memory()
		st.idx_dev = torch.zeros(max(n_lab, 1), bsz, dtype=torch.int64, device=dev)

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
			out = self.mwl(b, epoch, self.opts, **self.flags)
			loss, losses = out[0], out[1]
			loss.backward()
			if self.pre_step is not None:
				self.pre_step()
			for o in self.optimizers:
				o.step()
			return loss, losses

		self._load(st, batch)
		side = torch.cuda.Stream(device=dev)
		side.wait_stream(torch.cuda.current_stream(dev))
		with torch.cuda.stream(side):
			for _ in range(self.warmup):
				for o in self.optimizers:
					o.zero_grad(set_to_none=True)
				body()
		torch.cuda.current_stream(dev).wait_stream(side)
		torch.cuda.synchronize(dev)
		for o in self.optimizers:
			o.zero_grad(set_to_none=True)
		if self._pool is None:
			self._pool = torch.cuda.graph_pool_handle()
		st.graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(st.graph, pool=self._pool):
			st.loss, st.losses = body()
		return st

	def _load(self, st, batch):
		for k, v in batch.items():
			s = st.batch[k]
			if torch.is_tensor(v) and v.is_cuda:
				if s.data_ptr() != v.data_ptr():
					s.copy_(v, non_blocking=True)
			elif isinstance(v, Meshes):
				src, dst = _mesh_tensors(v), _mesh_tensors(s)
				if any(a.data_ptr() != b.data_ptr() for a, b in zip(src, dst)):
					torch._foreach_copy_(dst, src)
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
		"""Run one step on `batch`.  Returns (loss, losses): static tensors, valid once the stream has run the replay."""
		sig = self._signature(batch)
		st = self._graphs.get(sig)
		if st is None:
			st = self._graphs[sig] = self._capture(batch, epoch)
		self._load(st, batch)
		st.graph.replay()
		return st.loss, st.losses
