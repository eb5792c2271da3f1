"""The training-step loop around the hot path: host-side mirror of the two loops the reference keeps in src/train/trainer.py
(Trainer.train_epoch :88-143, Trainer.val_epoch :145-176) and of the keyword dictionary src/train/train.py builds for them
(train_network :52-70).  Same constructor keywords, same per-step order -- sample the batch's latent rows, zero_grad, model(batch, epoch,
opts=opts, **model_kwargs), backward, step -- same log layout (`self.log[epoch]['train_loss' | 'val_loss']`), same return values.

Why it exists here (SURVEY §8 f4 / "the step after the path"): at the reference's shipped batch size (1 scan per step) the GPU needs
~1.1 ms for a train_3d.yaml step and the host ~1.4-1.8 ms to enqueue it, and the reference reads `loss.item()` and every term's
`.item()` after each step (trainer.py:125-127), draining the queue every time.  This loop
  * replays the step as ONE HIP graph (find_amd.graph.GraphedStep) whenever the step is capture-safe -- find_amd optimisers, no PNG to
    write this epoch, no per-scan supervision switch, step-per-batch -- and runs it eagerly otherwise (the checkpoint epochs that ask for
    save_renders, train.py:58-66);
  * keeps the loss bookkeeping on the device (one row per step in a log buffer) and reads it back ONCE per epoch.
Progress bars, matplotlib plots and OBJ export (trainer.py:178-300) are visualisation / control plane: not provided."""
import contextlib
from collections import defaultdict

import numpy as np
import torch

from . import functional_render as FR
from .train_utils import backward, backward_on_this_thread, batch_to_device, sample_latent_vectors


def pretty_print_loss(loss_key: str):
	"""loss_chamf -> Chamf (trainer.py:14-16)."""
	return loss_key.replace('loss_', '').replace('_', ' ').title()


def stage_model_kwargs(args, epoch, num_epochs, save_every, render_dir=None):
	"""The keyword dictionary train_network hands ModelWithLoss.forward at `epoch` of a network / latent stage (train.py:52-70):
	(model_kwargs, save_model).  render_foot is on when a render loss is on OR this epoch saves a checkpoint and rendering is not
	switched off; save_renders on checkpoint epochs (0, every save_every-th, the last)."""
	import os
	is_last = epoch == num_epochs - 1
	save_model = epoch % save_every == 0 or is_last
	save_renders = save_model and not args.no_rendering
	kw = args.net_train_kwargs()
	kw['render_foot'] = any(k in args.render_losses and v for k, v in kw.items()) or save_renders
	kw['save_renders'] = save_renders
	kw['render_dir'] = os.path.join(render_dir if render_dir is not None else args.render_dir, 'train')
	kw['restyle_feature_maps'] = args.restyle_feature_maps
	kw['no_displacement'] = args.only_classifier_head
	kw['copy_mask_out'] = args.copy_over_masking
	kw['mask_out_pred_faces'] = args.mask_out_pred
	kw['gt_z_cutoff'] = args.gt_z_cutoff
	return kw, save_model


class _DeviceLog:
	"""Loss values of an epoch, kept on the device: row i = [total, term_0, term_1, ...] of step i; read back once."""

	def __init__(self):
		self.keys, self.rows = None, []

	def add(self, loss, loss_dict):
		keys = tuple(loss_dict)
		if self.keys is None:
			self.keys = keys
		vals = [loss] + [loss_dict[k] for k in keys]
		# (a fresh tensor per step: the graph's outputs are static buffers the next replay overwrites)
		self.rows.append((keys, torch.stack([v.detach().reshape(()) for v in vals])))

	def read(self):
		"""{'Loss': [...], key: [...]} as Python floats: ONE device-to-host copy per distinct term set."""
		out = defaultdict(list)
		by_keys = defaultdict(list)
		for keys, row in self.rows:
			by_keys[keys].append(row)
		for keys, rows in by_keys.items():
			vals = torch.stack(rows).cpu().tolist()
			for r in vals:
				out['Loss'].append(r[0])
				for k, v in zip(keys, r[1:]):
					out[k].append(v)
		return out


class Trainer:
	def __init__(self, optims, model, train_loader, val_loader, opts, latent_vectors_train: list = None, latent_vectors_val: list = None,
				 val_optim=None, device='cuda', n_repeat=1, graph='auto'):
		"""optims: optimiser or list of optimisers stepped after every training batch; model: ModelWithLoss; loaders: iterables of collated
		batches with a length; latent_vectors_*: the LatentVector lists sampled into each batch; val_optim: the optimiser of val_epoch
		(the reference's "validation" is itself an optimisation step on the val latents, trainer.py:160-164).
		graph: 'auto' (HIP-graph replay when the step is capture-safe, else eager), True (raise when it is not), False (always eager)."""
		self.opts = opts
		self.optims = optims if isinstance(optims, list) else [optims]
		self.model = model
		self.train_loader, self.val_loader = train_loader, val_loader
		self.latent_vectors_train, self.latent_vectors_val = latent_vectors_train, latent_vectors_val
		self.val_optim = val_optim
		self.device = device
		self.log = defaultdict(dict)   # per-epoch losses
		self.best_epochs = []
		self.n_repeat = n_repeat
		self.graph = graph
		self._graphed = {}
		self.last_mode = None   # 'graph' | 'eager': how the most recent epoch ran (tests, bench)
		self.last_captures = 0  # HIP graphs captured during the most recent epoch (ragged scans share bucketed graphs: a handful per dataset)
		# Every step of this trainer -- eager, warm-up, capture -- runs on ONE stream of its own: autograd ties a parameter's gradient
		# accumulation to the stream its node was created on, and a capture that meets a node of another stream (kept alive by any loss
		# or latent row of an eager step) does not survive hipStreamEndCapture (find_amd.graph.GraphedStep: stream).
		self._stream = None
		# what a render that meets an unclipped face or an overfull pixel does inside this Trainer's epochs (functional_render.FLAG_POLICY:
		# 'strict' everywhere else): warn and go on -- a training run must not end on a close-up visualisation render
		self.render_flag_policy = 'warn'

	def sample_latent_vectors(self, batch, latent_vectors=None):
		return sample_latent_vectors(batch, self.latent_vectors_train if latent_vectors is None else latent_vectors)

	# ------------------------------------------------------------------ graph or eager?
	def _why_not_graph(self, optims, model_kwargs):
		from . import optim
		if self.graph is False:
			return 'graph=False'
		o = self.opts
		if model_kwargs.get('save_renders'):
			return 'save_renders writes a PNG from the host'
		if getattr(o, 'step_per_epoch', False):
			return 'opts.step_per_epoch accumulates gradients over the epoch'
		if getattr(o, 'restrict_3d_n_train', None) is not None or getattr(o, 'restrict_3d_train_key', None) is not None:
			return 'per-scan 3-D supervision switches are read on the host'
		if not any(model_kwargs.get(k) for k in ('chamf', 'smooth', 'texture', 'pix', 'sil')):
			return 'no loss term enabled'
		for op in optims:
			if not isinstance(op, (optim.Adam, optim.SGD)):
				return f'{type(op).__name__} is not capture-safe (use find_amd.optim)'
			if isinstance(op, optim.Adam) and not all(g.get('capturable', False) for g in op.param_groups):
				return 'find_amd.optim.Adam needs capturable=True'
		if not str(self.device).startswith('cuda'):
			return 'not a GPU device'
		return None

	def _step_stream(self):
		if not str(self.device).startswith('cuda'):
			return None
		if self._stream is None:
			self._stream = torch.cuda.Stream(device=self.device)
		self._stream.wait_stream(torch.cuda.current_stream(self.device))   # whatever produced the loader's tensors
		return self._stream

	def _captures(self):
		return sum(g.n_captures for g in self._graphed.values())

	def _epoch_done(self):
		if self._stream is not None:
			torch.cuda.current_stream(self.device).wait_stream(self._stream)
		# the render watchdog's counters of this epoch (a warning per bad render: Trainer.render_flag_policy; outside a Trainer a bad
		# render is an error, functional_render.FLAG_POLICY): the epoch's losses were read on the host just before, so they have all arrived
		with FR.flag_policy(self.render_flag_policy):
			FR.check_render_flags(wait=True)

	def _graphed_step(self, optims, latent_vectors, model_kwargs):
		from .graph import GraphedStep
		key = (tuple(id(o) for o in optims), tuple(sorted((k, v if isinstance(v, (bool, int, float, str, type(None))) else id(v)) for k, v in model_kwargs.items())))
		gs = self._graphed.get(key)
		if gs is None:
			flags = {k: v for k, v in model_kwargs.items() if k != 'render_dir'}
			gs = self._graphed[key] = GraphedStep(self.model, self.opts, optims, latent_vectors=latent_vectors, warmup=1, stream=self._step_stream(), **flags)
		return gs

	def _mode(self, optims, latent_vectors, model_kwargs):
		why = self._why_not_graph(optims, model_kwargs)
		if why is None:
			return self._graphed_step(optims, latent_vectors, model_kwargs)
		if self.graph is True:
			raise RuntimeError(f'find_amd.trainer: graph=True but this step cannot be captured: {why}')
		return None

	# ------------------------------------------------------------------ the two loops
	def train_epoch(self, epoch, save_model=False, model_kwargs={}):
		model_kwargs['is_train'] = True
		log = _DeviceLog()
		gs = self._mode(self.optims, self.latent_vectors_train, model_kwargs)
		self.last_mode = 'graph' if gs is not None else 'eager'
		captures0 = self._captures()
		step_per_epoch = bool(getattr(self.opts, 'step_per_epoch', False))
		stream = self._step_stream()
		n_steps = 0
		with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()), backward_on_this_thread(), FR.flag_policy(self.render_flag_policy):
			[o.zero_grad() for o in self.optims]
			for _ in range(self.n_repeat):
				for batch in self.train_loader:
					if gs is not None:
						# the latent rows are sampled inside the captured step (row indices travel through a pinned staging buffer)
						loss, loss_dict = gs(batch_to_device(batch, self.device), epoch)
					else:
						# (a copy: the reference updates the DataLoader's fresh dict in place; a loader that hands out the SAME dicts every epoch
						# would otherwise keep the sampled rows -- and with them the step's autograd graph -- alive)
						batch = dict(batch)
						batch.update(**self.sample_latent_vectors(batch, latent_vectors=self.latent_vectors_train))
						batch = batch_to_device(batch, self.device)
						if not step_per_epoch:
							[o.zero_grad() for o in self.optims]
						loss, loss_dict = self.model(batch, epoch, opts=self.opts, **model_kwargs)
						# (`if loss == 0: continue`, trainer.py:108: a step without any term -- sum({}.values()) -- is skipped.  The reference's
						# test also reads a tensor loss back to compare it with 0; a step whose terms are all exactly 0.0 is not skipped here.)
						if not torch.is_tensor(loss):
							continue
						backward(loss)
						if not step_per_epoch:
							[o.step() for o in self.optims]
					log.add(loss, loss_dict)
					n_steps += 1
			if step_per_epoch:
				[o.step() for o in self.optims]
		self._epoch_done()
		vals = log.read()
		epoch_losses = {('Loss' if k == 'Loss' else pretty_print_loss(k)): v for k, v in vals.items()}
		self.log[epoch]['train_loss'] = dict(epoch_losses)
		self.last_captures = self._captures() - captures0
		how = self.last_mode + (f', {self.last_captures} graph(s) captured' if self.last_captures else '')
		return ('*' * save_model) + f'[{epoch}] ' + '|'.join(f'{k}:{np.mean(v):.2f}' for k, v in epoch_losses.items()) + f' ({n_steps} steps, {how})'

	def val_epoch(self, epoch, model_kwargs={}):
		model_kwargs['is_train'] = False
		if len(self.val_loader) == 0:
			return '', {}
		log = _DeviceLog()
		gs = self._mode([self.val_optim], self.latent_vectors_val, model_kwargs)
		self.last_mode = 'graph' if gs is not None else 'eager'
		captures0 = self._captures()
		stream = self._step_stream()
		with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()), backward_on_this_thread(), FR.flag_policy(self.render_flag_policy):
			for batch in self.val_loader:
				if gs is not None:
					loss, loss_dict = gs(batch_to_device(batch, self.device), epoch)
				else:
					batch = dict(batch)
					batch.update(**self.sample_latent_vectors(batch, latent_vectors=self.latent_vectors_val))
					batch = batch_to_device(batch, self.device)
					self.val_optim.zero_grad()
					loss, loss_dict = self.model(batch, epoch, opts=self.opts, **model_kwargs)
					backward(loss)
					self.val_optim.step()
				log.add(loss, loss_dict)
		self._epoch_done()
		vals = log.read()
		epoch_losses = {('Loss' if k == 'Loss' else k.replace('loss_', '').lower()): v for k, v in vals.items()}
		self.log[epoch]['val_loss'] = {pretty_print_loss(k): v for k, v in epoch_losses.items()}
		self.last_captures = self._captures() - captures0
		msg = f'[{epoch} - VAL] ' + '|'.join(f'{pretty_print_loss(k)}:{np.mean(v):.2f}' for k, v in epoch_losses.items())
		return msg, {k: np.mean(v) for k, v in epoch_losses.items()}

	def plot(self, out_loc):
		raise NotImplementedError('loss plots (matplotlib, trainer.py:178-222) are visualisation: out of scope; the numbers are in Trainer.log')

	def export_meshes(self, export_loc, is_train=False, export_gt=False):
		raise NotImplementedError('OBJ export through trimesh (trainer.py:259-300) is out of scope; model.get_meshes_from_batch returns the meshes')
