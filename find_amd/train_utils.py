"""Caller-side helpers the reference keeps in src/train/trainer.py, mirrored so a step can be driven without the
reference tree: latent-row sampling (trainer.py:29-46) and batch_to_device (trainer.py:19-26).  Host Python only."""
import contextlib
import os


def backward_on_this_thread():
	"""Context for a training loop: `loss.backward()` runs its nodes on the CALLING thread instead of handing them to autograd's per-device
	worker thread.  The results are the same; what changes is the host's time.  A FIND step is ~100 short launches and its backward is a
	dozen Python autograd functions: on the worker thread every one of them pays a hand-over of the interpreter lock between two threads,
	and the pass starts with a wake-up during which the GPU drains (tools/host_ab.py, headline step of 16 feet: host time to enqueue a step
	2.0 - 2.2 ms with the worker thread, 1.35 - 1.55 ms without -- the eager loop goes from host-bound to GPU-bound).  find_amd.trainer.Trainer
	and bench.py run their steps inside it; a maintainer who keeps the reference's own loop wraps it the same way.
	FIND_AUTOGRAD_THREADS=1 keeps torch's default."""
	import torch
	if os.environ.get('FIND_AUTOGRAD_THREADS', '0') == '1':
		return contextlib.nullcontext()
	return torch.autograd.set_multithreading_enabled(False)


_ONES = {}


def backward(loss):
	"""loss.backward() with the seed gradient taken from a per-device cache: autograd otherwise makes torch.ones_like(loss) per call -- one more
	launch at the head of the backward pass, between kernels that wait for each other (Trainer, GraphedStep and bench.py call this)."""
	one = _ONES.get((loss.device, loss.dtype))
	if one is None:
		import torch
		one = _ONES[(loss.device, loss.dtype)] = torch.ones((), device=loss.device, dtype=loss.dtype)
	loss.backward(gradient=one if loss.dim() == 0 else None)


def batch_to_device(batch, device='cuda'):
	return {k: (v.to(device) if hasattr(v, 'to') else v) for k, v in batch.items()}


def sample_latent_vectors(batch, latent_vectors):
	"""Rows of every LatentVector for this batch, keyed by the vector's name: by label (vec.key in batch) when the table has
	labels, else by batch['idx'].  Tables on the GPU that are addressed by an index vector or a list of labels are looked up in ONE
	launch (find_latent_gather_many_fwd; one more for all their scatter gradients) instead of one per table."""
	if latent_vectors is None:
		return {}
	out, many = {}, []
	for vec in latent_vectors:
		if vec.labels is not None:
			assert vec.key in batch, f'Trying to sample from latent vector {vec.key} using keys, but not found in dataset'
			sel = batch[vec.key]
		else:
			sel = batch['idx']
		idx = vec.device_index(sel) if hasattr(vec, 'device_index') else None
		if idx is not None:
			many.append((vec, idx))
		else:
			out[vec.name] = vec[sel]
	if len(many) == 1:
		out[many[0][0].name] = many[0][0][many[0][1]]
	elif many:
		from . import functional as FN
		for (vec, _), rows in zip(many, FN.latent_gather_many([v.data for v, _ in many], [i for _, i in many])):
			out[vec.name] = rows
	return {vec.name: out[vec.name] for vec in latent_vectors}
