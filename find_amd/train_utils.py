"""Caller-side helpers the reference keeps in src/train/trainer.py, mirrored so a step can be driven without the
reference tree: latent-row sampling (trainer.py:29-46) and batch_to_device (trainer.py:19-26).  Host Python only."""


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
