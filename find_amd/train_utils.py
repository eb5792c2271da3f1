"""Caller-side helpers the reference keeps in src/train/trainer.py, mirrored so a step can be driven without the
reference tree: latent-row sampling (trainer.py:29-46) and batch_to_device (trainer.py:19-26).  Host Python only."""


def batch_to_device(batch, device='cuda'):
	return {k: (v.to(device) if hasattr(v, 'to') else v) for k, v in batch.items()}


def sample_latent_vectors(batch, latent_vectors):
	"""Rows of every LatentVector for this batch, keyed by the vector's name: by label (vec.key in batch) when the table has
	labels, else by batch['idx']."""
	if latent_vectors is None:
		return {}
	out = {}
	for vec in latent_vectors:
		if vec.labels is not None:
			assert vec.key in batch, f'Trying to sample from latent vector {vec.key} using keys, but not found in dataset'
			out[vec.name] = vec[batch[vec.key]]
		else:
			out[vec.name] = vec[batch['idx']]
	return out
