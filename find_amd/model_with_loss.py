"""ModelWithLoss on the MI355X hot path (reference src/model/model.py:969-1171): model + renderer + loss objects behind
forward(batch, epoch, opts, **flags) -> (loss, losses[, renders]).

Structure here: a registry of loss terms.  Each term names the forward() flag that enables it, the key it reports under, the
`opts.weight_*` attribute that scales it (model.py:1157-1158: every raw loss is multiplied by its weight, the total is their sum)
and whether it needs the 3-D supervision gate or the renders.  forward() evaluates the predicted meshes once, renders once if any
enabled term needs images, and walks the registry.  All arithmetic runs in libfind_hip.so through the loss / renderer objects."""
import contextlib
import os
from collections import namedtuple

import torch

from . import functional as FN
from . import losses as _losses
from .losses import DisplacementLoss, MeshSmoothnessLoss, SilhouetteLoss, TextureLossGTSpace
from .model import NeuralDisplacementField
from .renderer import FootRenderer

nn = torch.nn

model_zoo = dict(neural=NeuralDisplacementField)

OUT_OF_SCOPE_FLAGS = {
	'vgg_perc': 'perceptual / restyle / contrastive losses are out of scope (SURVEY.md §2 #4)',
	'restyle_perc_lat': 'perceptual / restyle / contrastive losses are out of scope (SURVEY.md §2 #4)',
	'restyle_perc_feat': 'perceptual / restyle / contrastive losses are out of scope (SURVEY.md §2 #4)',
	'restyle_perc_cluster': 'perceptual / restyle / contrastive losses are out of scope (SURVEY.md §2 #4)',
	'cont_pose': 'perceptual / restyle / contrastive losses are out of scope (SURVEY.md §2 #4)',
	'mask_out_pred_faces': 'mask_out_pred_faces belongs to the VertexFeatures model (out of scope)',
}

# flag -> reported key, weight attribute, needs the 3-D gate, needs renders, method computing the raw loss
Term = namedtuple('Term', 'flag key weight needs_3d needs_render fn')
TERMS = (
	Term('chamf', 'loss_chamf', 'weight_chamf', True, False, '_raw_chamf'),
	Term('smooth', 'loss_smooth', 'weight_smooth', True, False, '_raw_smooth'),
	Term('texture', 'loss_tex', 'weight_tex', True, False, '_raw_texture'),
	Term('pix', 'loss_pix', 'weight_pix', False, True, '_raw_pix'),
	Term('sil', 'loss_sil', 'weight_sil', False, True, '_raw_sil'),
)


# the GT render on a second stream beside the predicted one (ModelWithLoss._render_gt); FIND_OVERLAP_GT_RENDER=0 turns it off
OVERLAP_GT_RENDER = os.environ.get('FIND_OVERLAP_GT_RENDER', '1') != '0'
# the Chamfer term on that second stream beside the texture term's MLP pass (ModelWithLoss.forward); FIND_OVERLAP_CHAMFER=0 turns it off
OVERLAP_CHAMFER = os.environ.get('FIND_OVERLAP_CHAMFER', '1') != '0'
# the texture term (GT surface samples -> colour field -> masked MSE: a chain of its own, it reads nothing of the main pass) on the second
# stream from the START of the step, beside the main pass; FIND_TEXTURE_STREAM=0 turns it off
# = 2: the texture term on a THIRD stream, forked behind the main pass's forward -- its forward beside the Chamfer term's (as on the main stream),
# but its BACKWARD (autograd replays a node on its forward's stream) beside the main pass's backward instead of in front of it: the texture
# pass's dX chain and weight gradients (0.37 ms of the step when they sit between the loss glue and the main pass's backward) fill the small-kernel
# stretches of the main chain.  Measured in round 6 and NOT the default: 1.93-2.05 ms against 1.81-1.83 on the same box (tools/ab_env.sh) -- kernels
# that run beside each other on this chip slow each other by what the overlap gains; only less matrix-pipe time helps
TEXTURE_STREAM = int(os.environ.get('FIND_TEXTURE_STREAM', '0'))
# the GT scans' surface samples drawn before the main pass, beside it (ModelWithLoss.forward); FIND_PRESAMPLE_GT=0: inside the loss terms, as the reference orders them
PRESAMPLE_GT = os.environ.get('FIND_PRESAMPLE_GT', '1') != '0'
TEXTURE_FIRST = os.environ.get('FIND_TEXTURE_FIRST', '0') != '0'     # experiment: the texture term issued before the Chamfer term (ModelWithLoss.forward); measured SLOWER (1.69 against 1.63 ms)
TEX_BWD_LATE = os.environ.get('FIND_TEX_BWD_LATE', '0') != '0'       # experiment: the texture term's backward behind the loss-side chain's (ModelWithLoss.forward); no gain measured (1.70-1.73 against 1.68-1.70 ms)
LAZY_COLOURS = os.environ.get('FIND_LAZY_COLOURS', '1') != '0'       # switch for A/B runs and for the bench record with the reference's eager colour head
_SECOND_STREAMS = {}


def _second_stream(device):
	"""A stream whose launches really run beside the current stream's -- and wait behind as little as possible.  HIP multiplexes streams
	onto a few hardware queues (four by default; raising GPU_MAX_HW_QUEUES makes every step here SLOWER, bench.py 3.25 -> 4.6 ms: the
	sharing is part of how the MLP's side streams are laid out), and two streams on one queue run in order -- which queue a new stream
	lands on depends on what else the process has created.  The MLP context's three side streams hold the other three queues, so this
	stream shares one of them: find_ctx_stream_beside picks, among a dozen candidates, one on the queue of the context's Q stream (the
	large head layers' weight gradients: idle until the main pass's backward).  On T1's queue -- where the first candidate that "ran
	beside the caller" used to land -- the Chamfer backward waited ~0.2 ms behind the texture pass's weight gradients: headline step
	2.05 -> 1.98 ms.  FIND_SECOND_STREAM_ROLE = 0..3 (Q, T1, T2, R) for experiments, -1 = the old rule (first stream beside the caller)."""
	s = _SECOND_STREAMS.get(device)
	if s is not None:
		return s
	import ctypes
	from . import _lib
	main = torch.cuda.current_stream(device)
	cands = [torch.cuda.Stream(device=device) for _ in range(12)]
	role = int(os.environ.get('FIND_SECOND_STREAM_ROLE', '0'))
	index = ctypes.c_int32(-1)
	arr = (ctypes.c_void_p * len(cands))(*[c.cuda_stream for c in cands])
	with torch.cuda.device(device):
		torch.cuda.synchronize(device)
		if role >= 0:
			_lib.check(_lib.lib().find_ctx_stream_beside(_lib.ctx(), ctypes.c_void_p(main.cuda_stream), arr, len(cands), role, ctypes.byref(index)),
					   'find_ctx_stream_beside')
		if index.value < 0:   # (no candidate on that queue, or the old rule asked for: any stream that runs beside the caller's)
			for r in (1, 2, 3, 0):
				_lib.check(_lib.lib().find_ctx_stream_beside(_lib.ctx(), ctypes.c_void_p(main.cuda_stream), arr, len(cands), r, ctypes.byref(index)),
						   'find_ctx_stream_beside')
				if index.value >= 0:
					break
		torch.cuda.synchronize(device)
	pick = cands[max(0, index.value)]
	_SECOND_STREAMS[device] = pick
	return pick


_THIRD_STREAMS = {}


def _third_stream(device):
	"""A stream beside the caller's AND beside _second_stream: on the hardware queue of the context's T2 stream (idle in the forward; in the
	backward it carries the Fourier layer's weight gradient, which the texture pass's own chain forks and therefore never waits behind)."""
	s = _THIRD_STREAMS.get(device)
	if s is not None:
		return s
	import ctypes
	from . import _lib
	second = _second_stream(device)
	main = torch.cuda.current_stream(device)
	cands = [torch.cuda.Stream(device=device) for _ in range(12)]
	arr = (ctypes.c_void_p * len(cands))(*[c.cuda_stream for c in cands])
	index = ctypes.c_int32(-1)
	with torch.cuda.device(device):
		torch.cuda.synchronize(device)
		for role in (int(os.environ.get('FIND_THIRD_STREAM_ROLE', '2')), 1, 2):
			_lib.check(_lib.lib().find_ctx_stream_beside(_lib.ctx(), ctypes.c_void_p(main.cuda_stream), arr, len(cands), role, ctypes.byref(index)), 'find_ctx_stream_beside')
			if index.value >= 0:
				break
		torch.cuda.synchronize(device)
	pick = cands[max(0, index.value)]
	if pick is second:
		pick = cands[(max(0, index.value) + 1) % len(cands)]
	_THIRD_STREAMS[device] = pick
	return pick


def model_class_from_opts(opts):
	kind = getattr(opts, 'model_type', 'neural')
	if kind not in model_zoo:
		raise NotImplementedError(f"model_type '{kind}': only the neural displacement field is on the hot path (PCA / SUPR / "
								  'vertex-feature baselines are out of scope, SURVEY.md §2 #7-9)')
	return model_zoo[kind]


def model_from_opts(opts):
	return model_class_from_opts(opts).load(opts.load_model, device=opts.device, opts=opts)


class _Step:
	"""What one forward() call has at hand while the registry is walked."""
	__slots__ = ('batch', 'epoch', 'opts', 'res', 'is_train', 'use_z_cutoff', 'gt_z_cutoff', 'pred', 'gt', 'gt_chamf', 'gt_tex')


class ModelWithLoss(nn.Module):
	def __init__(self, *args, opts=None, device='cuda', **kwargs):
		super().__init__()
		cls = model_class_from_opts(opts)
		if opts.load_model:
			self.model = cls.load(opts.load_model, device=device, **kwargs, opts=opts)
		else:
			self.model = cls(*args, **kwargs, device=device, opts=opts)
		if opts.vgg_perc_loss or opts.use_restyle():
			raise NotImplementedError('VGG / Restyle perceptual losses need network weights that are not available; out of scope')
		self.device = device
		self.def_loss = DisplacementLoss()
		self.col_loss = TextureLossGTSpace()
		self.mesh_smooth_loss = MeshSmoothnessLoss()
		self.templ_smooth_loss = MeshSmoothnessLoss()
		self.pix_loss = nn.MSELoss()      # (attributes of the reference; the forward computes both losses through find_image_mse_*)
		self.sil_loss = SilhouetteLoss()
		# (max_faces_per_bin only sizes PyTorch3D's coarse bins -- 30 000 for the full-resolution scans, model.py:987; no effect on results)
		self.rdr = FootRenderer(image_size=256, device=device, bin_size=None, max_faces_per_bin=None if opts.low_poly_meshes else 30000)

	# ------------------------------------------------------------------ raw losses, one per registry entry
	def _raw_chamf(self, st):
		return self.def_loss(self.model, st.res, st.batch, st.epoch, z_cutoff=0.07 if st.use_z_cutoff else None, gt_z_cutoff=st.gt_z_cutoff,
							 **(dict(gt_samples=st.gt_chamf) if st.gt_chamf is not None else {}))['loss']

	def _raw_smooth(self, st):
		return self.mesh_smooth_loss(st.res['meshes'])

	def _raw_texture(self, st):
		sfx = 'train' if st.is_train else 'val'
		codes = {k: st.batch.get(f'{k}_{sfx}', None) for k in ('shapevec', 'texvec', 'posevec')}
		return self.col_loss(self.model, st.batch, **codes, **(dict(gt_samples=st.gt_tex) if st.gt_tex is not None else {}))

	def _raw_pix(self, st):
		# images are compared inside the silhouettes only (model.py:1101-1105): MSE(image * mask, gt image * gt mask), one pass each way
		# (find_image_mse_*; CPU tensors raise there: no fallback)
		return FN.image_mse(st.pred['image'], st.gt['image'], st.pred['mask'], st.gt['mask'])

	def _raw_sil(self, st):
		return FN.image_mse(st.pred['mask'], st.gt['mask'])

	# ------------------------------------------------------------------ pieces of a step
	def _views(self, opts):
		"""Camera poses shared by the GT and the predicted render of one step (model.py:1060-1071)."""
		free = dict(dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90)
		kind = opts.special_view_type
		if kind == 'topdown':
			return self.rdr.view_from('topdown')
		if kind == 'topdown_5':
			return self.rdr.combine_views(*self.rdr.view_from('topdown'), *self.rdr.sample_views(nviews=5, azim_min=-90, azim_max=90, seed=5, **free))
		if kind == 'sample_arc':
			return self.rdr.sample_views(nviews=opts.num_views, azim_min=0, azim_max=0, **free)
		return self.rdr.sample_views(nviews=opts.num_views, azim_min=-90, azim_max=90, **free)

	@staticmethod
	def _supervise_3d(batch, opts, is_train):
		"""The reference's switches that withhold 3-D supervision from some scans (model.py:1019-1030)."""
		limit = getattr(opts, 'restrict_3d_n_train', None)
		if is_train and limit is not None and batch['idx'].item() >= limit:
			return False
		if getattr(opts, 'restrict_3d_train_key', None) is not None and batch['name'][0] not in opts.train_3d_on_only:
			return False
		return True

	def _render_gt(self, st, views, masked_faces, images=True):
		"""The GT render of a step: (gt, R, T, second stream or None).  It carries no gradient and reads nothing of the prediction, so it
		runs on a second stream beside the predicted render: C3 step 8.3 -> 7.8 ms (the two rasteriser launches share the chip
		5 % better than they use it alone, and the predicted render's small launches run under the GT render's tail).  (Issued before the
		MLP forward instead, it takes CUs from the forward's GEMMs for longer than it saves: 8.3 ms.)  Not under stream capture."""
		R, T = views if views is not None else self._views(st.opts)
		dev = torch.device(st.batch['mesh'].device)
		if dev.type == 'cuda' and not R.is_cuda:
			# the camera poses go to the device once per step, through pinned memory: a pageable copy per render call (four of them) made
			# the host wait for the queue to drain each time
			R, T = (t.float().pin_memory().to(dev, non_blocking=True) for t in (R, T))
		side = None
		if OVERLAP_GT_RENDER and dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
			side = _second_stream(dev)
			side.wait_stream(torch.cuda.current_stream(dev))
		with torch.no_grad(), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
			# the GT scans are rendered again every step, as the reference does (model.py:1073-1075)
			gt = self.rdr(st.batch['mesh'], R, T, return_images=images, return_mask=True, mask_with_grad=True, mask_out_faces=True,
						  masked_faces=masked_faces, return_mask_out_masks=True)
		return gt, R, T, side

	def _render_pred(self, st, gt, R, T, side, copy_mask_out, images=True, mask_image=True):
		pred = self.rdr(st.res['meshes'], R, T, return_images=images, return_mask=True, mask_with_grad=True)
		if side is not None:
			main = torch.cuda.current_stream(side.device)
			for t in gt.values():   # allocated on the second stream, read on this one from here on
				if torch.is_tensor(t):
					t.record_stream(main)
			main.wait_stream(side)
		if copy_mask_out and not gt.get('nothing_hidden', False):  # what the GT's slicing plane hides is hidden in the prediction too (model.py:1091-1094)
			hidden = gt['mask_out_masks']
			# (the image is whitened only for a caller who looks at it: the pixel loss reads it through the mask, which is zero there)
			if images and mask_image:
				pred['image'] = torch.where(hidden.unsqueeze(-1), torch.ones_like(pred['image']), pred['image'])
			pred['mask'] = torch.where(hidden, torch.zeros_like(pred['mask']), pred['mask'])
		return pred

	def forward(self, batch, epoch, opts, chamf=False, smooth=False, texture=False, pix=False, vgg_perc=False, sil=False,
				restyle_perc_lat=False, restyle_perc_feat=False, restyle_perc_cluster=False, cont_pose=False, render_foot=False,
				save_renders=False, render_dir='_pix', is_train=True, use_z_cutoff=False, gt_z_cutoff=None, restyle_feature_maps=None,
				no_displacement=False, return_renders=False, copy_mask_out=True, mask_out_pred_faces=False, views=None):
		given = dict(vgg_perc=vgg_perc, restyle_perc_lat=restyle_perc_lat, restyle_perc_feat=restyle_perc_feat, restyle_perc_cluster=restyle_perc_cluster,
					 cont_pose=cont_pose, mask_out_pred_faces=mask_out_pred_faces)
		for name, why in OUT_OF_SCOPE_FLAGS.items():
			if given[name]:
				raise NotImplementedError(why)
		enabled = dict(chamf=chamf, smooth=smooth, texture=texture, pix=pix, sil=sil)

		st = _Step()
		st.batch, st.epoch, st.opts, st.is_train = batch, epoch, opts, is_train
		st.use_z_cutoff, st.gt_z_cutoff = use_z_cutoff, gt_z_cutoff
		# train_network asks for renders whenever a checkpoint is saved (train.py:58-66: save_renders at epoch 0, every *_save_every epochs
		# and on the last one), with or without a render loss: `if render_foot or save_renders` (model.py:1057)
		rendering = bool(render_foot or save_renders)
		# the images (shading, vertex normals) are rendered only when something reads them: the pixel loss, the caller or the PNG (the
		# reference renders them regardless, renderer.py:290-291; nothing downstream can tell)
		images = rendering and bool(pix or return_renders or save_renders)
		# The colours of the predicted mesh are read by the image render only: a step that renders no image -- nothing at all, or silhouettes
		# alone -- leaves the colour head of the template pass to whoever reads res['col'] / meshes.textures first (model.get_meshes:
		# lazy_colours): nobody, on the 3-D-loss stages and on a silhouette-loss step.
		dev = torch.device(batch['mesh'].device) if 'mesh' in batch else None
		supervise_3d = self._supervise_3d(batch, opts, is_train)
		# The texture term depends on the batch alone (GT scan, latent rows), not on the predicted mesh: issued first, on the second stream,
		# its MLP pass (matrix pipe) fills the holes of the main pass (sampling, nearest neighbours, smoothness, registration: no matrix
		# pipe) and the other way round -- forward and, since autograd replays a node on its forward's stream, backward.
		early = {}
		tex_side = None
		st.gt_chamf = st.gt_tex = None
		if (TEXTURE_STREAM == 1 and texture and supervise_3d and dev is not None and dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing()):
			tex_side = _second_stream(dev)
			main = torch.cuda.current_stream(dev)
			tex_side.wait_stream(main)   # (the latent rows of the batch were gathered on this stream)
			with torch.cuda.stream(tex_side):
				early['loss_tex'] = self._raw_texture(st)
			early['loss_tex'].record_stream(main)
		# The surface samples of the GT scans (Chamfer: 5000 per scan; texture: 1000 with colours) depend on the batch alone: drawn HERE, before
		# the main pass and -- outside a capture -- on the second stream beside it (round 5: they used to sit between the main pass and the
		# texture term's MLP pass, three short launches each on a chip the nearest-neighbour search was filling: 60 us for a 7-us kernel).
		pre_ev = None
		if PRESAMPLE_GT and supervise_3d and (chamf or texture) and tex_side is None and 'mesh' in batch:
			with_side = dev is not None and dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing()
			side = _second_stream(dev) if with_side else None
			if side is not None:
				side.wait_stream(torch.cuda.current_stream(dev))
			with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
				if chamf:
					st.gt_chamf = _losses.sample_points_from_meshes(batch['mesh'], num_samples=5000)
				if texture:
					st.gt_tex = _losses.sample_points_from_meshes(batch['mesh'], num_samples=1000, return_textures=True)
				if side is not None:
					pre_ev = torch.cuda.Event()
					pre_ev.record(side)
		st.res = self.model.get_meshes_from_batch(batch, is_train=is_train, no_displacement=no_displacement, **(dict(lazy_colours=True) if LAZY_COLOURS and not images else {}))
		if pre_ev is not None:
			main = torch.cuda.current_stream(dev)
			main.wait_event(pre_ev)   # (the texture term reads its samples on this stream)
			for t in (st.gt_chamf, *(st.gt_tex or ())):
				if torch.is_tensor(t):
					t.record_stream(main)
		st.pred = st.gt = None
		if rendering:
			st.gt, R, T, side = self._render_gt(st, views, batch.get('masked_faces', None), images)
			st.pred = self._render_pred(st, st.gt, R, T, side, copy_mask_out, images, mask_image=bool(return_renders or save_renders))
		raw, weights = {}, []
		active = [t for t in TERMS if enabled[t.flag] and not (t.needs_3d and not supervise_3d) and not (t.needs_render and not rendering)]
		# The Chamfer term -- surface sampling and a brute-force nearest-neighbour search: packed fp32 VALU work, no matrix pipe -- beside the
		# texture term's MLP pass (matrix pipe) on a second stream: they want different halves of a CU.  Autograd replays each term's
		# backward on the stream of its forward, so the two backward halves overlap as well.  Not under stream capture.
		aside = None
		if (OVERLAP_CHAMFER and tex_side is None and dev is not None and dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing()
				and any(t.flag == 'chamf' for t in active) and any(t.flag == 'texture' for t in active)):
			aside = _second_stream(dev)
		third = None
		if (TEXTURE_STREAM == 2 and aside is not None and any(t.needs_3d and t.flag != 'texture' for t in active)):
			third = _third_stream(dev)
		# (issue order = reverse backward order: with a third stream the texture term goes first, so that the loss-side chain the main pass's
		# backward waits for -- smoothness, Chamfer, registration -- is the first thing the backward pass enqueues)
		order = sorted(active, key=lambda t: t.flag != 'texture') if third is not None else active
		# Experiment (round 6, FIND_TEXTURE_FIRST=1; off): the texture term ISSUED before the Chamfer term, the second stream forked by an
		# event.  Under the tracer the texture pass's 5-us weight split waits 84 us for a CU behind the nearest-neighbour search's 1280
		# workgroups; untraced -- the host far ahead of the GPU -- the reordering costs 60 us per step (1.69 against 1.63 ms, three
		# alternating runs each): the traced timeline is not the untraced one.
		fork_ev = None
		if aside is not None and third is None and TEXTURE_FIRST:
			order = sorted(active, key=lambda t: {'texture': 0, 'chamf': 1}.get(t.flag, 2))
			fork_ev = torch.cuda.Event()
			fork_ev.record(torch.cuda.current_stream(dev))
		got = {}
		for term in order:
			if term.key in early:
				got[term.key] = early[term.key]
			elif (aside is not None and term.flag == 'chamf') or (third is not None and term.flag == 'texture'):
				side = aside if term.flag == 'chamf' else third
				main = torch.cuda.current_stream(dev)
				if fork_ev is not None:
					side.wait_event(fork_ev)
				else:
					side.wait_stream(main)
				with torch.cuda.stream(side):
					got[term.key] = getattr(self, term.fn)(st)
				got[term.key].record_stream(main)   # allocated on the side stream, read on this one
			else:
				got[term.key] = getattr(self, term.fn)(st)
		if TEX_BWD_LATE and 'loss_tex' in got and st.res is not None:
			# Backward order (round 6): autograd runs ready nodes by descending sequence number, i.e. in reverse order of creation -- the
			# texture term, created last, first.  Its MLP backward leaves ~110 us of weight-gradient work on the side streams, and the short
			# kernels of the loss-side chain that follow (smoothness, Chamfer, sampling backward -> d verts, which the main pass's backward
			# waits for) then crawl beside it: 409 us from the start of the backward to d verts against ~215 for the two chains alone
			# (tools/r6_phases.py, untraced).  The texture term's first backward node gets a sequence number just above the registration
			# node's instead: the loss-side chain runs first, on an empty chip, then the texture pass's chain, then registration.
			tex_fn, verts_fn = got['loss_tex'].grad_fn, st.res['verts'].grad_fn if torch.is_tensor(st.res.get('verts')) else None
			if tex_fn is not None and verts_fn is not None and hasattr(tex_fn, '_set_sequence_nr'):
				tex_fn._set_sequence_nr(verts_fn._sequence_nr() + 1)
		for term in active:   # (reported, and summed, in the registry's order whatever the issue order was)
			raw[term.key] = got[term.key]
			weights.append(float(getattr(opts, term.weight)))
		for side in (aside, third, tex_side):
			if side is not None:
				torch.cuda.current_stream(dev).wait_stream(side)
		if save_renders:
			self._save_renders(st, render_dir)
		# losses[k] = raw * opts.weight_k, loss = sum(losses.values())   (model.py:1157-1163): one launch for all terms (find_weighted_terms_*)
		if raw:
			loss, scaled = FN.weighted_terms(list(raw.values()), weights)
			losses = dict(zip(raw, scaled))
		else:
			loss, losses = 0, {}   # sum({}.values()) of the reference: the trainer's `if loss == 0: continue` (trainer.py:108) relies on it
		if return_renders and rendering:
			return loss, losses, dict(pred=st.pred, gt=st.gt)
		return loss, losses

	@staticmethod
	def _save_renders(st, render_dir):
		"""The GT | prediction strip of a step, `{render_dir}/{epoch:04d}_{idx:02d}.png` (model.py:1149-1156): the images of all feet and
		views stacked top to bottom, GT column left, prediction right, 8 bits per channel by truncation of 255 * value.  (The reference
		writes through cv2 after an RGB -> BGR swap, i.e. an RGB file; PIL writes the same pixels.  Reads batch['idx'] on the host, as the
		reference does: not for a captured step.)"""
		import numpy as np
		from PIL import Image
		idx = st.batch['idx'][0].item()
		gt, pred = st.gt['image'], st.pred['image']
		H, W = gt.shape[-3], gt.shape[-2]
		both = torch.cat([gt.detach().reshape(-1, W, 3), pred.detach().reshape(-1, W, 3)], dim=1)   # rows: (foot, view, y); columns: GT | prediction
		strip = (both.cpu().numpy() * 255).astype(np.uint8)
		os.makedirs(render_dir, exist_ok=True)
		Image.fromarray(strip, mode='RGB').save(os.path.join(render_dir, f'{st.epoch:04d}_{idx:02d}.png'))

	def save_model(self, *args, **kwargs):
		self.model.save_model(*args, **kwargs)
