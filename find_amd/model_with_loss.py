"""ModelWithLoss on the MI355X hot path: host-side mirror of reference src/model/model.py:969-1171 -- builds the model, the
renderer and the loss objects; forward(batch, epoch, opts, **flags) returns (loss, losses[, renders]) with every raw
loss multiplied by opts.weight_<name> (model.py:1157-1158).  All arithmetic below runs in libfind_hip.so."""
import torch

from .losses import DisplacementLoss, MeshSmoothnessLoss, SilhouetteLoss, TextureLossGTSpace
from .model import NeuralDisplacementField
from .renderer import FootRenderer

nn = torch.nn

model_zoo = dict(neural=NeuralDisplacementField)


def model_class_from_opts(opts):
	mt = getattr(opts, 'model_type', 'neural')
	if mt not in model_zoo:
		raise NotImplementedError(f"model_type '{mt}': only the neural displacement field is on the hot path (PCA / SUPR / "
								  'vertex-feature baselines are out of scope, SURVEY.md §2 #7-9)')
	return model_zoo[mt]


def model_from_opts(opts):
	return model_class_from_opts(opts).load(opts.load_model, device=opts.device, opts=opts)


class ModelWithLoss(nn.Module):
	def __init__(self, *args, opts=None, device='cuda', **kwargs):
		super().__init__()
		model_class = model_class_from_opts(opts)
		load = opts.load_model
		if load == '':
			self.model = model_class(*args, **kwargs, device=device, opts=opts)
		else:
			self.model = model_class.load(load, device=device, **kwargs, opts=opts)
		self.device = device
		self.def_loss = DisplacementLoss()
		self.col_loss = TextureLossGTSpace()
		self.mesh_smooth_loss = MeshSmoothnessLoss()
		self.templ_smooth_loss = MeshSmoothnessLoss()
		max_faces_per_bin = 30000 if not opts.low_poly_meshes else None  # reference heuristic; no effect on results
		self.rdr = FootRenderer(image_size=256, device=device, bin_size=None, max_faces_per_bin=max_faces_per_bin)
		self.pix_loss = nn.MSELoss()
		self.sil_loss = SilhouetteLoss()
		if opts.vgg_perc_loss or opts.use_restyle():
			raise NotImplementedError('VGG / Restyle perceptual losses need network weights that are not available; out of scope')

	def _views(self, opts):
		nviews = opts.num_views
		svt = opts.special_view_type
		if svt == 'topdown':
			return self.rdr.view_from('topdown')
		if svt == 'topdown_5':
			return self.rdr.combine_views(*self.rdr.view_from('topdown'),
										  *self.rdr.sample_views(nviews=5, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90,
																 azim_min=-90, azim_max=90, seed=5))
		if svt == 'sample_arc':
			return self.rdr.sample_views(nviews=nviews, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=0, azim_max=0)
		# same viewpoints for GT and prediction (model.py:1070-1071)
		return self.rdr.sample_views(nviews=nviews, dist_mean=0.3, dist_std=0, elev_min=-90, elev_max=90, azim_min=-90, azim_max=90)

	def forward(self, batch, epoch, opts, chamf=False, smooth=False, texture=False, pix=False, vgg_perc=False, sil=False,
				restyle_perc_lat=False, restyle_perc_feat=False, restyle_perc_cluster=False, cont_pose=False, render_foot=False,
				save_renders=False, render_dir='_pix', is_train=True, use_z_cutoff=False, gt_z_cutoff=None, restyle_feature_maps=None,
				no_displacement=False, return_renders=False, copy_mask_out=True, mask_out_pred_faces=False, views=None):
		if vgg_perc or restyle_perc_lat or restyle_perc_feat or restyle_perc_cluster or cont_pose:
			raise NotImplementedError('perceptual / restyle / contrastive losses are out of scope (SURVEY.md §2 #4)')
		if save_renders:
			raise NotImplementedError('save_renders writes PNGs through cv2 (visualisation); use return_renders and save them yourself')
		if mask_out_pred_faces:
			raise NotImplementedError('mask_out_pred_faces belongs to the VertexFeatures model (out of scope)')

		res = self.model.get_meshes_from_batch(batch, is_train=is_train, no_displacement=no_displacement)
		raw_losses = {}
		renders_to_return = dict()

		apply_loss_3d = True
		if is_train and getattr(opts, 'restrict_3d_n_train', None) is not None:
			if batch['idx'].item() >= opts.restrict_3d_n_train:
				apply_loss_3d = False
		if getattr(opts, 'restrict_3d_train_key', None) is not None:
			if batch['name'][0] not in opts.train_3d_on_only:
				apply_loss_3d = False

		if apply_loss_3d:
			if chamf:
				raw_losses['loss_chamf'] = self.def_loss(self.model, res, batch, epoch, z_cutoff=0.07 if use_z_cutoff else None,
														 gt_z_cutoff=gt_z_cutoff)['loss']
			if smooth:
				raw_losses['loss_smooth'] = self.mesh_smooth_loss(res['meshes'])
			if texture:
				sfx = 'train' if is_train else 'val'
				raw_losses['loss_tex'] = self.col_loss(self.model, batch, shapevec=batch.get(f'shapevec_{sfx}', None),
													   texvec=batch.get(f'texvec_{sfx}', None), posevec=batch.get(f'posevec_{sfx}', None))

		if render_foot:
			R, T = views if views is not None else self._views(opts)
			with torch.no_grad():  # the GT is re-rendered every step, as in the reference (model.py:1073-1075)
				gt_rdrs = self.rdr(batch['mesh'], R, T, return_mask=True, mask_with_grad=True, mask_out_faces=True,
								   masked_faces=batch.get('masked_faces', None), return_mask_out_masks=True)
			pred_rdrs = self.rdr(res['meshes'], R, T, return_mask=True, mask_with_grad=True)
			if copy_mask_out:  # apply the GT's masked-out region to the prediction (model.py:1091-1094)
				mo = gt_rdrs['mask_out_masks']
				pred_rdrs['image'] = torch.where(mo.unsqueeze(-1), torch.ones_like(pred_rdrs['image']), pred_rdrs['image'])
				pred_rdrs['mask'] = torch.where(mo, torch.zeros_like(pred_rdrs['mask']), pred_rdrs['mask'])
			if return_renders:
				renders_to_return.update(dict(pred=pred_rdrs, gt=gt_rdrs))
			if pix:
				pix_pred = pred_rdrs['image'] * pred_rdrs['mask'].unsqueeze(-1)
				pix_gt = gt_rdrs['image'] * gt_rdrs['mask'].unsqueeze(-1)
				raw_losses['loss_pix'] = self.pix_loss(pix_pred, pix_gt)
			if sil:
				raw_losses['loss_sil'] = self.sil_loss(pred_rdrs['mask'], gt_rdrs['mask'])

		losses = {k: v * getattr(opts, k.replace('loss', 'weight')) for k, v in raw_losses.items()}
		loss = sum(losses.values())
		if return_renders and render_foot:
			return loss, losses, renders_to_return
		return loss, losses

	def save_model(self, *args, **kwargs):
		self.model.save_model(*args, **kwargs)
