"""Option object carrying the reference's flag NAMES and DEFAULTS that the hot path reads (reference
src/train/opts.py:11-256).  The reference builds this from argparse + YAML; that control plane is out of scope here --
this is a plain attribute bag so ModelWithLoss / tests / bench can be driven with the same field names, and a real
reference `Opts` instance can be passed instead (only attribute access is used)."""


class Opts:
	DEFAULTS = dict(
		model_type='neural', load_model='', device='cuda', low_poly_meshes=False, dont_load_latents=False,
		template_features_pth=None,
		# loss switches (opts.py:85-94)
		chamf_loss=False, smooth_loss=False, texture_loss=False, pix_loss=False, sil_loss=False, vgg_perc_loss=False,
		restyle_perc_lat_loss=False, restyle_perc_feat_loss=False, restyle_perc_cluster_loss=False, cont_pose_loss=False,
		# loss weights (opts.py:97-101)
		weight_chamf=10000., weight_smooth=1000., weight_pix=1., weight_vgg_perc=0.1, weight_restyle_perc_lat=0.25,
		weight_restyle_perc_feat=1., weight_tex=1., weight_sil=5., weight_restyle_perc_cluster=1., weight_cont_pose=1.,
		# renderer (opts.py:117-124)
		num_views=5, special_view_type=None, copy_over_masking=False, mask_out_pred=False,
		# experiment restrictions
		restrict_3d_n_train=None, restrict_3d_train_key=None, train_3d_on_only=None,
		restyle_features_per_vertex=False, restyle_cluster_per_vertex=False, restyle_no_masking=False,
		use_pose_code=False, use_latent_labels=False, gt_z_cutoff=None, use_z_cutoff=False,
	)

	def __init__(self, **kw):
		for k, v in self.DEFAULTS.items():
			setattr(self, k, v)
		for k, v in kw.items():
			self.set_option(k, v)

	def set_option(self, key, value):
		assert key in self.DEFAULTS, f'Option {key} not recognised'  # the reference asserts the key exists too (opts.py:187-188)
		setattr(self, key, value)

	def use_restyle(self):
		return self.restyle_perc_lat_loss or self.restyle_perc_feat_loss or self.restyle_perc_cluster_loss

	def net_train_kwargs(self):
		"""Loss flags handed to ModelWithLoss.forward for the network / latent stages (opts.py:207-215)."""
		return dict(chamf=self.chamf_loss, smooth=self.smooth_loss, texture=self.texture_loss, pix=self.pix_loss, sil=self.sil_loss,
					render_foot=self.pix_loss or self.sil_loss, copy_mask_out=self.copy_over_masking)
