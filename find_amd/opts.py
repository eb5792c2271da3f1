"""Option object carrying the reference's flag NAMES and DEFAULTS that the hot path reads (reference
src/train/opts.py:11-256).  The reference builds this from argparse + YAML; that control plane is out of scope here --
this is a plain attribute bag so ModelWithLoss / tests / bench can be driven with the same field names, and a real
reference `Opts` instance can be passed instead (only attribute access is used)."""


class Opts:
	restyle_losses = ['restyle_perc_lat', 'restyle_perc_feat', 'restyle_perc_cluster']
	render_losses = ['pix', 'sil', 'vgg_perc', 'restyle_perc_lat', 'restyle_perc_feat', 'restyle_perc_cluster']   # opts.py:12-13

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
		# what train.py's stages and train_network read around the step (opts.py:38-76,140-152): optimiser rates, schedule, switches
		batch_size_train=1, batch_size_val=1, lr_net=5e-5, lr_reg=1e-5, lr_val=5e-5, lr_latent=1e-4,
		reg=False, reg_epochs=2000, reg_save_every=250, no_net_train=False, net_epochs=1000, net_save_every=100, net_val_every=50,
		val_crit='chamf', no_latent_refinement=False, latent_epochs=500, latent_optim='Adam', latent_save_every=250, latent_val_every=25,
		no_rendering=False, restyle_feature_maps=None, only_classifier_head=False, step_per_epoch=False, net_repeat_dataset=1,
		render_dir='_pix',
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
		"""Loss flags handed to ModelWithLoss.forward for the network / latent stages: the reference's ten keys (opts.py:207-215).
		render_foot / save_renders / copy_mask_out ... are added by the caller (train.py:58-70; find_amd.trainer.stage_model_kwargs)."""
		return dict(chamf=self.chamf_loss, smooth=self.smooth_loss, texture=self.texture_loss, pix=self.pix_loss, sil=self.sil_loss,
					vgg_perc=self.vgg_perc_loss, restyle_perc_lat=self.restyle_perc_lat_loss, restyle_perc_feat=self.restyle_perc_feat_loss,
					restyle_perc_cluster=self.restyle_perc_cluster_loss, cont_pose=self.cont_pose_loss)
