"""FootRenderer on the MI355X hot path: host-side mirror of reference src/model/renderer.py (same class name, constructor
keywords, view helpers and forward() keywords / outputs); all rendering arithmetic runs in libfind_hip.so
(find_amd.functional_render).  Keypoint splatting and the N-channel feature shader are out of scope."""
from typing import Union

import numpy as np
import torch

from . import functional_render as FR
from .cameras import look_at_view_transform
from .structures import Meshes, TexturesUV, TexturesVertex

nn = torch.nn


class FootRenderer(nn.Module):
	def __init__(self, image_size, device='cuda', background_color=(1., 1., 1.), bin_size=None, z_clip_value=None, max_faces_per_bin=None):
		"""bin_size / max_faces_per_bin are accepted for signature compatibility: binning is an acceleration structure of
		PyTorch3D's CUDA rasteriser and must not change results; the HIP rasteriser tiles internally."""
		super().__init__()
		self.image_size = image_size
		self.device = device
		self.background_color = tuple(float(c) for c in background_color)
		self.bin_size, self.max_faces_per_bin = bin_size, max_faces_per_bin
		self.light_location = (0., 0., 100.)  # PointLights(location=[[0, 0, 100]])  (renderer.py:114)
		self.params = FR.make_params(image_size, faces_per_pixel=100, background=self.background_color, light_pos=self.light_location,
									 znear=0.02, z_clip=z_clip_value)

	# ------------------------------------------------------------------ camera poses
	# Every FIND camera looks along a ray through `at` with the world x axis as "up" (the foot's long axis; renderer.py:152,171,198).
	# What the callers rely on: (i) sample_views draws distance, elevation, azimuth -- in that order, `nviews` values each -- from numpy's
	# GLOBAL generator and reseeds it only for a truthy seed (renderer.py:146-151: `if seed`, so seed=0 means "do not reseed");
	# (ii) the six named poses below (renderer.py:176-196).  Both are data that must match; the code around them is this file's own.
	UP = ((1, 0, 0),)
	NAMED_VIEWS = {   # name: (dist, elev, azim, look-at point)
		'topdown': (0.30, 0, 0, (0, 0, 0)),
		'side1': (0.35, 90, 0, (0, 0, 0)),
		'side2': (0.35, -90, 180, (0, 0, 0)),
		'toes': (0.10, 0, 0, (0.1, 0, 0)),
		'45': (0.35, -45, 0, (0, 0, 0)),
		'60': (0.35, -60, 0, (0, 0, 0)),
	}

	@classmethod
	def _poses(cls, dist, elev, azim, at=((0, 0, 0),)):
		return look_at_view_transform(dist=dist, elev=elev, azim=azim, up=cls.UP, at=at)

	def sample_views(self, nviews=1, dist_mean=.25, dist_std=0.05, elev_min=-90, elev_max=90, azim_min=0, azim_max=360, seed: int = None):
		"""Random poses: dist ~ N(dist_mean, dist_std), elev ~ U(elev_min, elev_max), azim ~ U(azim_min, azim_max), degrees."""
		if seed:
			np.random.seed(seed)
		draws = [np.random.normal(dist_mean, dist_std, nviews)]
		draws += [np.random.uniform(lo, hi, nviews) for lo, hi in ((elev_min, elev_max), (azim_min, azim_max))]
		return self._poses(*draws)

	def linspace_views(self, nviews=1, dist=.3, dist_min=None, dist_max=None, elev_min=None, elev_max=None, azim_min=None, azim_max=None,
					   at=((0, 0, 0),)):
		"""Evenly spaced poses: each of dist / elev / azim sweeps its [min, max] when a min is given, else stays at dist / 0 / 0."""
		def sweep(lo, hi, fixed):
			return fixed if lo is None else np.linspace(lo, hi, nviews)
		return self._poses(sweep(dist_min, dist_max, dist), sweep(elev_min, elev_max, 0), sweep(azim_min, azim_max, 0), at=at)

	def view_from(self, view_kw='topdown'):
		"""One pose per name in NAMED_VIEWS (a name or a list of names)."""
		names = [view_kw] if isinstance(view_kw, str) else list(view_kw)
		for v in names:
			assert v in self.NAMED_VIEWS, f'View description `{view_kw}` not understood'
		dist, elev, azim, at = zip(*(self.NAMED_VIEWS[v] for v in names))
		return self._poses(np.array(dist), np.array(elev), np.array(azim), at=np.array(at, dtype=np.float64))

	def combine_views(self, R1, T1, R2, T2):
		return torch.cat([R1, R2], dim=0), torch.cat([T1, T2], dim=0)

	# ------------------------------------------------------------------ render
	def forward(self, input_meshes: Meshes, R, T, return_images=True, return_depth=False, return_mask=False, mask_with_grad=True,
				mask_out_faces=False, masked_faces=None, keypoints=None, keypoints_blend=False, lights=None, return_mask_out_masks=False,
				return_features=False, features=None) -> dict:
		"""Render N meshes from M views: image [N,M,H,W,3], mask [N,M,H,W] (soft silhouette when mask_with_grad), optional
		depth and mask-out masks (reference renderer.py:247-383).  Image index = mesh*M + view."""
		if keypoints is not None or keypoints_blend:
			raise NotImplementedError('keypoint splatting (points renderer) is visualisation-only and out of scope')
		if return_features or features is not None:
			raise NotImplementedError('per-vertex feature rendering needs the restyle encoder and is out of scope')
		if lights is not None:
			raise NotImplementedError('custom lights are not used on the FIND path; the renderer keeps PointLights((0,0,100))')
		dev = input_meshes.device
		R, T = R.to(dev).float(), T.to(dev).float()
		N, M = len(input_meshes), R.shape[0]
		verts = input_meshes.verts_padded()
		faces = input_meshes.faces_shared()
		if faces is None:
			faces = input_meshes.faces_padded()
		F = faces.shape[-2]
		colors = None
		tex = input_meshes.textures
		uv_tex = isinstance(tex, TexturesUV)
		if return_images and not uv_tex:
			if not isinstance(tex, TexturesVertex):
				raise NotImplementedError('return_images needs TexturesVertex or TexturesUV textures')
			colors = tex.verts_features_padded()[..., :3]
		want_soft = return_mask and (mask_with_grad or not return_images)
		# (pix_to_face is read below only to hide faces: when the caller names some, or for UV textures' (0,0)-UV convention)
		want_frags = (mask_out_faces and (masked_faces is not None or uv_tex)) or return_depth
		if not (return_images or want_soft or want_frags):
			return dict()
		if return_images and uv_tex:
			# GT scans (dataset.py:263-271): no gradient flows to a UV-textured mesh anywhere in the reference
			if verts.requires_grad:
				raise NotImplementedError('UV-textured meshes are rendered without gradient (GT scans); use TexturesVertex for predicted meshes')
			mask, renders, p2f, zbuf = FR.render_uv(verts, tex, faces, R, T, self.params, want_mask=want_soft, want_frags=want_frags)
		else:
			mask, renders, p2f, zbuf = FR.render(verts, colors, faces, R, T, self.params, want_mask=want_soft, want_image=return_images,
												 want_frags=want_frags)
		out = dict()
		if return_depth:
			out['depth'] = zbuf
		if return_mask and not want_soft:  # hard mask from the image render (renderer.py:313)
			mask = torch.any(renders < 1, dim=-1).float()

		# (the all-False map costs a 16-MB fill @512^2: it is made where somebody hides faces or asks for the map, and `nothing_hidden` -- not in
		# the reference -- tells ModelWithLoss that copying it into the prediction would change nothing)
		hides = mask_out_faces and (masked_faces is not None or uv_tex)
		mask_out = torch.zeros((N, M, self.params.image_h, self.params.image_w), dtype=torch.bool, device=dev) if (hides or return_mask_out_masks) else None
		nothing_hidden = not hides
		if hides:
			img = torch.arange(N * M, device=dev, dtype=torch.int32).view(N, M, 1, 1)
			local = torch.where(p2f >= 0, p2f - img * F, p2f)  # face count within the mesh (renderer.py:322-327)
			for n in range(N):
				if masked_faces is not None:
					mf = masked_faces[n] if isinstance(masked_faces, list) else masked_faces
				else:
					# TexturesUV convention (renderer.py:340-349): a final UV vertex at (0, 0) marks faces to mask out -- those whose three
					# UV indices all point at it; the first mesh without the marker ends the search, as the reference's `break`
					vu, fu = tex.verts_uvs_padded()[n], tex.faces_uvs_padded()[n]
					if not bool((vu[-1] == 0).all()):
						nothing_hidden = nothing_hidden or n == 0
						break
					mf = torch.argwhere(torch.all(fu == vu.shape[0] - 1, dim=-1)).flatten()
				mask_out[n] = torch.isin(local[n], mf.to(dev).to(local.dtype))
			if return_images:
				renders = torch.where(mask_out.unsqueeze(-1), torch.ones_like(renders), renders)
			if return_mask:
				mask = torch.where(mask_out, torch.zeros_like(mask), mask)

		if return_images:
			out['image'] = renders
		if return_mask:
			out['mask'] = mask
		if return_mask_out_masks:
			out['mask_out_masks'] = mask_out
			out['nothing_hidden'] = nothing_hidden
		return out
