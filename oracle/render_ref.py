"""Oracle (TEST INFRASTRUCTURE ONLY) for rows a7-a10: render path of FootRenderer.forward (src/model/renderer.py:247-383).

Two layers:
  * C (oracle/raster_ref.c via ctypes): projection, NAIVE rasterisation (top-K by depth), soft silhouette, Phong + softmax
    blend -- the forward oracle and the CPU baseline (it mirrors PyTorch3D's CPU algorithm: every pixel x every face).
  * torch (this file): a differentiable restatement of the per-fragment math given the DISCRETE selection
    (pix_to_face) made by the C rasteriser, so torch.autograd supplies reference gradients w.r.t. vertices/colours.
    PyTorch3D's hand-written rasteriser backward equals the true derivative of these formulas (the point-segment
    distance treats the clamped projection parameter as constant, which is exact by the envelope theorem).

PARITY: the blend (softmax_blend, silhouette alpha) is PINNED to the reference's own softmax_blend (src/model/renderer.py:23-72, imported
and run by tests/golden/make_golden_pins.py -> tests/golden/blend.npz; tests/test_oracle_pins.py).  The rasteriser, the Phong shading and
the projection stay UNPINNED (PyTorch3D absent, no reference tests): anchored by tests/test_oracle_raster.py."""
import ctypes
import math
import os

import numpy as np
import torch

from . import camera_ref

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RenderParams(ctypes.Structure):
	_fields_ = [('image_h', ctypes.c_int32), ('image_w', ctypes.c_int32), ('fov_deg', ctypes.c_float), ('znear', ctypes.c_float),
				('zfar', ctypes.c_float), ('sil_blur_radius', ctypes.c_float), ('sil_sigma', ctypes.c_float),
				('sil_faces_per_pixel', ctypes.c_int32), ('rgb_sigma', ctypes.c_float), ('rgb_gamma', ctypes.c_float),
				('background', ctypes.c_float * 3), ('light_pos', ctypes.c_float * 3), ('ambient', ctypes.c_float),
				('diffuse', ctypes.c_float), ('specular', ctypes.c_float), ('shininess', ctypes.c_float), ('z_clip', ctypes.c_float)]


def default_params(image_size=256, faces_per_pixel=100):
	"""FootRenderer's settings (renderer.py:113-128, 274) + PyTorch3D defaults (BlendParams, PointLights, Materials)."""
	p = RenderParams()
	p.image_h = p.image_w = image_size
	p.fov_deg, p.znear, p.zfar = 60.0, 0.02, 100.0
	p.sil_sigma = 1e-4
	p.sil_blur_radius = float(np.log(1. / 1e-4 - 1.) * 1e-4)
	p.sil_faces_per_pixel = faces_per_pixel
	p.rgb_sigma, p.rgb_gamma = 1e-4, 1e-4
	p.background[:] = [1., 1., 1.]
	p.light_pos[:] = [0., 0., 100.]
	p.ambient, p.diffuse, p.specular, p.shininess = 0.5, 0.3, 0.2, 64.0
	p.z_clip = 0.01
	return p


def lib():
	global _LIB
	if _LIB is None:
		path = os.path.join(_HERE, '_build', 'liboracle.so')
		if not os.path.exists(path):
			import subprocess
			subprocess.run(['make', '-C', _HERE, '-s'], check=True)
		_LIB = ctypes.CDLL(path)
	return _LIB


def _f32(a):
	return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
	return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
	return a.ctypes.data_as(ctypes.c_void_p)


def project(rp, verts, R, T):
	verts, R, T = _f32(verts), _f32(R), _f32(T)
	N, V, _ = verts.shape
	M = R.shape[0]
	out = np.empty((N * M, V, 3), np.float32)
	lib().ref_project(ctypes.byref(rp), _p(verts), _p(R), _p(T), N, M, V, _p(out))
	return out


def rasterize(vproj, faces, n_views, H, W, K, blur_radius, perspective_correct=True, clip_bary=None, cull_backfaces=False, z_clip=0.01):
	"""Naive rasterisation.  faces (F,3) shared or (n_meshes,F,3).  Returns pix_to_face, zbuf, bary, dists, each (n_img,H,W,K[,3])."""
	vproj, faces = _f32(vproj), _i32(faces)
	n_img, V, _ = vproj.shape
	fb = 1 if faces.ndim == 2 else faces.shape[0]
	F = faces.shape[-2]
	if clip_bary is None:
		clip_bary = blur_radius > 0.0  # renderer.py:219-221
	p2f = np.empty((n_img, H, W, K), np.int32)
	zbuf = np.empty((n_img, H, W, K), np.float32)
	bary = np.empty((n_img, H, W, K, 3), np.float32)
	dists = np.empty((n_img, H, W, K), np.float32)
	lib().ref_rasterize(_p(vproj), _p(faces), fb, n_img, n_views, V, F, H, W, K, ctypes.c_float(blur_radius), int(perspective_correct),
						int(clip_bary), int(cull_backfaces), ctypes.c_float(z_clip), _p(p2f), _p(zbuf), _p(bary), _p(dists))
	return p2f, zbuf, bary, dists


def silhouette(p2f, dists, sigma=1e-4):
	p2f, dists = _i32(p2f), _f32(dists)
	K = p2f.shape[-1]
	mask = np.empty(p2f.shape[:-1], np.float32)
	lib().ref_silhouette(_p(p2f), _p(dists), ctypes.c_int64(mask.size), K, ctypes.c_float(sigma), _p(mask))
	return mask


def softmax_blend(p2f, dists, zbuf, colors, sigma=1e-4, gamma=1e-4, znear=0.02, zfar=100.0, background=(1., 1., 1.)):
	"""FootRenderer's softmax_blend (src/model/renderer.py:23-72) over fragment buffers: p2f / dists / zbuf (..., K), colors (..., K, C)
	-> (..., C).  The same C function blends the K = 1 fragments of render(); pinned by tests/golden/blend.npz."""
	p2f, dists, zbuf, colors = _i32(p2f), _f32(dists), _f32(zbuf), _f32(colors)
	K, C = p2f.shape[-1], colors.shape[-1]
	out = np.empty(p2f.shape[:-1] + (C,), np.float32)
	bg = _f32(background)
	lib().ref_softmax_blend(_p(p2f), _p(dists), _p(zbuf), _p(colors), ctypes.c_int64(out.size // C), K, C, ctypes.c_float(sigma), ctypes.c_float(gamma),
							ctypes.c_float(znear), ctypes.c_float(zfar), _p(bg), _p(out))
	return out


def vertex_normals(verts, faces):
	verts, faces = _f32(verts), _i32(faces)
	N, V, _ = verts.shape
	fb = 1 if faces.ndim == 2 else faces.shape[0]
	out = np.empty_like(verts)
	lib().ref_vertex_normals(_p(verts), _p(faces), fb, N, V, faces.shape[-2], _p(out))
	return out


def render(verts, faces, colors, R, T, image_size=256, want_mask=True, want_image=True, rp=None):
	"""FootRenderer.forward numerics: returns dict(mask (N,M,H,W), image (N,M,H,W,3), pix_to_face, zbuf (K=1 fragments))."""
	rp = rp or default_params(image_size)
	verts, R, T = _f32(verts), _f32(R), _f32(T)
	N, V, _ = verts.shape
	M = R.shape[0]
	H, W = rp.image_h, rp.image_w
	vproj = project(rp, verts, R, T)
	out = {}
	if want_mask:
		p2f, zb, ba, di = rasterize(vproj, faces, M, H, W, rp.sil_faces_per_pixel, rp.sil_blur_radius, z_clip=rp.z_clip)
		out['mask'] = silhouette(p2f, di, rp.sil_sigma).reshape(N, M, H, W)
	if want_image:
		p2f, zb, ba, di = rasterize(vproj, faces, M, H, W, 1, 0.0, z_clip=rp.z_clip)
		nrm = vertex_normals(verts, faces)
		colors = _f32(colors)
		fc = _i32(faces)
		fb = 1 if fc.ndim == 2 else fc.shape[0]
		cc = camera_ref.camera_center(R, T)
		img = np.empty((N * M, H, W, 3), np.float32)
		lib().ref_phong_blend(ctypes.byref(rp), _p(p2f), _p(zb), _p(ba), _p(di), _p(verts), _p(nrm), _p(colors), _p(fc), fb, _p(cc),
							  N * M, M, V, fc.shape[-2], _p(img))
		out['image'] = img.reshape(N, M, H, W, 3)
		out['pix_to_face'] = p2f[..., 0].reshape(N, M, H, W)
		out['zbuf'] = zb[..., 0].reshape(N, M, H, W)
	return out


# ------------------------------------------------------------------------------------------------ differentiable part
def _edge(px, py, ax, ay, bx, by):
	return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def _pld(px, py, ax, ay, bx, by):
	bax, bay = bx - ax, by - ay
	l2 = bax * bax + bay * bay
	t = ((bax * (px - ax) + bay * (py - ay)) / l2.clamp(min=1e-30)).clamp(0.0, 1.0)
	qx, qy = ax + t * bax, ay + t * bay
	d = (qx - px) ** 2 + (qy - py) ** 2
	dend = (px - bx) ** 2 + (py - by) ** 2
	return torch.where(l2 <= 1e-8, dend, d)


def torch_project(verts, R, T, fov_deg=60.0):
	"""(N,V,3) x (M,3,3),(M,3) -> (N*M, V, 3) (x_ndc, y_ndc, z_view); image index = n*M + m."""
	s = 1.0 / math.tan(math.radians(fov_deg) / 2)
	pv = torch.einsum('nvi,mij->nmvj', verts, R) + T[None, :, None, :]
	z = pv[..., 2]
	out = torch.stack([s * pv[..., 0] / z, s * pv[..., 1] / z, z], dim=-1)
	return out.reshape(-1, verts.shape[1], 3)


def torch_fragments(vproj, faces, p2f, n_views, H, W, clip_bary, perspective_correct=True, pixels=None):
	"""Differentiably recompute (zbuf, bary, signed dists) of the fragments selected in p2f (packed ids).
	Dense form: p2f (n_img,H,W,K), every pixel.  Compact form (pixels=(img, yi, xi), three (P,) index tensors): p2f (P,K), only those
	pixels -- the full-size tests hand over the covered pixels only (a foot fills a fifth of the image; the rest has no fragment)."""
	n_img, V, _ = vproj.shape
	faces = faces.long()
	F = faces.shape[-2]
	valid = p2f >= 0
	if pixels is None:
		img = torch.arange(n_img).view(n_img, 1, 1, 1).expand_as(p2f)
		yy = (1.0 - (2.0 * torch.arange(H, dtype=vproj.dtype) + 1.0) / H).view(1, H, 1, 1)
		xx = (1.0 - (2.0 * torch.arange(W, dtype=vproj.dtype) + 1.0) / W).view(1, 1, W, 1)
		px, py = xx.expand(p2f.shape), yy.expand(p2f.shape)
	else:
		pi, pyi, pxi = pixels
		img = pi.view(-1, 1).expand_as(p2f)
		py = (1.0 - (2.0 * pyi.to(vproj.dtype) + 1.0) / H).view(-1, 1).expand(p2f.shape)
		px = (1.0 - (2.0 * pxi.to(vproj.dtype) + 1.0) / W).view(-1, 1).expand(p2f.shape)
	f = (p2f - img * F).clamp(min=0)
	if faces.dim() == 2:
		fv = faces[f]  # (..., K, 3)
	else:
		fv = faces[(img // n_views), f]
	vsel = vproj[img.unsqueeze(-1).expand_as(fv), fv]  # (..., K, 3 verts, 3 comps)
	x0, y0, z0 = vsel[..., 0, 0], vsel[..., 0, 1], vsel[..., 0, 2]
	x1, y1, z1 = vsel[..., 1, 0], vsel[..., 1, 1], vsel[..., 1, 2]
	x2, y2, z2 = vsel[..., 2, 0], vsel[..., 2, 1], vsel[..., 2, 2]
	area = _edge(x2, y2, x0, y0, x1, y1) + 1e-8
	w0 = _edge(px, py, x1, y1, x2, y2) / area
	w1 = _edge(px, py, x2, y2, x0, y0) / area
	w2 = _edge(px, py, x0, y0, x1, y1) / area
	if perspective_correct:
		t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2
		den = (t0 + t1 + t2).clamp(min=1e-8)
		w0, w1, w2 = t0 / den, t1 / den, t2 / den
	inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
	c0, c1, c2 = w0, w1, w2
	if clip_bary:
		c0, c1, c2 = w0.clamp(min=0), w1.clamp(min=0), w2.clamp(min=0)
		s = (c0 + c1 + c2).clamp(min=1e-5)
		c0, c1, c2 = c0 / s, c1 / s, c2 / s
	pz = c0 * z0 + c1 * z1 + c2 * z2
	d = torch.minimum(torch.minimum(_pld(px, py, x0, y0, x1, y1), _pld(px, py, x0, y0, x2, y2)), _pld(px, py, x1, y1, x2, y2))
	dist = torch.where(inside, -d, d)
	bary = torch.stack([c0, c1, c2], dim=-1)
	return pz, bary, dist, valid, fv


def covered_pixels(p2f, order=None):
	"""(img, yi, xi) index tensors of the pixels of a dense (n_img,H,W,K) selection that hold at least one fragment (slot 0: the
	fragments of a pixel are sorted, empty slots last).  order: 'reverse' or an int seed -- the same pixels in another order (the gradient
	scatter then sums each vertex's contributions in another order: how the full-size tests measure the summation-order term of fp32)."""
	pix = torch.nonzero(p2f[..., 0] >= 0, as_tuple=True)
	if order is None:
		return pix
	n = pix[0].shape[0]
	perm = torch.arange(n - 1, -1, -1) if order == 'reverse' else torch.randperm(n, generator=torch.Generator().manual_seed(int(order)))
	return tuple(p[perm] for p in pix)


def torch_silhouette(dist, valid, sigma=1e-4):
	prob = torch.sigmoid(-dist / sigma) * valid
	return 1.0 - torch.prod(1.0 - prob, dim=-1)


def torch_vertex_normals(verts, faces):
	faces = faces.long()
	N, V, _ = verts.shape
	out = []
	for n in range(N):
		f = faces if faces.dim() == 2 else faces[n]
		v = verts[n]
		fn = torch.linalg.cross(v[f[:, 2]] - v[f[:, 1]], v[f[:, 0]] - v[f[:, 1]], dim=1)
		vn = torch.zeros_like(v)
		for k in range(3):
			vn = vn.index_add(0, f[:, k], fn)
		out.append(vn / vn.norm(dim=1, keepdim=True).clamp(min=1e-6))
	return torch.stack(out)


def torch_phong_image(rp, verts, colors, faces, R, T, p2f1, n_views, compact=False, order=None):
	"""Differentiable K=1 Phong image given the hard selection p2f1 (n_img,H,W,1).  compact: evaluate the covered pixels only (the others
	are background exactly: no fragment, weights zero) -- same numbers, a fraction of the memory at 256^2 / 512^2."""
	N, V, _ = verts.shape
	H, W = rp.image_h, rp.image_w
	vproj = torch_project(verts, R, T, rp.fov_deg)
	n_img = vproj.shape[0]
	bg = torch.tensor(list(rp.background), dtype=verts.dtype)
	if compact:
		pix = covered_pixels(p2f1, order)
		sel = p2f1[pix]                                   # (P, 1)
		pz, bary, dist, valid, fv = torch_fragments(vproj, faces, sel, n_views, H, W, clip_bary=False, pixels=pix)
		img_of = pix[0].view(-1, 1, 1)
	else:
		pz, bary, dist, valid, fv = torch_fragments(vproj, faces, p2f1, n_views, H, W, clip_bary=False)
		img_of = torch.arange(n_img).view(n_img, 1, 1, 1, 1)
	mesh = (img_of // n_views).expand_as(fv)
	nrm = torch_vertex_normals(verts, faces)

	def interp(attr):
		return (bary.unsqueeze(-1) * attr[mesh, fv]).sum(dim=-2)  # (..., 1, 3)

	pos, nn, tex = interp(verts), interp(nrm), interp(colors)
	n = nn / nn.norm(dim=-1, keepdim=True).clamp(min=1e-6)
	light = torch.tensor(list(rp.light_pos), dtype=verts.dtype)
	l = light - pos
	l = l / l.norm(dim=-1, keepdim=True).clamp(min=1e-6)
	cosang = (n * l).sum(-1)
	diff = rp.diffuse * torch.relu(cosang)
	cc = -torch.einsum('mj,mij->mi', T, R)  # camera centres
	view = (img_of % n_views).reshape(img_of.shape[:-1])  # (..., 1)
	vd = cc[view] - pos
	vd = vd / vd.norm(dim=-1, keepdim=True).clamp(min=1e-6)
	r = -l + 2 * cosang.unsqueeze(-1) * n
	al = torch.relu((vd * r).sum(-1)) * (cosang > 0)
	spec = rp.specular * al ** rp.shininess
	col = (rp.ambient + diff).unsqueeze(-1) * tex + spec.unsqueeze(-1)
	img = torch_softmax_blend(col, dist, pz, valid, rp.rgb_sigma, rp.rgb_gamma, rp.znear, rp.zfar, bg)
	if compact:
		full = bg.expand(n_img, H, W, 3).clone()
		img = full.index_put(pix, img)
	return img.reshape(N, n_views, H, W, 3)


def torch_softmax_blend(col, dist, pz, valid, sigma, gamma, znear, zfar, background):
	"""Differentiable softmax blend (renderer.py:46-70): col (..., K, C), dist / pz / valid (..., K) -> (..., C).  Pinned by blend.npz."""
	eps = 1e-10
	prob = torch.sigmoid(-dist / sigma) * valid
	z_inv = (zfar - pz) / (zfar - znear) * valid
	z_inv_max = z_inv.max(dim=-1, keepdim=True).values.clamp(min=eps)
	wnum = prob * torch.exp((z_inv - z_inv_max) / gamma)
	delta = torch.exp((eps - z_inv_max) / gamma).clamp(min=eps)
	den = wnum.sum(-1, keepdim=True) + delta
	return ((wnum.unsqueeze(-1) * col).sum(-2) + delta * background) / den


def torch_mask(rp, verts, faces, R, T, p2f, n_views, compact=False, order=None):
	"""Differentiable soft silhouette given the K-fragment selection p2f (n_img,H,W,K).  compact: the covered pixels only (an empty
	pixel's mask is 1 - prod(1) = 0 exactly)."""
	N = verts.shape[0]
	vproj = torch_project(verts, R, T, rp.fov_deg)
	if compact:
		pix = covered_pixels(p2f, order)
		pz, bary, dist, valid, fv = torch_fragments(vproj, faces, p2f[pix], n_views, rp.image_h, rp.image_w, clip_bary=True, pixels=pix)
		m = torch_silhouette(dist, valid, rp.sil_sigma)
		out = torch.zeros(vproj.shape[0], rp.image_h, rp.image_w, dtype=verts.dtype).index_put(pix, m)
		return out.reshape(N, n_views, rp.image_h, rp.image_w)
	pz, bary, dist, valid, fv = torch_fragments(vproj, faces, p2f, n_views, rp.image_h, rp.image_w, clip_bary=True)
	return torch_silhouette(dist, valid, rp.sil_sigma).reshape(N, n_views, rp.image_h, rp.image_w)
