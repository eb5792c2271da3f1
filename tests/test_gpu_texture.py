"""TexturesUV through the C-ABI: find_uv_sample against oracle/texture_ref.py; the surface sampler's
return_textures and the renderer's GT images with UV maps."""
import numpy as np
import pytest
import torch

from oracle import camera_ref, texture_ref
from tests.test_oracle_texture import _scene

pytestmark = pytest.mark.gpu


def test_uv_sample_matches_oracle():
	from find_amd import functional_render as FR
	maps, vu, fu, fi, b = _scene(seed=3, Nm=3, H=37, W=29, Vt=40, F=25, P=700)
	got = FR.uv_sample(torch.from_numpy(maps).cuda(), torch.from_numpy(vu).cuda(), torch.from_numpy(fu).cuda(), torch.from_numpy(fi).cuda(),
					   torch.from_numpy(b).cuda())
	ref = texture_ref.uv_sample(maps, vu, fu, fi, b)
	assert np.abs(got.cpu().numpy() - ref).max() < 1e-5
	# rows of one map consecutive (feet x views): 2 rows per map
	fi2 = np.repeat(fi, 2, axis=0); b2 = np.repeat(b, 2, axis=0)
	got2 = FR.uv_sample(torch.from_numpy(maps).cuda(), torch.from_numpy(vu).cuda(), torch.from_numpy(fu).cuda(), torch.from_numpy(fi2).cuda(),
						torch.from_numpy(b2).cuda())
	assert torch.equal(got2[0::2], got) and torch.equal(got2[1::2], got)


def _uv_mesh(n=2):
	"""Ellipsoid scans whose UV vertices coincide with the mesh vertices (uv = normalised x, y) and a map that is LINEAR in (u, v):
	the texel at any surface point then equals the barycentric mix of the per-vertex colours, so TexturesUV and TexturesVertex agree."""
	from find_amd import synthetic
	from find_amd.structures import Meshes, TexturesUV, TexturesVertex
	v, f = synthetic.ellipsoid_mesh(14, 18)
	g = torch.Generator().manual_seed(1)
	verts = v[None] * (1 + 0.1 * torch.rand(n, 1, 3, generator=g))
	lo, hi = verts.amin(dim=1, keepdim=True), verts.amax(dim=1, keepdim=True)
	uv = ((verts - lo) / (hi - lo))[..., :2].contiguous()
	H, W = 64, 48
	yy, xx = torch.meshgrid(torch.linspace(1, 0, H), torch.linspace(0, 1, W), indexing='ij')  # v decreases down the rows
	maps = torch.stack([0.2 + 0.6 * xx, 0.1 + 0.8 * yy, 0.5 * xx + 0.4 * yy], dim=-1)[None].expand(n, -1, -1, -1).contiguous()
	cols = torch.stack([0.2 + 0.6 * uv[..., 0], 0.1 + 0.8 * uv[..., 1], 0.5 * uv[..., 0] + 0.4 * uv[..., 1]], dim=-1)
	tuv = TexturesUV(maps, f[None].expand(n, -1, -1).contiguous(), uv)
	return verts, f, tuv, TexturesVertex(cols), Meshes, cols


def test_sampler_return_textures_uv_equals_vertex_colours_on_a_linear_map():
	from find_amd.losses import sample_points_from_meshes
	verts, f, tuv, tv, Meshes, _ = _uv_mesh()
	g = torch.Generator().manual_seed(2)
	fi = torch.randint(0, f.shape[0], (2, 400), generator=g).cuda()
	uvd = torch.rand(2, 400, 2, generator=g).cuda()
	p1, c1 = sample_points_from_meshes(Meshes(verts.cuda(), f.cuda(), tuv.to('cuda')), 400, return_textures=True, draws=(fi, uvd))
	p2, c2 = sample_points_from_meshes(Meshes(verts.cuda(), f.cuda(), tv.to('cuda')), 400, return_textures=True, draws=(fi, uvd))
	assert torch.equal(p1, p2)
	assert (c1 - c2).abs().max().item() < 1e-5


def test_uv_textured_render_equals_vertex_colour_render_on_a_linear_map_and_masks_faces():
	from find_amd.renderer import FootRenderer
	from find_amd.structures import TexturesUV
	verts, f, tuv, tv, Meshes, _ = _uv_mesh()
	rdr = FootRenderer(image_size=96, device='cuda')
	rng = np.random.RandomState(5)
	R, T = camera_ref.look_at_view_transform(dist=np.full(3, 0.3), elev=rng.uniform(-60, 60, 3), azim=rng.uniform(-90, 90, 3), up=((1, 0, 0),))
	R, T = torch.from_numpy(R).cuda(), torch.from_numpy(T).cuda()
	a = rdr(Meshes(verts.cuda(), f.cuda(), tuv.to('cuda')), R, T, return_mask=True)
	b = rdr(Meshes(verts.cuda(), f.cuda(), tv.to('cuda')), R, T, return_mask=True)
	assert torch.equal(a['mask'], b['mask'])
	# perspective-correct barycentrics feed both paths; the only difference is bilinear-vs-barycentric rounding on a linear ramp
	assert (a['image'] - b['image']).abs().max().item() < 2e-4
	assert (a['image'] < 1).any()
	# mask-out-faces convention: a final UV vertex at (0,0) marks the faces to drop (renderer.py:340-349)
	n = verts.shape[0]
	vu = torch.cat([tuv.verts_uvs_padded(), torch.zeros(n, 1, 2)], dim=1)
	fu = tuv.faces_uvs_padded().clone()
	marked = torch.arange(0, f.shape[0], 3)
	fu[:, marked] = vu.shape[1] - 1
	tm = TexturesUV(tuv.maps_padded(), fu, vu)
	c = rdr(Meshes(verts.cuda(), f.cuda(), tm.to('cuda')), R, T, return_mask=True, mask_out_faces=True, return_mask_out_masks=True)
	mo = c['mask_out_masks']
	assert mo.any() and not mo.all()
	assert (c['image'][mo] == 1).all() and (c['mask'][mo] == 0).all()
	d = rdr(Meshes(verts.cuda(), f.cuda(), tv.to('cuda')), R, T, return_mask=True, mask_out_faces=True, masked_faces=marked, return_mask_out_masks=True)
	assert torch.equal(d['mask_out_masks'], mo)


def test_dataset_batch_feeds_texture_loss_and_gt_render(tmp_path):
	"""End to end for f1 + f2: OBJ/PNG/JSON folder -> Foot3DDataset -> BatchCollator (ragged Meshes + TexturesUV) -> the texture loss of
	the hot path (UV-sampled GT colours) and a UV-textured GT render."""
	import json, os
	from tests.test_host_dataset import CFG_POSE, _write_scan
	from find_amd import synthetic
	from find_amd.dataset import BatchCollator, Foot3DDataset
	from find_amd.losses import TextureLossGTSpace
	from find_amd.renderer import FootRenderer
	root = str(tmp_path)
	data = []
	for k, fid in enumerate(['0005', '0006']):
		rel = f'{fid}/A/{fid}-A'
		_write_scan(os.path.join(root, 'Meshes_sliced'), rel + '.obj', rel + '.png', 6 + k, (0.0, 0.0, 0.0))
		data.append({'Foot ID': fid, 'Scan ID': 'A', 'footedness': 'Left', 'pose': ['T-Pose'], 'keypoints': None, 'OBJ file': rel + '.obj', 'PNG file': rel + '.png'})
	jpath = os.path.join(root, 'index.json')
	with open(jpath, 'w') as fh:
		json.dump({'keypoint_labels': ['a'], 'data': data}, fh)
	cfg = {'DATASET_FOLDER': root, 'DATASET_JSON': jpath, 'DATASET_NAME': 'Meshes_sliced', 'LOWPOLY_DATASET_NAME': 'x', 'VAL_FEET': [], 'TEMPLATE_FEET': [],
		   'POSE_VECTOR': CFG_POSE}
	ds = Foot3DDataset(cfg, device='cpu')
	batch = BatchCollator(device='cuda').collate_batches([ds[0], ds[1]])
	model = synthetic.make_model(1002, train_size=2, val_size=1, device='cuda')
	lat = synthetic.latents(2, seed=0, device='cuda')
	loss = TextureLossGTSpace()(model, batch, num_samples=300, shapevec=lat['shapevec'], texvec=lat['texvec'], posevec=lat['posevec'])
	assert torch.isfinite(loss) and loss.item() > 0
	loss.backward()
	assert any(p.grad is not None and p.grad.abs().max() > 0 for p in model.mlp_col.parameters())
	rdr = FootRenderer(image_size=48, device='cuda')
	R, T = rdr.sample_views(nviews=2, dist_mean=0.3, dist_std=0, elev_min=20, elev_max=70, azim_min=-30, azim_max=30) if hasattr(rdr, 'sample_views') else None
	out = rdr(batch['mesh'], R, T, return_mask=True)
	assert out['image'].shape == (2, 2, 48, 48, 3) and torch.isfinite(out['image']).all() and (out['image'] < 1).any()
