"""find_amd.dataset on a synthetic Foot3D-shaped folder: OBJ + PNG + JSON index, the reference's filters, item
keys and quirks (src/data/dataset.py:116-299), the collator's ragged Meshes + joined TexturesUV."""
import json
import os

import numpy as np
import pytest
import torch

CFG_POSE = {'SIZE': 8, 0: ['T-Pose'], 1: ['Plantarflex', 'Dorsiflex'], 2: ['Inversion', 'Eversion'], 3: ['Lateral', 'Medial'],
			4: ['Toe Flexion', 'Toe Extension'], 5: ['Toe Abduction', 'Toe Adduction'], 6: ['Standing on Floor'], 7: ['Tiptoes']}


def _write_scan(folder, rel_obj, rel_png, n_side, offset, quad=False):
	from PIL import Image
	os.makedirs(os.path.dirname(os.path.join(folder, rel_obj)), exist_ok=True)
	g = np.random.RandomState(n_side)
	lines = []
	for i in range(n_side):
		for j in range(n_side):
			lines.append(f'v {i * 0.01 + offset[0]:.6f} {j * 0.01 + offset[1]:.6f} {0.002 * ((i * j) % 3) + offset[2]:.6f}')
	for i in range(n_side):
		for j in range(n_side):
			lines.append(f'vt {i / (n_side - 1):.6f} {j / (n_side - 1):.6f}')
	idx = lambda i, j: i * n_side + j + 1
	for i in range(n_side - 1):
		for j in range(n_side - 1):
			a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
			if quad:
				lines.append(f'f {a}/{a} {b}/{b} {c}/{c} {d}/{d}')
			else:
				lines.append(f'f {a}/{a} {b}/{b} {c}/{c}')
				lines.append(f'f {a}/{a} {c}/{c} {d}/{d}')
	with open(os.path.join(folder, rel_obj), 'w') as fh:
		fh.write('# synthetic scan\nmtllib none.mtl\n' + '\n'.join(lines) + '\n')
	img = (g.rand(8, 6, 3) * 255).astype(np.uint8)
	Image.fromarray(img).save(os.path.join(folder, rel_png))
	return img


@pytest.fixture()
def foot3d(tmp_path):
	root = str(tmp_path)
	mesh_dir = os.path.join(root, 'Meshes_sliced')
	scans = [('0003', 'A', 'Left', ['T-Pose'], None), ('0005', 'A', 'Left', ['T-Pose'], [1, 2, 3]), ('0005', 'B', 'Left', ['Strong Dorsiflex', 'Eversion'], None),
			 ('0007', 'A', 'Right', ['Tiptoes'], [4, 5, 6]), ('0021', 'A', 'Left', ['T-Pose'], None)]
	data, imgs = [], {}
	for k, (fid, sid, side, pose, kps) in enumerate(scans):
		rel = f'{fid}/{sid}/{fid}-{sid}'
		imgs[f'{fid}-{sid}'] = _write_scan(mesh_dir, rel + '.obj', rel + '.png', 4 + k, (0.1 * k, -0.05, 0.02), quad=(k == 2))
		data.append({'Foot ID': fid, 'Scan ID': sid, 'footedness': side, 'pose': pose, 'keypoints': kps, 'OBJ file': rel + '.obj', 'PNG file': rel + '.png'})
	jpath = os.path.join(root, 'index.json')
	with open(jpath, 'w') as fh:
		json.dump({'keypoint_labels': ['a', 'b', 'c'], 'data': data}, fh)
	cfg = {'DATASET_FOLDER': root, 'DATASET_JSON': jpath, 'DATASET_NAME': 'Meshes_sliced', 'LOWPOLY_DATASET_NAME': 'Meshes_sliced_simplified',
		   'VAL_FEET': ['0021'], 'TEMPLATE_FEET': ['0003'], 'POSE_VECTOR': CFG_POSE}
	return cfg, imgs


def test_filters_keys_and_items(foot3d):
	from find_amd.dataset import Foot3DDataset
	cfg, imgs = foot3d
	tr = Foot3DDataset(cfg, device='cpu')
	assert [f'{a}-{b}' for a, b in zip(tr.foot_ids, tr.scan_ids)] == ['0005-A', '0005-B']      # template, val and right feet dropped
	assert len(Foot3DDataset(cfg, device='cpu', left_only=False)) == 3
	assert Foot3DDataset(cfg, device='cpu', is_train=False).foot_ids == ['0021']
	assert Foot3DDataset(cfg, device='cpu', tpose_only=True).scan_ids == ['A']
	assert Foot3DDataset(cfg, device='cpu', specific_feet=['0003']).foot_ids == ['0003']
	assert tr.get_all_keys() == {'shape': ['0005'], 'pose': ['0005-A', '0005-B'], 'tex': ['0005'], 'reg': ['0005-A', '0005-B']}
	it = tr[1]
	assert set(it) == {'faces', 'verts', 'textures', 'idx', 'name', 'has_keypoints', 'kp_idxs', 'is_tpose', 'orig_footedness', 'pose_descr', 'pose_code',
					   'shape', 'pose', 'tex', 'reg'}
	assert it['name'] == '0005-B' and not it['has_keypoints'] and (it['kp_idxs'] == 0).all() and not it['is_tpose']
	assert it['pose_descr'] == 'Strong Dorsiflex,Eversion'
	assert it['pose_code'].tolist() == [0, 1, 1, 0, 0, 0, 0, 0]                                  # 'Strong ' stripped; second entries are +1
	assert it['verts'].shape == (36, 3) and it['faces'].shape == (50, 3)                         # 5x5 quads fan-triangulated
	assert it['verts'].mean(0).abs().max() < 1e-6                                                # centred
	assert torch.equal(it['textures'].maps_padded()[0], torch.from_numpy(imgs['0005-B'].astype(np.float32) / 255))
	assert it['textures'].faces_uvs_padded().shape == (1, 50, 3) and it['textures'].verts_uvs_padded().shape == (1, 36, 2)
	assert tr[0]['has_keypoints'] and tr[0]['kp_idxs'].tolist() == [1, 2, 3]
	with pytest.raises(LookupError):
		from find_amd.dataset import get_pose_code
		get_pose_code(['Moonwalk'], cfg)


def test_right_feet_are_mirrored_and_caching_and_no_texture(foot3d):
	from find_amd.dataset import Foot3DDataset, NoTextureLoading, load_obj
	cfg, _ = foot3d
	ds = Foot3DDataset(cfg, device='cpu', left_only=False, full_caching=True)
	i = ds.scan_ids.index('A', 2) if ds.foot_ids[2] == '0007' else [n for n, f in enumerate(ds.foot_ids) if f == '0007'][0]
	raw, _, _ = load_obj(os.path.join(ds.folder, ds.data[i]['OBJ file']))
	want = raw.clone(); want[:, 1] = -want[:, 1]; want = want - want.mean(0)
	assert torch.allclose(ds[i]['verts'], want) and torch.allclose(ds[i]['verts'], want)          # second read comes from the cache, unflipped twice
	with NoTextureLoading(ds):
		assert ds[0]['textures'] is None
	assert ds[0]['textures'] is not None


def test_spatial_order_relabels_a_scan_without_changing_it(foot3d):
	"""Foot3DDataset(spatial_order=True) / dataset.spatial_order: the same surface in a spatially coherent memory order (the
	rasteriser's binning pass) -- the set of (position, uv) corners of every triangle, the orientation and the keypoints' positions are those
	of the file; the faces are sorted by their lowest vertex."""
	from find_amd.dataset import Foot3DDataset, ScanStore, spatial_order
	cfg, _ = foot3d
	ScanStore.shared.clear()
	plain = Foot3DDataset(cfg, device='cpu', is_train=True, train_and_val=True)
	moved = Foot3DDataset(cfg, device='cpu', is_train=True, train_and_val=True, spatial_order=True)
	assert len(plain) == len(moved) > 0
	for i in range(len(plain)):
		a, b = plain[i], moved[i]

		def corners(item):
			v, f = item['verts'].double(), item['faces']
			uv = item['textures'].verts_uvs_padded()[0].double()[item['textures'].faces_uvs_padded()[0]]   # (F, 3, 2)
			tri = torch.cat([v[f], uv], -1)                                                                  # (F, 3, 5)
			# a triangle up to rotation of its corners: start at the lexicographically smallest corner
			key = tri[..., 0] * 1e6 + tri[..., 1] * 1e3 + tri[..., 2]
			k = key.argmin(1)
			rows = torch.arange(tri.shape[0])
			tri = torch.stack([tri[rows, k], tri[rows, (k + 1) % 3], tri[rows, (k + 2) % 3]], 1).reshape(-1, 15)
			return tri[np.lexsort(tri.numpy().T[::-1])]

		assert torch.allclose(corners(a), corners(b), atol=1e-7)   # (the centring subtracts a mean formed in another order)
		fb = b['faces']
		assert bool((fb[:, 0] <= fb[:, 1]).all() and (fb[:, 0] <= fb[:, 2]).all()) and bool((fb[1:, 0] >= fb[:-1, 0]).all())
		if a['has_keypoints']:
			assert torch.allclose(a["verts"][torch.as_tensor(a["kp_idxs"])], b["verts"][torch.as_tensor(b["kp_idxs"])], atol=1e-7)
	# the utility alone: a permuted grid comes back with short runs
	g = torch.Generator().manual_seed(0)
	n = 24
	ii, jj = torch.meshgrid(torch.arange(n), torch.arange(n), indexing='ij')
	v = torch.stack([ii.flatten() * 0.01, jj.flatten() * 0.01, torch.zeros(n * n)], -1)
	idx = lambda i, j: i * n + j
	f = torch.tensor([[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)] for i in range(n - 1) for j in range(n - 1)])
	perm = torch.randperm(n * n, generator=g)
	inv = torch.empty_like(perm); inv[perm] = torch.arange(n * n)
	v2, f2, vertex_of, face_of, shift = spatial_order(v[perm], inv[f][torch.randperm(f.shape[0], generator=g)])
	cen = v2[f2].mean(1)
	run = [float((cen[k:k + 64].max(0).values - cen[k:k + 64].min(0).values).norm()) for k in range(0, f2.shape[0], 64)]
	fs = inv[f][torch.randperm(f.shape[0], generator=g)]
	cen0 = v[perm][fs].mean(1)
	run0 = [float((cen0[k:k + 64].max(0).values - cen0[k:k + 64].min(0).values).norm()) for k in range(0, fs.shape[0], 64)]
	assert sum(run) / len(run) < 0.5 * sum(run0) / len(run0)   # (a 64-face patch of this grid is ~0.08 across, a Z-order run ~0.15, a shuffled one the whole 0.33)


def test_collator_builds_ragged_meshes_with_joined_uv_textures(foot3d):
	from find_amd.dataset import BatchCollator, Foot3DDataset
	cfg, _ = foot3d
	ds = Foot3DDataset(cfg, device='cpu')
	batch = BatchCollator(device='cpu').collate_batches([ds[0], ds[1]])
	m = batch['mesh']
	assert len(m) == 2 and m.num_verts_per_mesh().tolist() == [25, 36] and m.num_faces_per_mesh().tolist() == [32, 50]
	assert m.verts_padded().shape == (2, 36, 3) and m.faces_padded().shape == (2, 50, 3) and (m.faces_padded()[0, 32:] == -1).all()
	t = m.textures
	assert t.maps_padded().shape == (2, 8, 6, 3) and t.faces_uvs_padded().shape == (2, 50, 3) and t.verts_uvs_padded().shape == (2, 36, 2)
	assert batch['name'] == ['0005-A', '0005-B'] and batch['idx'].tolist() == [0, 1] and batch['pose_code'].shape == (2, 8)
	assert batch['shape'] == ['0005', '0005'] and batch['reg'] == ['0005-A', '0005-B']


def test_template_average_colour_comes_from_the_uv_texture(tmp_path):
	"""NeuralDisplacementField(template_mesh_loc=...): avg_col is the mean of 1000 surface samples of the template's UV texture
	(reference model.py:266-277), not of per-vertex colours; without a texture use_avg_colour=True fails loudly."""
	from PIL import Image
	from find_amd.model import NeuralDisplacementField
	root = str(tmp_path)
	_write_scan(root, 'templ.obj', 'templ.png', 6, (0.0, 0.0, 0.0))
	kw = dict(device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=2, val_size=1, shapevec_size=8, texvec_size=8, posevec_size=8)
	# no .mtl yet: the OBJ names one that does not exist
	m0 = NeuralDisplacementField(template_mesh_loc=os.path.join(root, 'templ.obj'), **kw)
	assert m0.template_verts.shape == (1, 36, 3) and m0.template_verts.data.mean(dim=1).abs().max() < 1e-6   # centred
	assert float(m0.avg_col.abs().max()) == 0.0
	with pytest.raises(NotImplementedError):
		NeuralDisplacementField(template_mesh_loc=os.path.join(root, 'templ.obj'), use_avg_colour=True, **kw)
	# a uniform texture: every sample has that colour
	Image.fromarray(np.full((8, 6, 3), (51, 102, 204), np.uint8)).save(os.path.join(root, 'templ.png'))
	with open(os.path.join(root, 'none.mtl'), 'w') as fh:
		fh.write('newmtl material_0\nKd 1 1 1\nmap_Kd templ.png\n')
	m1 = NeuralDisplacementField(template_mesh_loc=os.path.join(root, 'templ.obj'), use_avg_colour=True, **kw)
	assert torch.allclose(m1.avg_col.data, torch.tensor([0.2, 0.4, 0.8]), atol=1e-6)
	# a left-to-right ramp in the red channel: the mean over a uniformly covered UV square is its midpoint
	ramp = np.zeros((8, 64, 3), np.uint8)
	ramp[..., 0] = np.linspace(0, 255, 64).astype(np.uint8)[None, :]
	Image.fromarray(ramp).save(os.path.join(root, 'templ.png'))
	torch.manual_seed(0)
	m2 = NeuralDisplacementField(template_mesh_loc=os.path.join(root, 'templ.obj'), **kw)
	assert abs(float(m2.avg_col[0]) - 0.5) < 0.05 and float(m2.avg_col[1:].abs().max()) == 0.0
