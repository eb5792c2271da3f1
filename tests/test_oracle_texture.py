"""oracle/texture_ref.py: the bilinear read is pinned against torch.nn.functional.grid_sample on the flipped map, exactly the
calls PyTorch3D's TexturesUV.sample_textures makes (flip [H axis], uv*2-1, align_corners=True, padding_mode='border')."""
import numpy as np
import torch

from oracle import texture_ref


def _scene(seed=0, Nm=2, H=13, W=9, Vt=11, F=7, P=50):
	g = np.random.RandomState(seed)
	maps = g.rand(Nm, H, W, 3).astype(np.float32)
	verts_uvs = (g.rand(Nm, Vt, 2) * 1.2 - 0.1).astype(np.float32)  # a few outside [0,1]: border padding
	faces_uvs = g.randint(0, Vt, (Nm, F, 3))
	face_idx = g.randint(-1, F, (Nm, P))
	b = g.rand(Nm, P, 3).astype(np.float32)
	b /= b.sum(-1, keepdims=True)
	return maps, verts_uvs, faces_uvs, face_idx, b


def test_uv_sample_matches_flip_plus_grid_sample():
	maps, vu, fu, fi, b = _scene()
	got = texture_ref.uv_sample(maps, vu, fu, fi, b)
	for m in range(maps.shape[0]):
		ok = fi[m] >= 0
		f = np.where(ok, fi[m], 0)
		uv = (b[m][:, :, None] * vu[m][fu[m][f]]).sum(1)
		grid = torch.from_numpy(uv * 2.0 - 1.0).float().view(1, 1, -1, 2)
		tex = torch.flip(torch.from_numpy(maps[m:m + 1]), [1]).permute(0, 3, 1, 2)  # (1,3,H,W), flipped along H
		ref = torch.nn.functional.grid_sample(tex, grid, mode='bilinear', align_corners=True, padding_mode='border')[0, :, 0].T.numpy()
		ref = np.where(ok[:, None], ref, 0.0)
		assert np.abs(got[m] - ref).max() < 2e-6


def test_texel_centres_and_corners_are_exact():
	H, W = 4, 5
	maps = np.arange(H * W * 3, dtype=np.float32).reshape(1, H, W, 3)
	vu = np.array([[[0, 0], [1, 0], [0, 1], [1, 1]]], np.float32)
	fu = np.array([[[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]]])
	fi = np.array([[0, 1, 2, 3]])
	b = np.tile(np.array([1, 0, 0], np.float32), (1, 4, 1))
	got = texture_ref.uv_sample(maps, vu, fu, fi, b)
	# (u,v) = (0,0) is the bottom-left texel of the image = last row of the array
	assert np.array_equal(got[0, 0], maps[0, H - 1, 0]) and np.array_equal(got[0, 1], maps[0, H - 1, W - 1])
	assert np.array_equal(got[0, 2], maps[0, 0, 0]) and np.array_equal(got[0, 3], maps[0, 0, W - 1])
