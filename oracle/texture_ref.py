"""CPU restatement of pytorch3d.renderer.TexturesUV.sample_textures as the reference uses it on GT scans
(src/data/dataset.py:263-271, src/model/losses.py:39-43, src/model/renderer.py:329-346).  TEST INFRASTRUCTURE ONLY.

PyTorch3D @ 1706eb82 (requirements_mac_linux.txt:31, not vendored): pixel_uvs = barycentric mix of the face's three UV vertices;
`pixel_uvs * 2 - 1`; the map is flipped vertically (`torch.flip(texture_maps, [2])`); F.grid_sample(mode='bilinear',
align_corners=True, padding_mode='border').  The bilinear read itself is pinned against torch.nn.functional.grid_sample
(tests/test_oracle_texture.py); the call sequence around it is PyTorch3D's published one ("parity unpinned" for that part,
like the other PyTorch3D-backed rows)."""
import numpy as np


def uv_sample(maps, verts_uvs, faces_uvs, face_idx, bary):
	"""maps (Nm,H,W,3), verts_uvs (Nm,Vt,2), faces_uvs (Nm,F,3) int, face_idx (R,P) (-1 -> zeros), bary (R,P,3); R multiple of Nm
	with the rows of one map consecutive.  Returns (R,P,3) float32."""
	maps = np.asarray(maps, np.float32); verts_uvs = np.asarray(verts_uvs, np.float32); bary = np.asarray(bary, np.float32)
	faces_uvs = np.asarray(faces_uvs); face_idx = np.asarray(face_idx)
	Nm, H, W, _ = maps.shape
	R, P = face_idx.shape
	out = np.zeros((R, P, 3), np.float32)
	for r in range(R):
		m = r // (R // Nm)
		fu = faces_uvs[m if faces_uvs.shape[0] > 1 else 0]
		ok = face_idx[r] >= 0
		f = np.where(ok, face_idx[r], 0)
		uv = (bary[r][:, :, None] * verts_uvs[m][fu[f]]).sum(1)          # (P,2)
		x = np.clip(uv[:, 0] * np.float32(W - 1), 0, W - 1).astype(np.float32)
		y = np.clip((np.float32(1.0) - uv[:, 1]) * np.float32(H - 1), 0, H - 1).astype(np.float32)
		x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64)
		x1 = np.minimum(x0 + 1, W - 1); y1 = np.minimum(y0 + 1, H - 1)
		tx = (x - x0)[:, None].astype(np.float32); ty = (y - y0)[:, None].astype(np.float32)
		mp = maps[m]
		c = (1 - tx) * (1 - ty) * mp[y0, x0] + tx * (1 - ty) * mp[y0, x1] + (1 - tx) * ty * mp[y1, x0] + tx * ty * mp[y1, x1]
		out[r] = np.where(ok[:, None], c, 0.0)
	return out
