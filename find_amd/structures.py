"""Minimal mesh containers exposing the PyTorch3D method subset FIND's callers use:
verts_padded, faces_padded, verts_packed, faces_packed, num_verts_per_mesh, num_faces_per_mesh, update_padded,
extend, __len__, __getitem__, clone, textures(.verts_features_padded), device, to.

Enumerated from reference src/model/renderer.py:271-329, src/utils/pytorch3d_tools.py:9-36,
src/eval/eval_3d.py:143,168-171, src/train/trainer.py:274-276.  Storage is padded (N, Vmax, 3) / (N, Fmax, 3)
with per-mesh counts; faces padding is -1 as in PyTorch3D.  When every mesh shares one topology (the template),
`faces_shared` keeps a single (F,3) int32 tensor so the HIP kernels read the face list once."""
import torch


class TexturesVertex:
	"""Per-vertex colours (N, V, C) -- pytorch3d.renderer.TexturesVertex subset (model.py:497)."""

	def __init__(self, verts_features):
		if isinstance(verts_features, (list, tuple)):
			verts_features = _pad_list(verts_features, 0.0)[0]
		if verts_features.dim() != 3:
			raise ValueError('TexturesVertex expects (N, V, C) features')
		self._feat = verts_features

	def verts_features_padded(self):
		return self._feat

	def extend(self, M):
		return TexturesVertex(self._feat.repeat_interleave(M, dim=0))

	def clone(self):
		return TexturesVertex(self._feat.clone())

	def detach(self):
		return TexturesVertex(self._feat.detach())

	def to(self, device):
		return TexturesVertex(self._feat.to(device))

	def __getitem__(self, idx):
		f = self._feat[idx]
		return TexturesVertex(f.unsqueeze(0) if f.dim() == 2 else f)

	def __len__(self):
		return self._feat.shape[0]


class LazyTexturesVertex(TexturesVertex):
	"""Per-vertex colours that are evaluated when first read (find_amd.model.get_meshes(lazy_colours=True)): a 3-D-loss step never looks
	at the colours of the predicted mesh, a caller who does gets them as from TexturesVertex."""

	def __init__(self, thunk):
		self._thunk, self._value = thunk, None

	@property
	def _feat(self):
		if self._value is None:
			self._value = self._thunk()
			self._thunk = None
			if self._value.dim() != 3:
				raise ValueError('TexturesVertex expects (N, V, C) features')
		return self._value

	@property
	def evaluated(self):
		return self._value is not None


class TexturesUV:
	"""UV-mapped textures -- pytorch3d.renderer.TexturesUV subset used by the reference for GT scans (src/data/dataset.py:263-271):
	maps (N, H, W, 3), faces_uvs (N, F, 3) indices into verts_uvs (N, Vt, 2).  Sampling = find_amd.functional_render.uv_sample
	(flip + grid_sample(align_corners=True, padding_mode='border'), PyTorch3D's defaults)."""

	def __init__(self, maps, faces_uvs, verts_uvs):
		if isinstance(maps, (list, tuple)):
			maps = torch.stack(list(maps))
		if isinstance(faces_uvs, (list, tuple)):
			faces_uvs = _pad_list(list(faces_uvs), -1)[0]
		if isinstance(verts_uvs, (list, tuple)):
			verts_uvs = _pad_list(list(verts_uvs), 0.0)[0]
		if maps.dim() != 4 or maps.shape[-1] != 3 or faces_uvs.dim() != 3 or verts_uvs.dim() != 3 or verts_uvs.shape[-1] != 2:
			raise ValueError('TexturesUV expects maps (N,H,W,3), faces_uvs (N,F,3), verts_uvs (N,Vt,2)')
		if not (maps.shape[0] == faces_uvs.shape[0] == verts_uvs.shape[0]):
			raise ValueError('TexturesUV: batch sizes differ')
		self._maps, self._faces_uvs, self._verts_uvs = maps.float(), faces_uvs, verts_uvs.float()

	def maps_padded(self):
		return self._maps

	def faces_uvs_padded(self):
		return self._faces_uvs

	def verts_uvs_padded(self):
		return self._verts_uvs

	def extend(self, M):
		return TexturesUV(self._maps.repeat_interleave(M, dim=0), self._faces_uvs.repeat_interleave(M, dim=0), self._verts_uvs.repeat_interleave(M, dim=0))

	def clone(self):
		return TexturesUV(self._maps.clone(), self._faces_uvs.clone(), self._verts_uvs.clone())

	def detach(self):
		return TexturesUV(self._maps.detach(), self._faces_uvs, self._verts_uvs.detach())

	def to(self, device):
		return TexturesUV(self._maps.to(device), self._faces_uvs.to(device), self._verts_uvs.to(device))

	def __getitem__(self, idx):
		sl = (lambda t: t[idx].unsqueeze(0) if isinstance(idx, int) else t[idx])
		return TexturesUV(sl(self._maps), sl(self._faces_uvs), sl(self._verts_uvs))

	def __len__(self):
		return self._maps.shape[0]


def _pad_list(tensors, pad_value):
	n = max(int(t.shape[0]) for t in tensors)
	out = tensors[0].new_full((len(tensors), n) + tuple(tensors[0].shape[1:]), pad_value)
	lens = []
	for i, t in enumerate(tensors):
		out[i, :t.shape[0]] = t
		lens.append(int(t.shape[0]))
	return out, lens


class Meshes:
	def __init__(self, verts, faces, textures=None, _num_verts=None, _num_faces=None):
		"""verts: (N,V,3) tensor or list of (Vi,3); faces: (N,F,3) tensor, list of (Fi,3), or a single (F,3)
		tensor shared by every mesh."""
		if isinstance(verts, (list, tuple)):
			verts, nv = _pad_list([v.float() for v in verts], 0.0)
		else:
			nv = _num_verts if _num_verts is not None else [int(verts.shape[1])] * int(verts.shape[0])
		self._verts = verts
		self._num_verts = list(nv)
		N = verts.shape[0]
		self._faces_shared = None
		if isinstance(faces, (list, tuple)):
			faces, nf = _pad_list([f.long() for f in faces], -1)
		elif faces.dim() == 2:
			self._faces_shared = faces.to(torch.int32).contiguous()
			nf = [int(faces.shape[0])] * N
			faces = None
		else:
			nf = _num_faces if _num_faces is not None else [int(faces.shape[1])] * N
			if faces.shape[0] == 1 and N != 1:
				self._faces_shared = faces[0].to(torch.int32).contiguous()
				faces = None
		self._faces = faces
		self._num_faces = list(nf)
		self.textures = textures

	# ------------------------------------------------------------------ PyTorch3D-compatible accessors
	def __len__(self):
		return int(self._verts.shape[0])

	@property
	def device(self):
		return self._verts.device

	def verts_padded(self):
		return self._verts

	def faces_padded(self):
		if self._faces is None:
			return self._faces_shared.long().unsqueeze(0).expand(len(self), -1, -1)
		return self._faces

	def num_verts_per_mesh(self):
		return torch.tensor(self._num_verts, dtype=torch.int64, device=self.device)

	def num_faces_per_mesh(self):
		return torch.tensor(self._num_faces, dtype=torch.int64, device=self.device)

	def verts_list(self):
		return [self._verts[i, :n] for i, n in enumerate(self._num_verts)]

	def faces_list(self):
		fp = self.faces_padded()
		return [fp[i, :n] for i, n in enumerate(self._num_faces)]

	def verts_packed(self):
		if all(n == self._verts.shape[1] for n in self._num_verts):
			return self._verts.reshape(-1, 3)
		return torch.cat(self.verts_list(), dim=0)

	def faces_packed(self):
		"""Faces with vertex indices offset into verts_packed (PyTorch3D convention)."""
		offs, out, o = [], [], 0
		for n in self._num_verts:
			offs.append(o)
			o += n
		for f, off in zip(self.faces_list(), offs):
			out.append(f + off)
		return torch.cat(out, dim=0)

	def is_homogeneous(self):
		return len(set(self._num_verts)) == 1 and len(set(self._num_faces)) == 1

	def faces_shared(self):
		"""(F,3) int32 face list if every mesh has the same topology tensor, else None."""
		return self._faces_shared

	def update_padded(self, new_verts_padded):
		if new_verts_padded.shape != self._verts.shape:
			raise ValueError(f'update_padded: shape {tuple(new_verts_padded.shape)} != {tuple(self._verts.shape)}')
		m = Meshes.__new__(Meshes)
		m.__dict__.update(self.__dict__)
		m._verts = new_verts_padded
		return m

	def extend(self, M):
		"""Each mesh repeated M times consecutively (renderer.py:271)."""
		tex = self.textures.extend(M) if self.textures is not None else None
		verts = self._verts.repeat_interleave(M, dim=0)
		nv = [n for n in self._num_verts for _ in range(M)]
		nf = [n for n in self._num_faces for _ in range(M)]
		if self._faces_shared is not None:
			return Meshes(verts, self._faces_shared, tex, _num_verts=nv)
		return Meshes(verts, self._faces.repeat_interleave(M, dim=0), tex, _num_verts=nv, _num_faces=nf)

	def clone(self):
		tex = self.textures.clone() if self.textures is not None else None
		faces = self._faces_shared.clone() if self._faces_shared is not None else self._faces.clone()
		return Meshes(self._verts.clone(), faces, tex, _num_verts=list(self._num_verts),
					  _num_faces=None if self._faces_shared is not None else list(self._num_faces))

	def detach(self):
		m = self.update_padded(self._verts.detach())
		if m.textures is not None and hasattr(m.textures, 'detach'):
			m.textures = m.textures.detach()
		return m

	def to(self, device):
		tex = self.textures.to(device) if self.textures is not None else None
		faces = self._faces_shared.to(device) if self._faces_shared is not None else self._faces.to(device)
		return Meshes(self._verts.to(device), faces, tex, _num_verts=list(self._num_verts),
					  _num_faces=None if self._faces_shared is not None else list(self._num_faces))

	def __getitem__(self, idx):
		if isinstance(idx, int):
			idx = [idx]
		elif isinstance(idx, slice):
			idx = list(range(len(self)))[idx]
		elif torch.is_tensor(idx):
			idx = idx.tolist()
		verts = self._verts[idx]
		nv = [self._num_verts[i] for i in idx]
		tex = None
		if self.textures is not None:
			tex = self.textures[idx]
		if self._faces_shared is not None:
			return Meshes(verts, self._faces_shared, tex, _num_verts=nv)
		return Meshes(verts, self._faces[idx], tex, _num_verts=nv, _num_faces=[self._num_faces[i] for i in idx])


def join_meshes_as_batch(meshes):
	"""pytorch3d.structures.join_meshes_as_batch subset used by eval_3d.py:130,143."""
	verts, faces, feats = [], [], []
	for m in meshes:
		verts += m.verts_list()
		faces += m.faces_list()
		if m.textures is not None:
			f = m.textures.verts_features_padded()
			feats += [f[i, :n] for i, n in enumerate(m._num_verts)]
	tex = TexturesVertex(feats) if len(feats) == len(verts) and feats else None
	return Meshes(verts, faces, tex)


def extend_template(meshes, N=1):
	"""One template -> N meshes without copying vertices (pytorch3d_tools.py:7-16)."""
	verts = meshes.verts_padded().expand(N, -1, -1)
	fs = meshes.faces_shared()
	faces = fs if fs is not None else meshes.faces_padded()[0]
	tex = None
	if meshes.textures is not None:
		tex = TexturesVertex(meshes.textures.verts_features_padded().expand(N, -1, -1))
	return Meshes(verts, faces, tex)
