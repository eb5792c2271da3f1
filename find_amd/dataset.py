"""Foot3D dataset reader and collator: the host-side producer of the `batch` dict the hot path consumes.
Mirror of reference src/data/dataset.py -- Foot3DDataset (:116-299: JSON index -> OBJ + PNG -> centred vertices, TexturesUV),
collate_batched_meshes (:28-52), BatchCollator (:55-67), NoTextureLoading (:70-85), get_pose_code (:88-109) -- with the same
constructor keywords, filters, item keys and quirks; PyTorch3D's OBJ loader / trimesh / cv2 are replaced by a small numpy OBJ
parser and PIL.  The configuration (`src/cfg.yaml` in the reference) is passed in as a dict: DATASET_FOLDER, DATASET_JSON,
DATASET_NAME, LOWPOLY_DATASET_NAME, VAL_FEET, TEMPLATE_FEET, POSE_VECTOR."""
import json
import os
from collections import defaultdict, namedtuple

import numpy as np
import torch
from torch.utils.data import Dataset, _utils

from .structures import Meshes, TexturesUV

ObjFaces = namedtuple('ObjFaces', 'verts_idx textures_idx')
ObjProps = namedtuple('ObjProps', 'verts_uvs')
CachedMesh = namedtuple('CachedMesh', 'verts face_dict props tex_img')
_cache = {}  # external cache used when full_caching is on (dataset.py:112)


def load_obj(loc, device='cpu'):
	"""Vertices, faces and UVs of a Wavefront OBJ: returns (verts (V,3) float32, ObjFaces(verts_idx (F,3), textures_idx (F,3) int64,
	-1 where a corner has no vt), ObjProps(verts_uvs (Vt,2))) -- the fields of pytorch3d.io.load_obj the reference reads
	(dataset.py:238, 266-267).  Polygons are fan-triangulated; negative (relative) indices are resolved; materials are ignored."""
	vs, vts, fv, ft = [], [], [], []
	with open(loc, 'r') as fh:
		for line in fh:
			if line.startswith('v '):
				p = line.split()
				vs.append((float(p[1]), float(p[2]), float(p[3])))
			elif line.startswith('vt '):
				p = line.split()
				vts.append((float(p[1]), float(p[2])))
			elif line.startswith('f '):
				corners = []
				for tok in line.split()[1:]:
					parts = tok.split('/')
					vi = int(parts[0])
					ti = int(parts[1]) if len(parts) > 1 and parts[1] != '' else 0
					corners.append((vi - 1 if vi > 0 else len(vs) + vi, (ti - 1 if ti > 0 else len(vts) + ti) if ti != 0 else -1))
				for k in range(1, len(corners) - 1):
					tri = (corners[0], corners[k], corners[k + 1])
					fv.append([c[0] for c in tri])
					ft.append([c[1] for c in tri])
	verts = torch.tensor(np.asarray(vs, dtype=np.float32).reshape(-1, 3), device=device)
	faces = ObjFaces(torch.tensor(np.asarray(fv, dtype=np.int64).reshape(-1, 3), device=device),
					 torch.tensor(np.asarray(ft, dtype=np.int64).reshape(-1, 3), device=device))
	props = ObjProps(torch.tensor(np.asarray(vts, dtype=np.float32).reshape(-1, 2), device=device))
	return verts, faces, props


def load_texture_png(loc):
	"""(H, W, 3) float32 in [0,1], RGB (the reference: cv2.imread + BGR2RGB, / 255; dataset.py:255-256)."""
	from PIL import Image
	with Image.open(loc) as im:
		return torch.from_numpy(np.asarray(im.convert('RGB'), dtype=np.float32) / 255.0)


def join_textures_uv(texlist):
	"""TexturesUV.join_batch for maps of one size (dataset.py:43): faces_uvs / verts_uvs are padded per mesh."""
	maps = torch.cat([t.maps_padded() for t in texlist], dim=0)
	fu = [t.faces_uvs_padded()[i] for t in texlist for i in range(len(t))]
	vu = [t.verts_uvs_padded()[i] for t in texlist for i in range(len(t))]
	return TexturesUV(maps, fu, vu)


def collate_batched_meshes(batch):
	"""List of items -> one Meshes with padded, ragged geometry and joined textures (dataset.py:28-52)."""
	if batch is None or len(batch) == 0:
		return None
	col = {k: [d[k] for d in batch] for k in batch[0].keys()}
	if not {'verts', 'faces'}.issubset(col.keys()):
		return None
	textures = None
	if 'textures' in col and col['textures'][0] is not None:
		textures = join_textures_uv(col['textures'])
	return Meshes(verts=col['verts'], faces=col['faces'], textures=textures)


class BatchCollator:
	def __init__(self, device='cuda'):
		self.device = device

	def collate_batches(self, batch):
		non_mesh = [{k: v for k, v in e.items() if k not in ['verts', 'faces', 'textures']} for e in batch]
		out = _utils.collate.default_collate(non_mesh)
		out['mesh'] = collate_batched_meshes(batch).to(self.device)
		for k, v in out.items():
			if torch.is_tensor(v):
				out[k] = v.to(self.device)
		return out


class NoTextureLoading:
	"""Context manager turning texture loading off, e.g. during registration (dataset.py:70-85)."""

	def __init__(self, *datasets):
		self.datasets = datasets
		self.states = []

	def __enter__(self, *args):
		for d in self.datasets:
			self.states.append(d._load_texture)
			d._load_texture = False

	def __exit__(self, *args):
		for n, d in enumerate(self.datasets):
			d._load_texture = self.states[n]


def get_pose_code(pose_list, cfg):
	"""List of pose descriptions -> vector, through cfg['POSE_VECTOR'] (dataset.py:88-109)."""
	lookup = cfg['POSE_VECTOR']
	N = lookup['SIZE']
	vec = np.zeros(N)
	for p in pose_list:
		p = p.replace('Strong ', '')
		for i in range(N):
			if p in lookup[i]:
				if len(lookup[i]) == 1:
					vec[i] = 1
				elif len(lookup[i]) == 2:
					vec[i] = [-1, 1][lookup[i].index(p)]
				else:
					raise ValueError(f'lookup for pose element {i} is not 1 or 2 long.')
				break
		else:
			raise LookupError(f'Pose {p} not found in lookup.')
	return vec


class Foot3DDataset(Dataset):
	def __init__(self, cfg, dataset_json=None, N=None, tpose_only=False, left_only=True, specific_feet=None, full_caching=False, is_train=True,
				 train_and_val=False, device='cuda', low_res_textures=False, low_poly_meshes=False):
		super().__init__()
		self.cfg = cfg
		dataset_json = dataset_json if dataset_json is not None else cfg['DATASET_JSON']
		self.folder = os.path.join(cfg['DATASET_FOLDER'], cfg['DATASET_NAME'] if not low_poly_meshes else cfg['LOWPOLY_DATASET_NAME'])
		with open(dataset_json) as fh:
			data = json.load(fh)
		self.meta = {k: v for k, v in data.items() if k != 'data'}
		self.data = data['data']
		# the template foot is skipped unless feet are named explicitly (dataset.py:150)
		self.data = [d for d in self.data if d['Foot ID'] not in cfg['TEMPLATE_FEET'] or specific_feet is not None]
		self.is_train = is_train
		if not (train_and_val or specific_feet):
			if is_train:
				self.data = [d for d in self.data if d['Foot ID'] not in cfg['VAL_FEET']]
			else:
				self.data = [d for d in self.data if d['Foot ID'] in cfg['VAL_FEET']]
		if tpose_only:
			self.data = [d for d in self.data if 'T-Pose' in d.get('pose', [])]
		if left_only:
			self.data = [d for d in self.data if d.get('footedness', None) == 'Left']
		if specific_feet:
			self.data = [d for d in self.data if d['Foot ID'] in specific_feet]
			assert len(self.data) > 0, f'No feet found with IDs `{specific_feet}`.'
		if N is not None:
			self.data = self.data[:N]
		self.full_caching = full_caching
		self.keypoint_labels = self.meta['keypoint_labels']
		self.nkeypoints = len(self.keypoint_labels)
		self.device = device
		self._load_texture = True
		self.low_res_textures = low_res_textures

	def __len__(self):
		return len(self.data)

	@property
	def foot_ids(self):
		return [ann['Foot ID'] for ann in self.data]

	@property
	def scan_ids(self):
		return [ann['Scan ID'] for ann in self.data]

	def get_keys(self, idx):
		"""Latent-table keys of an item: shape / texture shared by the scans of one foot, pose / registration per scan (dataset.py:190-200)."""
		ann = self.data[idx]
		name = f"{ann['Foot ID']}-{ann['Scan ID']}"
		return {'shape': ann['Foot ID'], 'pose': name, 'tex': ann['Foot ID'], 'reg': name}

	def get_all_keys(self):
		out = defaultdict(list)
		for i in range(len(self)):
			for k, v in self.get_keys(i).items():
				if v not in out[k]:
					out[k].append(v)
		return out

	def get_pose_from_model_id(self, model_id):
		foot_id, scan_id = model_id.split('-')
		for ann in self.data:
			if ann['Foot ID'] == foot_id and ann['Scan ID'] == scan_id:
				return ann['pose']
		raise LookupError(f'model_id {model_id} not found in dataset.')

	def get_by_id(self, ID):
		idx = [n for n, ann in enumerate(self.data) if ID == f"{ann['Foot ID']}-{ann['Scan ID']}"]
		assert len(idx) == 1, f'{len(idx)} matches found for ID {ID}.'
		return self[idx[0]]

	def __getitem__(self, idx):
		ann = self.data[idx]
		name = f"{ann['Foot ID']}-{ann['Scan ID']}"
		obj_loc = os.path.join(self.folder, ann['OBJ file'])
		tex_loc = os.path.join(self.folder, ann['PNG file'])
		load_from_cache = self.full_caching and name in _cache
		cached = None
		if load_from_cache:
			cached = _cache[name]
			if self._load_texture and cached.tex_img is None:
				load_from_cache = False  # the texture was not needed when this scan was cached: reload
		if not load_from_cache:
			verts, face_dict, props = load_obj(obj_loc, device=self.device)
			tex_img = None
			if self._load_texture:
				if self.low_res_textures:
					tex_loc = tex_loc.replace('.png', '_1k.png')
				tex_img = load_texture_png(tex_loc)
			cached = CachedMesh(verts, face_dict, props, tex_img)
			if self.full_caching:
				_cache[name] = cached
		tex_map = None
		if self._load_texture:
			if cached.tex_img is None:
				raise ValueError(f'Could not load texture - {name}.')
			tex_map = TexturesUV(cached.tex_img.unsqueeze(0).to(self.device), faces_uvs=cached.face_dict.textures_idx.unsqueeze(0).to(self.device),
								 verts_uvs=cached.props.verts_uvs.unsqueeze(0).to(self.device))
		verts, face_dict = cached.verts, cached.face_dict
		if ann['footedness'] == 'Right':
			verts = verts.clone()
			verts[..., 1] = -verts[..., 1]  # mirror right feet onto left ones (dataset.py:277-279; here without touching the cache)
		verts = verts - torch.mean(verts, dim=0)  # centroid to the origin
		has_keypoints = ann.get('keypoints') is not None
		keypoints = np.array(ann['keypoints']) if has_keypoints else np.zeros(self.nkeypoints)
		return {'faces': face_dict.verts_idx, 'verts': verts, 'textures': tex_map, 'idx': idx, 'name': name, 'has_keypoints': has_keypoints,
				'kp_idxs': keypoints, 'is_tpose': 'T-Pose' in ann.get('pose', []), 'orig_footedness': ann['footedness'],
				'pose_descr': ','.join(ann['pose']), 'pose_code': get_pose_code(ann['pose'], self.cfg), **self.get_keys(idx)}
