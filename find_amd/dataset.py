"""Foot3D scans on disk -> the `batch` dict the hot path consumes.

Behavioural spec, taken from what the callers of reference src/data/dataset.py rely on (train.py:118-131, trainer.py:97-100,
eval_3d.py:60-76) rather than from its text:

  on-disk format   one JSON index {"keypoint_labels": [...], "data": [record, ...]}; a record names a scan by 'Foot ID' + 'Scan ID'
                   and carries 'footedness' ('Left' / 'Right'), 'pose' (list of descriptions), 'keypoints' (vertex indices or null),
                   'OBJ file', 'PNG file' (paths below <DATASET_FOLDER>/<DATASET_NAME>, or <LOWPOLY_DATASET_NAME> for low-poly meshes)
  selection        a fixed cascade of record filters (SELECTION below), then an optional head(N)
  item             geometry centred on its centroid (right feet mirrored in y first), UV texture unless texture loading is switched
                   off, bookkeeping fields, and the four latent-table keys: shape / tex per foot, pose / reg per scan
  batch            default_collate of the non-geometry fields + one ragged `Meshes` with a joined `TexturesUV`

The configuration the reference reads from src/cfg.yaml is passed in as a dict (DATASET_FOLDER, DATASET_JSON, DATASET_NAME,
LOWPOLY_DATASET_NAME, VAL_FEET, TEMPLATE_FEET, POSE_VECTOR).  PyTorch3D's OBJ reader / cv2 are replaced by a small parser and PIL.
Public names and keywords are the reference's, so src/train and src/eval construct it unchanged (plus the leading cfg)."""
import json
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data._utils.collate import default_collate

from .structures import Meshes, TexturesUV

GEOMETRY_FIELDS = ('verts', 'faces', 'textures')


# ----------------------------------------------------------------------------------------------- files
@dataclass
class ObjFaces:
	verts_idx: torch.Tensor      # (F, 3) int64
	textures_idx: torch.Tensor   # (F, 3) int64, -1 where a corner has no texture coordinate


@dataclass
class ObjProps:
	verts_uvs: torch.Tensor      # (Vt, 2) float32


def _resolve(index, count):
	"""OBJ indices are 1-based; negative ones count back from the elements read so far."""
	return index - 1 if index > 0 else count + index


def load_obj(loc, device='cpu'):
	"""(verts (V,3) float32, ObjFaces, ObjProps) of a Wavefront OBJ: the three results of pytorch3d.io.load_obj the reference uses.
	Polygons are split into a triangle fan; material statements are skipped."""
	positions, uvs, tri_v, tri_t = [], [], [], []
	with open(loc, 'r') as fh:
		for raw in fh:
			tag, _, rest = raw.partition(' ')
			if tag == 'v':
				positions.append([float(x) for x in rest.split()[:3]])
			elif tag == 'vt':
				uvs.append([float(x) for x in rest.split()[:2]])
			elif tag == 'f':
				vi, ti = [], []
				for corner in rest.split():
					fields = corner.split('/')
					vi.append(_resolve(int(fields[0]), len(positions)))
					ti.append(_resolve(int(fields[1]), len(uvs)) if len(fields) > 1 and fields[1] else -1)
				for k in range(2, len(vi)):
					tri_v.append((vi[0], vi[k - 1], vi[k]))
					tri_t.append((ti[0], ti[k - 1], ti[k]))

	def tensor(rows, dtype, width):
		return torch.from_numpy(np.asarray(rows, dtype=dtype).reshape(-1, width)).to(device)

	return tensor(positions, np.float32, 3), ObjFaces(tensor(tri_v, np.int64, 3), tensor(tri_t, np.int64, 3)), ObjProps(tensor(uvs, np.float32, 2))


def load_texture_png(loc):
	"""RGB image as (H, W, 3) float32 in [0, 1]."""
	from PIL import Image
	with Image.open(loc) as im:
		return torch.from_numpy(np.asarray(im.convert('RGB'), dtype=np.float32) / 255.0)


# ----------------------------------------------------------------------------------------------- pose descriptions
def _pose_slots(cfg):
	"""cfg['POSE_VECTOR'] = {'SIZE': n, i: [name] or [negative name, positive name]}  ->  {name: (slot, value)}."""
	table = cfg['POSE_VECTOR']
	slots = {}
	for i in range(table['SIZE']):
		names = table[i]
		if len(names) not in (1, 2):
			raise ValueError(f'lookup for pose element {i} is not 1 or 2 long.')
		values = (1.0,) if len(names) == 1 else (-1.0, 1.0)
		for name, value in zip(names, values):
			slots.setdefault(name, (i, value))
	return table['SIZE'], slots


def get_pose_code(pose_list, cfg):
	"""Pose descriptions -> signed indicator vector ('Strong ' prefixes are ignored; an unknown description is a LookupError)."""
	size, slots = _pose_slots(cfg)
	vec = np.zeros(size)
	for descr in pose_list:
		key = descr.replace('Strong ', '')
		if key not in slots:
			raise LookupError(f'Pose {key} not found in lookup.')
		slot, value = slots[key]
		vec[slot] = value
	return vec


# ----------------------------------------------------------------------------------------------- selection
# Record filters in the order they apply.  Each entry: (name, enabled(options) -> bool, keep(record, options, cfg) -> bool).
SELECTION = (
	('template feet are held out unless feet are requested by ID',
	 lambda o: o['specific_feet'] is None, lambda r, o, c: r['Foot ID'] not in c['TEMPLATE_FEET']),
	('train / validation split by foot',
	 lambda o: not (o['train_and_val'] or o['specific_feet']), lambda r, o, c: (r['Foot ID'] in c['VAL_FEET']) != o['is_train']),
	('T-pose scans only',
	 lambda o: o['tpose_only'], lambda r, o, c: 'T-Pose' in r.get('pose', [])),
	('left feet only',
	 lambda o: o['left_only'], lambda r, o, c: r.get('footedness') == 'Left'),
	('feet requested by ID',
	 lambda o: bool(o['specific_feet']), lambda r, o, c: r['Foot ID'] in o['specific_feet']),
)


def select_records(records, options, cfg):
	for _, enabled, keep in SELECTION:
		if enabled(options):
			records = [r for r in records if keep(r, options, cfg)]
	return records


def scan_name(record):
	return f"{record['Foot ID']}-{record['Scan ID']}"


# ----------------------------------------------------------------------------------------------- scan store
def spatial_order(verts, faces):
	"""A relabelling of one mesh that puts neighbours in space next to each other in memory -- vertices by the Morton code of their position
	(10 bits per axis of the bounding box), faces by their lowest vertex, every face rotated (not flipped) so that this vertex comes first.
	Not in the reference: the rasteriser's binning pass culls with one bounding box per run of 64 consecutive faces and is up to 4 x slower on
	a mesh whose order is incoherent.  verts (V, 3) float, faces (F, 3) int -> (verts', faces', vertex_of, face_of):
	verts' = verts[vertex_of], faces'[i] is face face_of[i] of the input in the new labels, rotated by `shift[i]` = the third extra return.
	Geometry, orientation and the triangle set are unchanged; per-face data follows through face_of (+ shift for per-corner data),
	per-vertex indices (keypoints) through the inverse of vertex_of."""
	v = verts.detach().cpu().double().numpy()
	f = faces.detach().cpu().numpy().astype(np.int64)
	lo, hi = v.min(0), v.max(0)
	q = np.clip(((v - lo) / np.maximum(hi - lo, 1e-30) * 1023.0).astype(np.int64), 0, 1023)
	code = np.zeros(v.shape[0], np.int64)
	for bit in range(10):
		for ax in range(3):
			code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
	vertex_of = np.argsort(code, kind='stable')
	new_label = np.empty_like(vertex_of)
	new_label[vertex_of] = np.arange(v.shape[0])
	g = new_label[f]
	shift = g.argmin(1)
	rows = np.arange(g.shape[0])
	g = np.stack([g[rows, shift], g[rows, (shift + 1) % 3], g[rows, (shift + 2) % 3]], -1)
	face_of = np.lexsort((g[:, 2], g[:, 1], g[:, 0]))
	dev = verts.device
	return (verts[torch.from_numpy(vertex_of).to(dev)], torch.from_numpy(g[face_of]).to(device=faces.device, dtype=faces.dtype),
			torch.from_numpy(vertex_of).to(dev), torch.from_numpy(face_of).to(dev), torch.from_numpy(shift[face_of]).to(dev))


@dataclass
class Scan:
	verts: torch.Tensor
	face_dict: ObjFaces
	props: ObjProps
	tex_img: Optional[torch.Tensor]


class ScanStore:
	"""Reads scans from disk; with `keep` it remembers them by name (the reference's full_caching: a process-wide store, so train and
	validation datasets share it).  A scan remembered without its texture is read again when the texture is first asked for."""
	shared = {}

	def __init__(self, keep, reorder=False):
		self.keep = keep
		self.reorder = reorder

	def fetch(self, name, obj_loc, png_loc, want_texture, device):
		scan = self.shared.get(name) if self.keep else None
		if scan is None or (want_texture and scan.tex_img is None):
			verts, faces, props = load_obj(obj_loc, device=device)
			label = None
			if self.reorder:   # (spatial_order: same surface, coherent memory order; per-face UV indices and vertex labels follow)
				verts, fv, vertex_of, face_of, shift = spatial_order(verts, faces.verts_idx)
				ft = faces.textures_idx
				if ft is not None:
					ft = ft[face_of.to(ft.device)]
					sh = shift.to(ft.device)
					rows = torch.arange(ft.shape[0], device=ft.device)
					ft = torch.stack([ft[rows, sh], ft[rows, (sh + 1) % 3], ft[rows, (sh + 2) % 3]], -1)
				faces = ObjFaces(verts_idx=fv, textures_idx=ft)
				label = torch.empty_like(vertex_of)
				label[vertex_of] = torch.arange(vertex_of.shape[0], device=vertex_of.device)
			scan = Scan(verts, faces, props, load_texture_png(png_loc) if want_texture else None)
			scan.new_label = label   # file vertex index -> index in `verts` (None: unchanged)
			if self.keep:
				self.shared[name] = scan
		return scan


_cache = ScanStore.shared  # the name the reference exposes for its cache


class NoTextureLoading:
	"""`with NoTextureLoading(train_set, val_set): ...` -- items come without textures inside the block (registration stage)."""

	def __init__(self, *datasets):
		self.datasets = datasets
		self.states = []

	def __enter__(self, *args):
		self.states = [d._load_texture for d in self.datasets]
		for d in self.datasets:
			d._load_texture = False

	def __exit__(self, *args):
		for d, state in zip(self.datasets, self.states):
			d._load_texture = state


# ----------------------------------------------------------------------------------------------- dataset
class Foot3DDataset(Dataset):
	def __init__(self, cfg, dataset_json=None, N=None, tpose_only=False, left_only=True, specific_feet=None, full_caching=False, is_train=True,
				 train_and_val=False, device='cuda', low_res_textures=False, low_poly_meshes=False, spatial_order=False):
		"""spatial_order (not in the reference): relabel every scan into a spatially coherent vertex / face order when it is read
		(dataset.spatial_order): the same surfaces, keypoint indices and UV faces follow -- but the surface sampler then maps a given
		uniform draw to another face than the reference would for the same seed (the face order is the order of its area sums)."""
		super().__init__()
		self.cfg = cfg
		self.folder = os.path.join(cfg['DATASET_FOLDER'], cfg['LOWPOLY_DATASET_NAME'] if low_poly_meshes else cfg['DATASET_NAME'])
		with open(dataset_json or cfg['DATASET_JSON']) as fh:
			index = json.load(fh)
		self.meta = {k: v for k, v in index.items() if k != 'data'}
		options = dict(specific_feet=specific_feet, train_and_val=train_and_val, is_train=is_train, tpose_only=tpose_only, left_only=left_only)
		self.data = select_records(index['data'], options, cfg)
		if specific_feet:
			assert len(self.data) > 0, f'No feet found with IDs `{specific_feet}`.'
		if N is not None:
			self.data = self.data[:N]
		self.is_train = is_train
		self.full_caching = full_caching
		self.keypoint_labels = self.meta['keypoint_labels']
		self.nkeypoints = len(self.keypoint_labels)
		self.device = device
		self.low_res_textures = low_res_textures
		self._load_texture = True
		self._store = ScanStore(keep=full_caching, reorder=spatial_order)

	def __len__(self):
		return len(self.data)

	@property
	def foot_ids(self):
		return [r['Foot ID'] for r in self.data]

	@property
	def scan_ids(self):
		return [r['Scan ID'] for r in self.data]

	# ---- latent-table keys: what is shared between the scans of one foot, what is not
	def get_keys(self, idx):
		record = self.data[idx]
		per_foot, per_scan = record['Foot ID'], scan_name(record)
		return {'shape': per_foot, 'pose': per_scan, 'tex': per_foot, 'reg': per_scan}

	def get_all_keys(self):
		"""{'shape': [...], 'pose': [...], 'tex': [...], 'reg': [...]}: distinct keys in dataset order (the rows of the latent tables)."""
		seen = {}
		for i in range(len(self)):
			for table, key in self.get_keys(i).items():
				seen.setdefault(table, {}).setdefault(key, None)
		return {table: list(keys) for table, keys in seen.items()}

	def _find(self, model_id):
		return [i for i, r in enumerate(self.data) if scan_name(r) == model_id]

	def get_pose_from_model_id(self, model_id):
		hits = self._find(model_id)
		if not hits:
			raise LookupError(f'model_id {model_id} not found in dataset.')
		return self.data[hits[0]]['pose']

	def get_by_id(self, ID):
		hits = self._find(ID)
		assert len(hits) == 1, f'{len(hits)} matches found for ID {ID}.'
		return self[hits[0]]

	def __getitem__(self, idx):
		record = self.data[idx]
		name = scan_name(record)
		png = record['PNG file'].replace('.png', '_1k.png') if self.low_res_textures else record['PNG file']
		scan = self._store.fetch(name, os.path.join(self.folder, record['OBJ file']), os.path.join(self.folder, png), self._load_texture, self.device)
		textures = None
		if self._load_texture:
			if scan.tex_img is None:
				raise ValueError(f'Could not load texture - {name}.')
			textures = TexturesUV(scan.tex_img[None].to(self.device), faces_uvs=scan.face_dict.textures_idx[None].to(self.device),
								  verts_uvs=scan.props.verts_uvs[None].to(self.device))
		verts = scan.verts
		if record['footedness'] == 'Right':   # right feet are mirrored onto left ones (on a copy: the store keeps the file's geometry)
			verts = verts * verts.new_tensor([1.0, -1.0, 1.0])
		verts = verts - verts.mean(dim=0)
		kps = record.get('keypoints')
		if kps is not None and getattr(scan, 'new_label', None) is not None:
			kps = scan.new_label.cpu().numpy()[np.asarray(kps, dtype=np.int64)].tolist()
		item = dict(faces=scan.face_dict.verts_idx, verts=verts, textures=textures, idx=idx, name=name,
					has_keypoints=kps is not None, kp_idxs=np.array(kps) if kps is not None else np.zeros(self.nkeypoints),
					is_tpose='T-Pose' in record.get('pose', []), orig_footedness=record['footedness'], pose_descr=','.join(record['pose']),
					pose_code=get_pose_code(record['pose'], self.cfg))
		item.update(self.get_keys(idx))
		return item


# ----------------------------------------------------------------------------------------------- collation
def join_textures_uv(textures):
	"""Batch-1 TexturesUV objects of one map size -> one TexturesUV (faces_uvs / verts_uvs padded per mesh)."""
	per_mesh = [(t.maps_padded()[i], t.faces_uvs_padded()[i], t.verts_uvs_padded()[i]) for t in textures for i in range(len(t))]
	maps, faces_uvs, verts_uvs = zip(*per_mesh)
	return TexturesUV(torch.stack(maps), list(faces_uvs), list(verts_uvs))


def collate_batched_meshes(batch):
	"""Items -> ragged Meshes (None for an empty batch or items without geometry)."""
	if not batch or not {'verts', 'faces'} <= set(batch[0]):
		return None
	textures = [item.get('textures') for item in batch]
	joined = join_textures_uv(textures) if textures[0] is not None else None
	return Meshes(verts=[item['verts'] for item in batch], faces=[item['faces'] for item in batch], textures=joined)


class BatchCollator:
	"""collate_fn of the reference's DataLoaders: default_collate for everything but the geometry, which becomes batch['mesh']."""

	def __init__(self, device='cuda'):
		self.device = device

	def collate_batches(self, batch):
		out = default_collate([{k: v for k, v in item.items() if k not in GEOMETRY_FIELDS} for item in batch])
		out['mesh'] = collate_batched_meshes(batch).to(self.device)
		return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in out.items()}
