"""Which term of the composition fixture's 'net' case carries the GPU's largest gradient deviation (against a float64 oracle run)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_train3d import FixedDraws
from find_amd.model_with_loss import ModelWithLoss
from find_amd.opts import Opts
from find_amd.structures import Meshes, TexturesVertex
from find_amd.train_utils import sample_latent_vectors
from oracle import compose_ref
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'composition.npz'))
dev = torch.device('cuda')
lab = {k[len('labels/'):]: [str(s) for s in z[k]] for k in z.files if k.startswith('labels/')}
opts = Opts(chamf_loss=True, smooth_loss=True, texture_loss=True, use_pose_code=True, use_latent_labels=True)
mwl = ModelWithLoss(opts=opts, device='cpu', use_shapevec=True, use_texvec=True, use_posevec=True, train_size=3, val_size=3, shapevec_size=100,
					texvec_size=100, posevec_size=100, template_mesh_loc=None, latent_labels=lab)
m = mwl.model
m.set_template(torch.from_numpy(z['sd/template_verts'])[0], torch.from_numpy(z['sd/template_faces'])[0])
m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}, strict=True)
mwl = mwl.to(dev); m = mwl.model
gv, gf, gc = (torch.from_numpy(z[f'gt/{k}']) for k in ('verts', 'faces', 'colours'))
feet, names = [str(s) for s in z['batch/feet']], [str(s) for s in z['batch/names']]
idx = [0, 1, 2]
dr = [(torch.from_numpy(z[f'case/net/draw/{i}/face_idx']), torch.from_numpy(z[f'case/net/draw/{i}/uv'])) for i in range(3)]
params = dict(mwl.named_parameters())
B = torch.from_numpy(z['B']).double()
for flags in (dict(chamf=True), dict(smooth=True), dict(texture=True), dict(chamf=True, smooth=True, texture=True)):
	for prec in ('bf16x3', 'fp32'):
		from find_amd import functional as FF
		FF.set_mlp_precision(prec)
		b = dict(mesh=Meshes(gv[idx].to(dev).contiguous(), gf.to(dev), TexturesVertex(gc[idx].to(dev).contiguous())), idx=torch.tensor(idx, device=dev), name=[names[i] for i in idx],
				 shape=[feet[i] for i in idx], tex=[feet[i] for i in idx], pose=[names[i] for i in idx], reg=[names[i] for i in idx])
		b.update(sample_latent_vectors(b, m.latent_vectors_train))
		for p in mwl.parameters():
			p.grad = None
		with FixedDraws([(a.to(dev), c.to(dev)) for a, c in dr]):
			loss, _ = mwl(b, 0, opts, **flags)
		loss.backward()
		sd = {k[3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith('sd/')}
		sd = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
		rows = {k: torch.from_numpy(z[f'case/net/rows/{k}_train']) for k in ('shapevec', 'texvec', 'posevec', 'reg')}
		lat = {k: sd[f'{k}.data'][rows[k]] for k in rows}
		tot, _ = compose_ref.train3d_losses(sd, B, sd['template_verts'], sd['template_faces'][0], lat, gv[idx].double(), gf, gc[idx].double(),
											dict(gt=(dr[0][0], dr[0][1].double()), pred=(dr[1][0], dr[1][1].double()), tex=(dr[2][0], dr[2][1].double())),
											chamf=flags.get('chamf', False), smooth=flags.get('smooth', False), texture=flags.get('texture', False))
		tot.backward()
		worst = {}
		for k, p in params.items():
			kk = k[len('model.'):]
			if p.grad is None or sd[kk].grad is None:
				continue
			w = sd[kk].grad
			worst[kk] = float((p.grad.detach().cpu().double() - w).abs().max() / max(1e-3, w.abs().max()))
		top = sorted(worst.items(), key=lambda t: -t[1])[:4]
		print(flags, prec, 'loss rel err %.1e' % abs(loss.item() / tot.item() - 1), 'worst gradient errors:', [(k, '%.1e' % v) for k, v in top], flush=True)

# ---- the Chamfer term's gradient at the predicted VERTICES (before the MLP's backward), float64 oracle against the GPU
print('--- chamfer only: d loss / d verts')
FF.set_mlp_precision('fp32')
b = dict(mesh=Meshes(gv[idx].to(dev).contiguous(), gf.to(dev), TexturesVertex(gc[idx].to(dev).contiguous())), idx=torch.tensor(idx, device=dev), name=[names[i] for i in idx],
		 shape=[feet[i] for i in idx], tex=[feet[i] for i in idx], pose=[names[i] for i in idx], reg=[names[i] for i in idx])
b.update(sample_latent_vectors(b, m.latent_vectors_train))
from find_amd import losses as LL
res = m.get_meshes_from_batch(b, is_train=True)
verts = res['verts']; verts.retain_grad()
with FixedDraws([(a.to(dev), c.to(dev)) for a, c in dr]):
	l = LL.DisplacementLoss()(m, res, b, 0)['loss']
l.backward()
gG = verts.grad.detach().cpu().double()
from oracle import geom_ref
for dt in (torch.float64, torch.float32):
	v = verts.detach().cpu().to(dt).requires_grad_(True)
	tf = torch.from_numpy(z['sd/template_faces'])[0].long()
	gt_s = geom_ref.sample_points(gv[idx].to(dt), gf, dr[0][0], dr[0][1].to(dt))
	pr_s = geom_ref.sample_points(v, tf, dr[1][0], dr[1][1].to(dt))
	lo = geom_ref.chamfer_distance(pr_s, gt_s)
	lo.backward()
	print(dt, 'loss', l.item(), lo.item(), 'grad verts: max |oracle|', float(v.grad.abs().max()), 'max |gpu - oracle|', float((gG - v.grad.double()).abs().max()),
		  'rel', float((gG - v.grad.double()).abs().max() / v.grad.abs().max()))

# ---- is it a nearest-neighbour near-tie?  Relative gap between the nearest and the second nearest target of every query (float64), both directions
gt64 = geom_ref.sample_points(gv[idx].double(), gf, dr[0][0], dr[0][1].double())
pr64 = geom_ref.sample_points(verts.detach().cpu().double(), tf, dr[1][0], dr[1][1].double())
for a, bb, nm in ((pr64, gt64, 'pred -> gt'), (gt64, pr64, 'gt -> pred')):
	d = ((a[:, :, None, :] - bb[:, None, :, :]) ** 2).sum(-1)
	two = torch.topk(d, 2, dim=-1, largest=False).values
	gap = (two[..., 1] - two[..., 0]) / two[..., 1]
	print(nm, 'smallest relative gaps:', [float('%.2e' % g) for g in torch.sort(gap.reshape(-1)).values[:5]])
# GPU indices against the fp32 oracle's on the GPU's own sample positions
from find_amd import functional as FN
with FixedDraws([(a.to(dev), c.to(dev)) for a, c in dr]):
	gts = LL.sample_points_from_meshes(b['mesh'], num_samples=5000)
	prs = LL.sample_points_from_meshes(res['meshes'], num_samples=5000)
for a, bb, nm in ((prs, gts, 'pred -> gt'), (gts, prs, 'gt -> pred')):
	dG, iG = FN.knn1(a.detach(), bb.detach())
	dO, iO = geom_ref.knn1(a.detach().cpu(), bb.detach().cpu())
	print(nm, 'index mismatches GPU vs fp32 oracle on the same points:', int((iG.cpu().long() != iO.long()).sum()))
