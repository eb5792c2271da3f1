"""Which piece of the batch-1 train_3d step survives HIP-graph capture?  Runs every component in its own child process (a fault in
hipStreamEndCapture kills the process) and prints one line per component.  python tools/graph_bisect.py [component]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

COMPONENTS = ['mlp_shared', 'mlp_free_pts', 'mlp_col_only', 'register', 'latent_gather', 'face_areas', 'multinomial', 'rand', 'sample_points', 'chamfer', 'smooth',
			  'adam', 'sgd', 'texture_loss', 'chamf_loss', 'full_fwd', 'full_fwd_bwd', 'full_step']


def run(name):
	import torch
	from find_amd import functional as FN
	from find_amd import optim, synthetic
	from find_amd.losses import DisplacementLoss, MeshSmoothnessLoss, TextureLossGTSpace, sample_points_from_meshes
	from find_amd.structures import Meshes, TexturesVertex
	dev = torch.device('cuda:0')
	n_verts = 1002
	model = synthetic.make_model(n_verts, train_size=4, val_size=1, device=dev)
	lat = synthetic.latents(4, seed=0, device=dev)
	with torch.no_grad():
		for k in ('shapevec', 'texvec', 'posevec', 'reg'):
			getattr(model, k).data.copy_(lat[k])
	gv, gf, gc = synthetic.gt_feet(1, 1002, seed=0, device=dev)
	gt = Meshes(gv, gf, TexturesVertex(gc.clamp(0.05, 0.95)))
	idx = torch.tensor([2], device=dev)
	params = [p for p in model.parameters() if p.requires_grad]
	opt = optim.Adam(model.main_params, lr=1e-4, capturable=True)
	sgd = optim.SGD(model.reg_params, lr=1e-3, momentum=0.9)
	pts = (torch.rand(1, 1000, 3, device=dev) * 0.1)
	x = torch.rand(1, 5000, 3, device=dev)
	y = torch.rand(1, 5000, 3, device=dev)

	def latents():
		return dict(shapevec=model.shapevec[idx], texvec=model.texvec[idx], posevec=model.posevec[idx], reg=model.reg[idx])

	def meshes():
		l = latents()
		return model.get_meshes(shapevec=l['shapevec'], reg=l['reg'], texvec=l['texvec'], posevec=l['posevec'])

	def zero():
		for p in params:
			p.grad = None

	def body():
		zero()
		if name == 'mlp_shared':
			r = meshes()
			(r['verts'].sum() + r['col'].sum()).backward()
		elif name == 'mlp_free_pts':
			l = latents()
			r = model(pts, shapevec=l['shapevec'], texvec=l['texvec'], posevec=l['posevec'])
			(r['disp'].sum() + r['col'].sum()).backward()
		elif name == 'mlp_col_only':
			l = latents()
			r = model(pts, shapevec=l['shapevec'], texvec=l['texvec'], posevec=l['posevec'])
			r['col'].sum().backward()
		elif name == 'register':
			d = torch.zeros(1, n_verts, 3, device=dev, requires_grad=True)
			FN.register_points(model.template_verts.data, d, model.reg[idx]).sum().backward()
		elif name == 'latent_gather':
			model.shapevec[idx].sum().backward()
		elif name == 'face_areas':
			FN.face_areas(gv, gf)
		elif name == 'multinomial':
			a = FN.face_areas(gv, gf)
			torch.multinomial(a, 5000, replacement=True)
		elif name == 'rand':
			torch.rand(1, 5000, 2, device=dev)
		elif name == 'sample_points':
			sample_points_from_meshes(gt, 5000, return_textures=True)
		elif name == 'chamfer':
			xg = x.clone().requires_grad_(True)
			FN.chamfer_distance(xg, y)[0].backward()
		elif name == 'smooth':
			MeshSmoothnessLoss()(meshes()['meshes']).backward()
		elif name == 'adam':
			for p in model.main_params:
				p.grad = torch.ones_like(p)
			opt.step()
		elif name == 'sgd':
			for p in model.reg_params:
				p.grad = torch.ones_like(p)
			sgd.step()
		elif name == 'texture_loss':
			l = latents()
			TextureLossGTSpace()(model, dict(mesh=gt), shapevec=l['shapevec'], texvec=l['texvec'], posevec=l['posevec']).backward()
		elif name == 'chamf_loss':
			DisplacementLoss()(model, meshes(), dict(mesh=gt), 0)['loss'].backward()
		elif name in ('full_fwd', 'full_fwd_bwd', 'full_step'):
			l = latents()
			res = model.get_meshes(shapevec=l['shapevec'], reg=l['reg'], texvec=l['texvec'], posevec=l['posevec'])
			loss = DisplacementLoss()(model, res, dict(mesh=gt), 0)['loss'] * 1e4 + MeshSmoothnessLoss()(res['meshes']) * 1e3 + \
				TextureLossGTSpace()(model, dict(mesh=gt), shapevec=l['shapevec'], texvec=l['texvec'], posevec=l['posevec'])
			if name != 'full_fwd':
				loss.backward()
			if name == 'full_step':
				opt.step()
		else:
			raise SystemExit(f'unknown component {name}')

	s = torch.cuda.Stream()
	s.wait_stream(torch.cuda.current_stream())
	with torch.cuda.stream(s):
		for _ in range(2):
			body()
	torch.cuda.current_stream().wait_stream(s)
	torch.cuda.synchronize()
	zero()
	g = torch.cuda.CUDAGraph()
	with torch.cuda.graph(g):
		body()
	torch.cuda.synchronize()
	for _ in range(3):
		g.replay()
	torch.cuda.synchronize()
	print(f'{name}: capture + replay OK', flush=True)


if __name__ == '__main__':
	if len(sys.argv) > 1:
		run(sys.argv[1])
	else:
		for c in COMPONENTS:
			r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
			if r.returncode == 0:
				print(r.stdout.strip().splitlines()[-1], flush=True)
			else:
				tail = [l for l in (r.stderr or '').strip().splitlines() if l.strip()][-3:]
				print(f'{c}: FAILED rc={r.returncode}: ' + ' | '.join(t[:160] for t in tail), flush=True)
