"""A/B of the small-call machinery through find_ctx knobs -- fused layer chains (fused_max_units), splits
per foot of the grouped weight gradients (group_spf; 0 = cost model) -- on the MLP forward + backward at batch 1 (6890 template rows), on
1000 free points x 1 and x 16 feet (the texture pass) and on the C2 step (16 feet, shared trunk).  python tools/fused_ab.py [knob ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd import _lib, synthetic

dev = torch.device('cuda:0')


def timeit(fn, n=200, warm=30):
	for _ in range(warm):
		fn()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n):
		fn()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / n * 1e3


def mlp_case(n_feet, n_verts, free_pts):
	model = synthetic.make_model(n_verts if not free_pts else 1002, train_size=max(n_feet, 1), val_size=1, device=dev)
	lat = synthetic.latents(n_feet, seed=0, device=dev)
	pos = (torch.rand(n_feet, n_verts, 3, device=dev) * 0.2 - 0.1) if free_pts else None
	params = [p for p in model.parameters() if p.requires_grad]

	def step():
		for p in params:
			p.grad = None
		lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
		if free_pts:
			r = model(pos, shapevec=lv['shapevec'], texvec=lv['texvec'], posevec=lv['posevec'])
			(r['disp'].sum() + r['col'].sum()).backward()
		else:
			r = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
			(r['verts'].sum() + r['col'].sum()).backward()
	return step


cases = [('batch 1 x 6890 template rows', mlp_case(1, 6890, False)), ('1 x 1000 free points', mlp_case(1, 1000, True)),
		 ('16 x 1000 free points', mlp_case(16, 1000, True)), ('C2: 16 feet x 6890 (shared trunk)', bench.build_step(dev, 0)[2])]
# (knob, A, B): A/B pairs measured twice, interleaved
KNOBS = [('fused_max_units', 0, 512), ('ablate', 32, 0)] + [('group_spf', v, 0) for v in (1, 2, 4, 8)]
if len(sys.argv) > 1:
	KNOBS = [k for k in KNOBS if k[0] in sys.argv[1:]]
for name, step in cases:
	for knob, a, b in KNOBS:
		res = {}
		for v in (a, b, a, b):
			_lib.set_tuning(knob, v)
			res.setdefault(v, []).append(timeit(step, n=100 if 'C2' in name else 300))
		_lib.set_tuning(knob, b)
		print(f'{name}: {knob}={a}: {min(res[a]):.3f} ms   {knob}={b}: {min(res[b]):.3f} ms', flush=True)
