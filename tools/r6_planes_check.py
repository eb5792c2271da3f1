"""bf16-plane storage of the heads' hidden activations (find_ctx knob act_planes) against fp32 storage: the same 16 x 6890 shared-template
forward + backward, outputs and every gradient compared bit for bit (the split is exact; only the two bias gradients that dw6_planes_kernel
sums with v_dot2 may differ in the last bits), and the step time of both.  python tools/r6_planes_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from find_amd import _lib, synthetic
dev = torch.device('cuda:0')
n_feet = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = synthetic.make_model(6890, train_size=n_feet, val_size=1, device=dev)
lat = synthetic.latents(n_feet, seed=0, device=dev)
out = {}
for planes in (0, 1):
	_lib.set_tuning('act_planes', planes)
	lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
	for p in model.parameters():
		p.grad = None
	res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
	loss = (res['verts'] ** 2).sum() + (res['col'] ** 2).sum()
	loss.backward()
	torch.cuda.synchronize()
	out[planes] = dict(verts=res['verts'].detach().clone(), col=res['col'].detach().clone(), **{'lat.' + k: v.grad.clone() for k, v in lv.items()},
					   **{k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
worst = {}
for k in out[0]:
	a, b = out[0][k], out[1][k]
	if not torch.equal(a, b):
		worst[k] = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
print('tensors compared:', len(out[0]), '; not bit-identical:', {k: '%.1e' % v for k, v in worst.items()} or 'none')
assert all(v < 2e-6 for v in worst.values()), worst
assert all(k.endswith('bias') for k in worst), 'only bias gradients may differ (v_dot2 summation order)'
for planes in (0, 1, 0, 1):
	_lib.set_tuning('act_planes', planes)
	lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
	def step():
		for p in model.parameters():
			p.grad = None
		if os.environ.get('ONE_HEAD'):   # the displacement head alone (the colour head is never read): its kernels run one after the other
			res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'], lazy_colours=True)
			(res['verts'] ** 2).sum().backward()
			return
		res = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
		((res['verts'] ** 2).sum() + (res['col'] ** 2).sum()).backward()
	for _ in range(10):
		step()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(30):
		step()
	e1.record(); e1.synchronize()
	print(f'act_planes={planes}: {e0.elapsed_time(e1) / 30:.4f} ms per forward + backward (both heads)')
