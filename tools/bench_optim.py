"""Time one optimiser step over FIND's parameter set (MLP 868 k floats + latent tables + registration): fused (find_amd.optim)
vs torch.optim single-tensor / foreach / fused implementations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from find_amd import optim, synthetic

m = synthetic.make_model(6890, train_size=16, val_size=2, device='cuda')
ps = list(m.main_params) + list(m.latent_params)
for p in ps:
	p.grad = torch.randn_like(p) * 0.1


def t(opt, n=200):
	for _ in range(10):
		opt.step()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(n):
		opt.step()
	e1.record(); e1.synchronize()
	return e0.elapsed_time(e1) / n * 1e3


print('tensors %d, floats %d' % (len(ps), sum(p.numel() for p in ps)))
print('find_amd.optim.Adam         %.1f us/step' % t(optim.Adam(ps, lr=1e-4)))
print('torch Adam (foreach=False)  %.1f us/step' % t(torch.optim.Adam(ps, lr=1e-4, foreach=False)))
print('torch Adam (foreach=True)   %.1f us/step' % t(torch.optim.Adam(ps, lr=1e-4, foreach=True)))
try:
	print('torch Adam (fused=True)     %.1f us/step' % t(torch.optim.Adam(ps, lr=1e-4, fused=True)))
except Exception as e:
	print('torch fused Adam unavailable:', type(e).__name__)
