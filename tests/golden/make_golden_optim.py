"""Golden trajectories of torch.optim.Adam / torch.optim.SGD (the optimisers the reference constructs, train.py:161-168),
generated with torch on CPU:  python tests/golden/make_golden_optim.py  ->  tests/golden/optim.npz"""
import os

import numpy as np
import torch

SHAPES = [(7, 5), (13,), (4, 3, 2)]
STEPS = 6
CASES = {
	'adam_net': ('adam', dict(lr=5e-4)),                                  # lr_net-style
	'adam_wd': ('adam', dict(lr=1e-2, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.01)),
	'sgd_reg': ('sgd', dict(lr=1e-2, momentum=0.9)),                      # the reference's registration optimiser
	'sgd_nesterov_wd': ('sgd', dict(lr=3e-3, momentum=0.8, nesterov=True, weight_decay=0.05)),
	'sgd_plain': ('sgd', dict(lr=1e-2)),
}


def main():
	g = torch.Generator().manual_seed(11)
	out = {}
	p0 = [torch.randn(s, generator=g) for s in SHAPES]
	grads = [[torch.randn(s, generator=g) * (0.1 + 0.3 * k) for s in SHAPES] for k in range(STEPS)]
	for i, p in enumerate(p0):
		out[f'p0/{i}'] = p.numpy()
	for k in range(STEPS):
		for i, gr in enumerate(grads[k]):
			out[f'grad/{k}/{i}'] = gr.numpy()
	for name, (kind, kw) in CASES.items():
		ps = [torch.nn.Parameter(p.clone()) for p in p0]
		opt = (torch.optim.Adam if kind == 'adam' else torch.optim.SGD)(ps, foreach=False, **kw)
		for k in range(STEPS):
			for p, gr in zip(ps, grads[k]):
				p.grad = gr.clone()
			opt.step()
			for i, p in enumerate(ps):
				out[f'{name}/{k}/{i}'] = p.detach().numpy().copy()
	np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'optim.npz'), **out)
	print('wrote optim.npz with', len(out), 'arrays')


if __name__ == '__main__':
	main()
