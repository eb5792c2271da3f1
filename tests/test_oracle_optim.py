"""oracle/optim_ref.py against trajectories produced by torch.optim itself (tests/golden/optim.npz): the optimisers of the
reference's training step (src/train/train.py:161-168)."""
import os

import numpy as np
import pytest

from oracle import optim_ref

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'optim.npz')
SHAPES = 3
STEPS = 6
CASES = {
	'adam_net': ('adam', dict(lr=5e-4)),
	'adam_wd': ('adam', dict(lr=1e-2, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.01)),
	'sgd_reg': ('sgd', dict(lr=1e-2, momentum=0.9)),
	'sgd_nesterov_wd': ('sgd', dict(lr=3e-3, momentum=0.8, nesterov=True, weight_decay=0.05)),
	'sgd_plain': ('sgd', dict(lr=1e-2)),
}


@pytest.fixture(scope='module')
def gold():
	return np.load(GOLD)


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_matches_torch_optim(gold, name):
	kind, kw = CASES[name]
	for i in range(SHAPES):
		p = gold[f'p0/{i}']
		m = np.zeros_like(p); v = np.zeros_like(p); buf = None
		for k in range(STEPS):
			g = gold[f'grad/{k}/{i}']
			if kind == 'adam':
				p, m, v = optim_ref.adam_step(p, g, m, v, k + 1, **kw)
			else:
				p, buf = optim_ref.sgd_step(p, g, buf, **kw)
			ref = gold[f'{name}/{k}/{i}']
			# same fp32 formulas; fused-multiply-add contraction inside torch's kernels moves the last bit
			assert np.allclose(p, ref, rtol=2e-6, atol=1e-7), (name, i, k, np.abs(p - ref).max())
