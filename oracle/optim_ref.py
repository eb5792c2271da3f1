"""CPU restatement (numpy, fp32 arithmetic) of the optimiser updates the reference performs through torch.optim
(src/train/train.py:161-168: Adam, and SGD with momentum 0.9; stepped in src/train/trainer.py:121-123).
TEST INFRASTRUCTURE ONLY -- imported by tests/ alone.

Pinned: tests/golden/optim.npz holds parameter trajectories produced by torch.optim.Adam / torch.optim.SGD themselves
(tests/golden/make_golden_optim.py, run in the build container against torch 2.10 CPU); tests/test_oracle_optim.py checks this
restatement against them.  Formulas follow torch/optim/adam.py::_single_tensor_adam and torch/optim/sgd.py::_single_tensor_sgd."""
import numpy as np

F = np.float32


def adam_step(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
	"""One dense Adam update; `step` is the 1-based count including this update.  Returns (p, m, v) as new fp32 arrays."""
	p, g, m, v = (np.asarray(a, dtype=F) for a in (p, g, m, v))
	b1, b2 = betas
	if weight_decay != 0:
		g = g + F(weight_decay) * p
	m = m + (g - m) * F(1.0 - b1)                       # lerp_(grad, 1 - beta1)
	v = v * F(b2) + F(1.0 - b2) * g * g                # mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
	bc1 = 1.0 - b1 ** step                             # Python floats (double), as torch
	bc2 = 1.0 - b2 ** step
	step_size = F(lr / bc1)
	denom = np.sqrt(v) / F(bc2 ** 0.5) + F(eps)
	p = p - step_size * (m / denom)
	return p.astype(F), m.astype(F), v.astype(F)


def sgd_step(p, g, buf, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
	"""One SGD update; buf is None before the first step (torch clones the gradient).  Returns (p, buf)."""
	p, g = np.asarray(p, dtype=F), np.asarray(g, dtype=F)
	if weight_decay != 0:
		g = g + F(weight_decay) * p
	if momentum != 0:
		buf = g.copy() if buf is None else np.asarray(buf, dtype=F) * F(momentum) + F(1.0 - dampening) * g
		g = g + F(momentum) * buf if nesterov else buf
	p = p - F(lr) * g
	return p.astype(F), (None if buf is None else buf.astype(F))
