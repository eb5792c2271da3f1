"""Fused optimisers (C-ABI find_adam_step / find_sgd_step behind find_amd.optim) against the oracle's golden trajectories and
against torch.optim running on the same device: the three optimisers of the reference's step (train.py:161-168)."""
import numpy as np
import pytest
import torch

from tests.test_oracle_optim import CASES, GOLD, SHAPES, STEPS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gold():
	return np.load(GOLD)


@pytest.mark.parametrize('name', sorted(CASES))
def test_fused_step_matches_golden_trajectory(gold, name):
	from find_amd import optim
	kind, kw = CASES[name]
	ps = [torch.nn.Parameter(torch.from_numpy(gold[f'p0/{i}']).cuda()) for i in range(SHAPES)]
	opt = (optim.Adam if kind == 'adam' else optim.SGD)(ps, **kw)
	for k in range(STEPS):
		for i, p in enumerate(ps):
			p.grad = torch.from_numpy(gold[f'grad/{k}/{i}']).cuda()
		opt.step()
		for i, p in enumerate(ps):
			ref = gold[f'{name}/{k}/{i}']
			assert np.allclose(p.detach().cpu().numpy(), ref, rtol=2e-6, atol=1e-7), (name, k, i)


def test_find_parameter_set_matches_torch_optim_and_state_dict_is_compatible():
	"""The real parameter set (MLP + latent tables + registration, 16 training feet): 5 steps of the reference's three optimisers,
	fused vs torch.optim on the same GPU; the state_dict of one loads into the other."""
	from find_amd import optim, synthetic
	g = torch.Generator().manual_seed(0)
	models = [synthetic.make_model(1002, train_size=16, val_size=2, device='cuda') for _ in range(2)]
	models[1].load_state_dict(models[0].state_dict())

	def groups(m):
		return [p for p in m.main_params], [p for p in m.reg_params], [p for p in m.latent_params]

	(n0, r0, l0), (n1, r1, l1) = groups(models[0]), groups(models[1])
	fused = [optim.Adam(n0, lr=5e-4), optim.SGD(r0, lr=1e-2, momentum=0.9), optim.Adam(l0, lr=1e-3)]
	ref = [torch.optim.Adam(n1, lr=5e-4), torch.optim.SGD(r1, lr=1e-2, momentum=0.9), torch.optim.Adam(l1, lr=1e-3)]
	for k in range(5):
		for pa, pb in zip(n0 + r0 + l0, n1 + r1 + l1):
			gr = torch.randn(pa.shape, generator=g).cuda() * 0.1
			pa.grad, pb.grad = gr.clone(), gr.clone()
		for o in fused + ref:
			o.step()
	for pa, pb in zip(n0 + r0 + l0, n1 + r1 + l1):
		assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7)
	# state round trip: torch's optimiser continues from the fused one's state and vice versa
	sd = fused[0].state_dict()
	t2 = torch.optim.Adam(n1, lr=5e-4)
	t2.load_state_dict(sd)
	f2 = optim.Adam(n0, lr=5e-4)
	f2.load_state_dict(ref[0].state_dict())
	for pa, pb in zip(n0, n1):
		gr = torch.randn(pa.shape, generator=g).cuda() * 0.1
		pa.grad, pb.grad = gr.clone(), gr.clone()
	f2.step(); t2.step()
	for pa, pb in zip(n0, n1):
		assert torch.allclose(pa, pb, rtol=4e-6, atol=2e-7)


def test_optimiser_rejects_cpu_parameters():
	from find_amd import optim
	p = torch.nn.Parameter(torch.zeros(4))
	p.grad = torch.ones(4)
	with pytest.raises(RuntimeError, match='no CPU fallback'):
		optim.Adam([p]).step()


def test_adam_capturable_matches_default_path():
	"""Adam(capturable=True): step count on the device, bias corrections formed there in fp32 (find_adam_step_dev) -- the same
	trajectory as the default host-arithmetic path to fp32 rounding, and the same state_dict layout."""
	from find_amd import optim
	g = torch.Generator().manual_seed(3)
	shapes = [(256, 515), (256,), (7, 100), (3, 256)]
	init = [torch.randn(*s, generator=g) for s in shapes]
	grads = [[torch.randn(*s, generator=g) * 0.1 for s in shapes] for _ in range(8)]
	res = []
	for cap in (False, True):
		ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
		opt = optim.Adam(ps, lr=5e-4, weight_decay=1e-3, capturable=cap)
		for gs in grads:
			for p, gr in zip(ps, gs):
				p.grad = gr.cuda()
			opt.step()
		torch.cuda.synchronize()
		res.append((ps, opt))
	(pa, oa), (pb, ob) = res
	for a, b in zip(pa, pb):
		assert (a - b).abs().max().item() < 1e-6 * max(1.0, a.abs().max().item())
	sa, sb = oa.state_dict(), ob.state_dict()
	assert sa['state'].keys() == sb['state'].keys()
	for k in sa['state']:
		assert set(sa['state'][k]) == set(sb['state'][k]) == {'step', 'exp_avg', 'exp_avg_sq'}
		assert float(sa['state'][k]['step']) == float(sb['state'][k]['step']) == 8.0
		assert sb['state'][k]['step'].is_cuda and not sa['state'][k]['step'].is_cuda
		assert torch.allclose(sa['state'][k]['exp_avg_sq'], sb['state'][k]['exp_avg_sq'])
	# the capturable optimiser keeps one counter per bucket inside; a state_dict holds a copy per parameter (no aliasing, no later movement)
	steps = [sb['state'][k]['step'] for k in sb['state']]
	assert len({t.data_ptr() for t in steps}) == len(steps)
	for p, gr in zip(pb, grads[0]):
		p.grad = gr.cuda()
	ob.step()
	torch.cuda.synchronize()
	assert all(float(t) == 8.0 for t in steps) and float(ob.state_dict()['state'][0]['step']) == 9.0
	# ... and loads back into a fresh optimiser that continues the same trajectory
	ps2 = [torch.nn.Parameter(t.detach().clone()) for t in pb]
	o2 = optim.Adam(ps2, lr=5e-4, weight_decay=1e-3, capturable=True)
	import copy
	o2.load_state_dict(copy.deepcopy(ob.state_dict()))   # (torch's load_state_dict keeps tensors that already have the right dtype and device: the moments would be shared)
	for q, p, gr in zip(ps2, pb, grads[1]):
		q.grad = gr.cuda(); p.grad = gr.cuda()
	o2.step(); ob.step()
	torch.cuda.synchronize()
	for q, p in zip(ps2, pb):
		assert torch.equal(q, p)
