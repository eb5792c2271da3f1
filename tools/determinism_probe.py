#!/usr/bin/env python3
"""Run the main MLP pass (get_meshes_from_batch + backward) of n feet several times on the same inputs and compare every gradient bit for bit:
which parameters' gradients change from run to run?  python tools/determinism_probe.py [n_feet] [runs]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from find_amd import _lib
if os.environ.get('FIND_LIB'):   # a library variant under test
	_lib.LIB_PATH = os.environ['FIND_LIB']
from find_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device('cuda:0')
model = synthetic.make_model(6890, train_size=max(16, n), val_size=2, device=dev)
lat = synthetic.latents(max(16, n), seed=5, device=dev)
with torch.no_grad():
	for k in ('shapevec', 'texvec', 'posevec', 'reg'):
		getattr(model, k).data.copy_(lat[k])
names = [k for k, p in model.named_parameters() if p.requires_grad]
params = [p for p in model.parameters() if p.requires_grad]
idx = torch.arange(n, device=dev)
ref = None
for r in range(runs):
	for p in params:
		p.grad = None
	batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx], reg_train=model.reg[idx])
	out = model.get_meshes_from_batch(batch, is_train=True)
	(((out['verts'] ** 2).sum() + (out['col'] ** 2).sum()) / n).backward()
	torch.cuda.synchronize()
	g = [None if p.grad is None else p.grad.clone() for p in params]
	o = [out['verts'].detach().clone(), out['col'].detach().clone()]
	if ref is None:
		ref, oref, prev = g, o, g
		continue
	nprev = sum(int(a is not None and not torch.equal(a, b)) for a, b in zip(prev, g))
	prev = g
	bad = []
	for k, a, b in zip(names, ref, g):
		if a is not None and not torch.equal(a, b):
			d = (a - b).abs()
			bad.append(f'{k}{tuple(a.shape)}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e} (|g| max {a.abs().max().item():.3e})')
	fo = [torch.equal(a, b) for a, b in zip(oref, o)]
	print(f'run {r}: forward outputs equal {fo}; gradients differing from run 0: {len(bad)}, from the run before: {nprev}', flush=True)
	for line in bad[:int(os.environ.get('PROBE_LINES', '0'))]:
		print('   ', line)
