"""Untraced GPU timestamps along the caller's stream of the headline step (HIP events recorded at python-level boundaries and in tensor
hooks of the backward): where the step's time goes WITHOUT the tracer (whose host overhead reorders the step).  python tools/r6_phases.py [steps]"""
import os, sys, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import find_amd.model_with_loss as MWL
from find_amd.train_utils import backward_on_this_thread, sample_latent_vectors

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
run = bench.Run(1)
su = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)
mwl, opts, opt, batches, flags = su['mwl'], su['opts'], su['opt'], su['batches'], su['flags']
vectors = mwl.model.latent_vectors_train
marks = []   # (name, event) of the current step


def mark(name):
	e = torch.cuda.Event(enable_timing=True)
	e.record()
	marks.append((name, e, time.perf_counter()))


orig_get = mwl.model.get_meshes_from_batch
def get_meshes(*a, **k):
	mark('fwd: main pass issued from here')
	res = orig_get(*a, **k)
	mark('fwd: main pass done')
	res['verts'].register_hook(lambda g: mark('bwd: d verts known (loss-side backward done)'))
	res['disp'].register_hook(lambda g: mark('bwd: d disp known (registration backward done)')) if 'disp' in res and res['disp'].requires_grad else None
	return res
mwl.model.get_meshes_from_batch = get_meshes
import find_amd.losses as LO
orig_mm = LO.FN.masked_mse
def masked_mse(pred, target):
	out = orig_mm(pred, target)
	out.register_hook(lambda g: mark('bwd: texture term backward starts'))
	return out
LO.FN.masked_mse = masked_mse
def wrap_bwd(cls, label):
	orig = cls.backward
	def bwd(ctx, *g):
		mark(f'bwd: {label} backward entered')
		out = orig(ctx, *g)
		mark(f'bwd: {label} backward issued')
		return out
	cls.backward = staticmethod(bwd)
for _cls, _label in ((MWL.FN._MLP, 'MLP'), (MWL.FN._SmoothLoss, 'smoothness'), (MWL.FN._Chamfer, 'Chamfer'), (MWL.FN._SampleSurface, 'sampling'), (MWL.FN._Register, 'registration'), (MWL.FN._MaskedMSE, 'masked MSE')):
	wrap_bwd(_cls, _label)
orig_wt = MWL.FN.weighted_terms
def weighted(terms, weights):
	mark('fwd: loss terms done (joined)')
	total, scaled = orig_wt(terms, weights)
	total.register_hook(lambda g: mark('bwd: start'))
	return total, scaled
MWL.FN.weighted_terms = weighted
w0 = mwl.model.base[0].weight
w0.register_post_accumulate_grad_hook(lambda p: mark('bwd: base[0].weight accumulated (both MLP backward passes returned)'))


def step():
	del marks[:]
	opt.zero_grad(set_to_none=True)
	b = dict(batches[0])
	mark('step start')
	b.update(sample_latent_vectors(b, vectors))
	loss, _ = mwl(b, 0, opts, **flags)
	mark('fwd: done')
	loss.backward()
	mark('bwd: done')
	opt.step()
	mark('step end')
	return list(marks)


with backward_on_this_thread():
	for _ in range(10):
		step()
	torch.cuda.synchronize()
	rec = [step() for _ in range(steps)]
	torch.cuda.synchronize()
names = [n for n, _, _ in rec[0]]
rec = [r for r in rec if [n for n, _, _ in r] == names]
print(f'{"GPU time on the caller stream since step start (median of %d steps, us)" % steps}')
for i, n in enumerate(names):
	t = st.median(r[0][1].elapsed_time(r[i][1]) * 1e3 for r in rec)
	prev = st.median(r[0][1].elapsed_time(r[i - 1][1]) * 1e3 for r in rec) if i else 0.0
	h = st.median((r[i][2] - r[0][2]) * 1e6 for r in rec)
	print(f'{t:9.1f}  (+{t - prev:7.1f})  host {h:8.1f}  {n}')
per = st.median(rec[i][0][1].elapsed_time(rec[i + 1][0][1]) * 1e3 for i in range(len(rec) - 1))
print(f'step period {per:.1f} us   (host column: when the host ISSUED the mark, same origin; a GPU time equal to it = the GPU was waiting for the host)')
