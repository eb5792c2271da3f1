"""Marginal cost of each loss term of the headline step and phase boundaries on the GPU timeline (HIP events on the caller's stream, untraced):
python tools/r6_marginal.py [steps].  One process, one box: every variant in turn, twice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd.train_utils import backward_on_this_thread, sample_latent_vectors

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
run = bench.Run(1)
su = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)
mwl, opts, opt, batches = su['mwl'], su['opts'], su['opt'], su['batches']
vectors = mwl.model.latent_vectors_train


def make_step(flags, phases=None):
	def step():
		opt.zero_grad(set_to_none=True)
		b = dict(batches[0])
		b.update(sample_latent_vectors(b, vectors))
		if phases is not None:
			phases[0].record()
		loss, _ = mwl(b, 0, opts, **flags)
		if phases is not None:
			phases[1].record()
		loss.backward()
		if phases is not None:
			phases[2].record()
		opt.step()
		if phases is not None:
			phases[3].record()
	return step


def timeit(step, n):
	for _ in range(8):
		step()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n):
		step()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / n * 1e3


variants = [('all', dict(chamf=True, smooth=True, texture=True)), ('no texture', dict(chamf=True, smooth=True, texture=False)),
			('no smooth', dict(chamf=True, smooth=False, texture=True)), ('no chamf', dict(chamf=False, smooth=True, texture=True)),
			('chamf only', dict(chamf=True, smooth=False, texture=False)), ('texture only', dict(chamf=False, smooth=False, texture=True))]
with backward_on_this_thread():
	for rnd in range(2):
		for name, fl in variants:
			print(f'{name:14s} {timeit(make_step(fl), steps):.4f} ms/step', flush=True)
	# phase boundaries of the full step (events on the caller's stream; the host runs ahead, so these are GPU times)
	ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
	step_fns = [make_step(variants[0][1], e) for e in ev]
	for _ in range(8):
		make_step(variants[0][1])()
	torch.cuda.synchronize()
	for f in step_fns:
		f()
	torch.cuda.synchronize()
	import statistics as st
	fw = st.median(e[0].elapsed_time(e[1]) for e in ev)
	bw = st.median(e[1].elapsed_time(e[2]) for e in ev)
	op = st.median(e[2].elapsed_time(e[3]) for e in ev)
	pro = st.median(ev[i][3].elapsed_time(ev[i + 1][0]) for i in range(steps - 1))
	print(f'phases (median, ms): prologue {pro:.3f}  forward+losses {fw:.3f}  backward {bw:.3f}  optimiser {op:.3f}  sum {pro + fw + bw + op:.3f}')
