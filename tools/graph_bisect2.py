"""Which ingredient of the Trainer scenario (eager checkpoint epoch with rendering, then a capture) breaks hipStreamEndCapture?
Every variant runs in its own child process.  python tools/graph_bisect2.py [variant]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

VARIANTS = ['graph_only', 'eager_then_graph', 'eager_norender_then_graph', 'eager_del_then_graph', 'trainer_gc']


def run(name):
	import faulthandler
	import gc
	import pathlib
	faulthandler.enable()
	import numpy as np
	import torch
	if name == 'no_overlap':
		import find_amd.model_with_loss as M
		M.OVERLAP_GT_RENDER = False
	if name == 'no_watch':
		import find_amd.functional_render as FR
		FR.FLAG_POLICY = 'ignore'
	import test_gpu_trainloop as T
	tmp = pathlib.Path(tempfile.mkdtemp())
	if name == 'graph_only':
		T._trainer_run('auto', tmp, epochs=(1,))
	elif name == 'eager_norender_then_graph':
		# epoch 0 eager without rendering (graph=False trainer cannot switch: emulate with opts.no_rendering and graph forced off by step order)
		from find_amd.graph import GraphedStep
		from find_amd.train_utils import sample_latent_vectors
		mwl, opts, batch_of, _, opt = T._setup(1002, 1002, capturable=True)
		for i in (0, 3, 1, 2):
			b = batch_of(i)
			b.update(sample_latent_vectors(b, mwl.model.latent_vectors_train))
			opt.zero_grad(set_to_none=True)
			loss, _ = mwl(b, 0, opts, **opts.net_train_kwargs())
			loss.backward()
			opt.step()
		gs = GraphedStep(mwl, opts, [opt], warmup=1, **opts.net_train_kwargs())
		for i in (0, 3, 1, 2):
			gs(batch_of(i))
	elif name == 'eager_del_then_graph':
		from find_amd.graph import GraphedStep
		from find_amd.train_utils import sample_latent_vectors
		mwl, opts, batch_of, _, opt = T._setup(1002, 1002, capturable=True)
		for i in (0, 3, 1, 2):
			b = batch_of(i)
			b.update(sample_latent_vectors(b, mwl.model.latent_vectors_train))
			opt.zero_grad(set_to_none=True)
			loss, _ = mwl(b, 0, opts, **opts.net_train_kwargs())
			loss.backward()
			opt.step()
		del loss, b, _
		gc.collect()
		gs = GraphedStep(mwl, opts, [opt], warmup=1, **opts.net_train_kwargs())
		for i in (0, 3, 1, 2):
			gs(batch_of(i))
	elif name == 'trainer_gc':
		import find_amd.graph as G
		orig = G.GraphedStep._capture
		def cap(self, batch, epoch):
			gc.collect()
			return orig(self, batch, epoch)
		G.GraphedStep._capture = cap
		T._trainer_run('auto', tmp, epochs=(0, 1))
	elif name == 'render_eager_once_then_graph_step':
		from find_amd.graph import GraphedStep
		from find_amd.train_utils import sample_latent_vectors
		mwl, opts, batch_of, _, opt = T._setup(1002, 1002, capturable=True)
		b = batch_of(0)
		b.update(sample_latent_vectors(b, mwl.model.latent_vectors_train))
		opt.zero_grad(set_to_none=True)
		loss, _ = mwl(b, 0, opts, **opts.net_train_kwargs(), render_foot=True, save_renders=True, render_dir=str(tmp))
		loss.backward()
		opt.step()
		gs = GraphedStep(mwl, opts, [opt], warmup=1, **opts.net_train_kwargs())
		for i in (0, 3, 1, 2):
			gs(batch_of(i))
	else:
		T._trainer_run('auto', tmp, epochs=(0, 1))
	torch.cuda.synchronize()
	print('OK', name)


if __name__ == '__main__':
	if len(sys.argv) > 1:
		run(sys.argv[1])
	else:
		for v in VARIANTS:
			r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True)
			tail = (r.stdout + r.stderr).strip().splitlines()
			print(f'{v:40s} rc={r.returncode}  ' + (tail[-1][:150] if tail else ''), flush=True)
			if r.returncode != 0:
				print('\n'.join('      ' + t[:200] for t in tail[-14:]), flush=True)
