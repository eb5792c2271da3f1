"""Host time of the phases of the one-rank data-parallel step against the plain step (empty queue before every step): which phase the DP path's
extra host time sits in.  python tools/r6_dp_phases.py"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd import distributed as fdist
from find_amd.train_utils import backward_on_this_thread, sample_latent_vectors
run = bench.Run(1)
with socket.socket() as _s:
	_s.bind(('127.0.0.1', 0)); _port = _s.getsockname()[1]
torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_port}', rank=0, world_size=1)
su = bench.train3d_setup(run, 16, 16, stage='net', labels=False, dp=False)
m = su['mwl'].model
opt, mwl, opts, flags, batches = su['opt'], su['mwl'], su['opts'], su['flags'], su['batches']
def make(bucket):
	def step(T):
		t = [time.perf_counter()]
		opt.zero_grad(set_to_none=True)
		b = dict(batches[0]); b.update(sample_latent_vectors(b, m.latent_vectors_train)); t.append(time.perf_counter())
		loss, _ = mwl(b, 0, opts, **flags); t.append(time.perf_counter())
		loss.backward(); t.append(time.perf_counter())
		if bucket is not None:
			bucket.allreduce_(async_op=True); t.append(time.perf_counter())
			bucket.wait(); t.append(time.perf_counter())
		else:
			t += [t[-1], t[-1]]
		opt.step(); t.append(time.perf_counter())
		for k in range(6):
			T[k].append(t[k + 1] - t[k])
	return step
names = ['zero+latents', 'forward', 'backward', 'allreduce_', 'wait', 'opt.step']
with backward_on_this_thread():
	for label in ('plain', 'dp', 'plain', 'dp'):
		bucket = None
		if label == 'dp':
			bucket = fdist.GradBucket([p for p in m.parameters() if p.requires_grad], early=[p for seq in (m.base, m.mlp_disp, m.mlp_col) for p in seq.parameters()])
			bucket.arm_early(m.base[0].weight)
		step = make(bucket)
		T = [[] for _ in range(6)]
		for _ in range(30):
			step([[] for _ in range(6)])
		for _ in range(60):
			torch.cuda.synchronize()
			step(T)
		torch.cuda.synchronize()
		med = [sorted(x)[len(x) // 2] * 1e6 for x in T]
		print(f'{label:6s} ' + '  '.join(f'{n} {v:6.0f}' for n, v in zip(names, med)) + f'   total {sum(med):6.0f} us', flush=True)
		if bucket is not None:
			bucket.close()
torch.distributed.destroy_process_group()
