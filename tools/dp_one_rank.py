"""The headline step on ONE GPU through the data-parallel code path: a one-rank RCCL group, parameters broadcast, gradients in the
bucket, all-reduce before the optimiser step -- what a rank of `bench.py --gpus N` runs, minus the other ranks.  RCCL creates its
streams before the MLP context creates its side streams: if the step time moved with that (HIP maps streams onto four hardware queues in
creation order), the multi-GPU line would not be the single-GPU line.  python tools/dp_one_rank.py [steps] [dp | pg_only (group and its streams created, no bucket) | plain]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd import distributed as fdist
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
run = bench.Run(1)
mode = sys.argv[2] if len(sys.argv) > 2 else 'dp'
if mode in ('dp', 'pg_only'):
	import socket
	with socket.socket() as _s:   # (a free port: a fixed one is still in TIME_WAIT when this tool runs twice within a minute -- the store then waits for its time-out)
		_s.bind(('127.0.0.1', 0))
		_port = _s.getsockname()[1]
	torch.distributed.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_port}', rank=0, world_size=1)
	t = torch.ones(4, device=run.dev)
	torch.distributed.all_reduce(t)
	torch.cuda.synchronize()
su = bench.train3d_setup(run, 16, 16, stage='net', labels=False, dp=False)
m = su['mwl'].model
bucket = None
if mode == 'dp':
	fdist.broadcast_parameters([p for p in m.parameters() if p.is_floating_point()])
	# (as bench.py's ranks: the MLP weights' part of the bucket leaves inside the backward, the latent tables' part behind it)
	bucket = fdist.GradBucket([p for p in m.parameters() if p.requires_grad], early=[p for seq in (m.base, m.mlp_disp, m.mlp_col) for p in seq.parameters()])
	bucket.arm_early(m.base[0].weight)
opt, mwl, opts, flags, batches = su['opt'], su['mwl'], su['opts'], su['flags'], su['batches']
from find_amd.train_utils import sample_latent_vectors
state = dict(i=0)
def step():
	opt.zero_grad(set_to_none=True)
	b = dict(batches[state['i'] % len(batches)]); state['i'] += 1
	b.update(sample_latent_vectors(b, m.latent_vectors_train))
	loss, _ = mwl(b, 0, opts, **flags)
	loss.backward()
	if bucket is not None:
		bucket.allreduce_(async_op=True)
		bucket.wait()
	opt.step()
import ctypes
from find_amd import _lib
g = (ctypes.c_int32 * 5)()
step(); torch.cuda.synchronize()
_lib.check(_lib.lib().find_ctx_stream_groups(_lib.ctx(run.dev), _lib.current_stream(run.dev), ctypes.cast(g, ctypes.c_void_p)), 'find_ctx_stream_groups')
print(f'{mode}: hardware-queue groups [caller, Q, T1, T2, R] = {list(g)}', flush=True)
for rep in range(2):
	print(f'{mode}: {run.timed(step, steps, 10):.3f} ms/step', flush=True)
# host enqueue time of a step with an empty queue (is this mode host-bound?)
if os.environ.get('FIND_DP_HOST', '0') != '0':
	import time
	from find_amd.train_utils import backward_on_this_thread
	with backward_on_this_thread():
		ts = []
		for _ in range(40):
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			step()
			ts.append(time.perf_counter() - t0)
		torch.cuda.synchronize()
	ts.sort()
	print(f'{mode}: host enqueue per step (empty queue): median {ts[len(ts) // 2] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms', flush=True)
if bucket is not None:
	print(f'{mode}: the weights\' part of the bucket ({bucket.n_early * 4} of {bucket.numel * 4} bytes) left inside the backward in {bucket.early_issued} steps', flush=True)
	bucket.close()
if mode in ('dp', 'pg_only'):
	torch.distributed.destroy_process_group()
