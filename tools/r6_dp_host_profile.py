"""cProfile of the one-rank data-parallel step's host side (tools/dp_one_rank.py's step): which python calls cost the DP path its host time."""
import cProfile, os, pstats, socket, sys, time
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
fdist.broadcast_parameters([p for p in m.parameters() if p.is_floating_point()])
bucket = fdist.GradBucket([p for p in m.parameters() if p.requires_grad], early=[p for seq in (m.base, m.mlp_disp, m.mlp_col) for p in seq.parameters()])
bucket.arm_early(m.base[0].weight)
opt, mwl, opts, flags, batches = su['opt'], su['mwl'], su['opts'], su['flags'], su['batches']
def step():
	opt.zero_grad(set_to_none=True)
	b = dict(batches[0])
	b.update(sample_latent_vectors(b, m.latent_vectors_train))
	loss, _ = mwl(b, 0, opts, **flags)
	loss.backward()
	bucket.allreduce_(async_op=True)
	bucket.wait()
	opt.step()
with backward_on_this_thread():
	for _ in range(40):
		step()
	torch.cuda.synchronize()
	pr = cProfile.Profile()
	pr.enable()
	for _ in range(100):
		step()
	torch.cuda.synchronize()
	pr.disable()
pstats.Stats(pr).sort_stats('cumtime').print_stats(40)
torch.distributed.destroy_process_group()
