"""Soak: N headline steps (and N C3 steps) in one process -- step time per block of 200 and device memory in use must stay flat.
python tools/soak_step.py [n_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
run = bench.Run(1)
su = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)
step = su['step']
for _ in range(50):
	step()
torch.cuda.synchronize()
for blk in range(n // 200):
	t0 = time.perf_counter()
	for _ in range(200):
		step()
	torch.cuda.synchronize()
	print(f'steps {blk * 200:5d}..: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step, allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB, '
		  f'reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB', flush=True)
