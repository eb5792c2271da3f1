"""Where the host's time goes in one batch-1 train_3d step (the step is host-bound there): cProfile over 200 eager steps, the 45 most
expensive functions by own time, then by cumulative time.  python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
run = bench.Run(1)
step = bench.train3d_setup(run, n_items=4, batch_size=1, stage='net', labels=True, seed=0)['step']
from find_amd.train_utils import backward_on_this_thread
with backward_on_this_thread():   # (as Trainer / bench run it; the backward's Python functions then show in the profile)
	for _ in range(30):
		step()
	torch.cuda.synchronize()
	pr = cProfile.Profile()
	pr.enable()
	for _ in range(steps):
		step()
	torch.cuda.synchronize()
	pr.disable()
for key in ('tottime', 'cumtime'):
	print(f'==== by {key} (per step: divide by {steps})')
	pstats.Stats(pr).sort_stats(key).print_stats(45)
