"""Where the host's time goes in one train_3d step (batch 1: the step is host-bound; batch 16: the host is within 10 % of the GPU): cProfile
over 200 eager steps, the 45 most expensive functions by own time, then by cumulative time.  python tools/host_profile.py [steps] [batch]"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
run = bench.Run(1)
step = bench.train3d_setup(run, n_items=max(4, batch), batch_size=batch, stage='net', labels=batch == 1, seed=0, dp=False)['step']
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
