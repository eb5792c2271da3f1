"""Debug: the bench's batch-1 train_3d loop run asynchronously, synchronised every `every` steps.  python tools/debug_b1.py [steps] [every] [stage]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
stage = sys.argv[3] if len(sys.argv) > 3 else 'net'
run = bench.Run(1)
su = bench.train3d_setup(run, 16, 1, stage=stage, labels=True, dp=False)
for i in range(steps):
	su['step']()
	if i % every == every - 1:
		torch.cuda.synchronize()
		print('ok through step', i, flush=True)
torch.cuda.synchronize()
print('done', flush=True)
