"""How long the host needs to enqueue one headline step (forward + backward launches, autograd bookkeeping) when the GPU queue is
empty -- the floor the step time cannot go under however fast the kernels are.  python tools/cpu_enqueue_time.py [fp16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from find_amd import functional as F  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == 'fp16':
	F.set_mlp_precision('fp16')
dev = torch.device('cuda:0')
model, params, step = bench.build_step(dev, seed=0)
for _ in range(5):
	step()
ts = []
for _ in range(20):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	step()
	ts.append(time.perf_counter() - t0)
	torch.cuda.synchronize()
ts.sort()
print(f'host enqueue time per step: median {ts[len(ts) // 2] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
	step()
torch.cuda.synchronize()
print(f'step time, 30 back-to-back steps: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms')
