"""Where the host's time goes in one HEADLINE step (batch 16): host enqueue time with an empty queue, then cProfile over N eager steps
(by own time).  python tools/host_profile_b16.py [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
run = bench.Run(1)
step = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)['step']
from find_amd.train_utils import backward_on_this_thread
_ctx = backward_on_this_thread()   # as bench.Run.timed and Trainer run their loops (FIND_AUTOGRAD_THREADS=1: torch's worker thread)
_ctx.__enter__()
for _ in range(40):
	step()
ts = []
for _ in range(30):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	step()
	ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
ts.sort()
print(f'host enqueue per step (empty queue): median {ts[len(ts) // 2] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms')
t0 = time.perf_counter()
for _ in range(steps):
	step()
torch.cuda.synchronize()
print(f'step time back to back: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms')
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
	step()
torch.cuda.synchronize()
pr.disable()
print(f'==== by tottime (per step: divide by {steps})')
pstats.Stats(pr).sort_stats('tottime').print_stats(45)
