"""Host enqueue time of the headline step (empty queue) with and without the library's side streams (the event record / wait pairs of its
forks and joins): python tools/r6_host_events.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd import _lib
from find_amd.train_utils import backward_on_this_thread
run = bench.Run(1)
step = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)['step']
with backward_on_this_thread():
	for name, knobs in (('default', {}), ('bwd_streams=0', {'bwd_streams': 0}), ('bwd_streams=0 fwd_streams=0', {'bwd_streams': 0, 'fwd_streams': 0}), ('default', {})):
		for k, v in knobs.items():
			_lib.set_tuning(k, v)
		for _ in range(30):
			step()
		ts = []
		for _ in range(40):
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			step()
			ts.append(time.perf_counter() - t0)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(100):
			step()
		torch.cuda.synchronize()
		bb = (time.perf_counter() - t0) / 100
		ts.sort()
		print(f'{name:32s} host enqueue median {ts[20] * 1e3:.3f} ms min {ts[0] * 1e3:.3f}; back to back {bb * 1e3:.3f} ms', flush=True)
		for k in knobs:
			_lib.set_tuning(k, 1)
