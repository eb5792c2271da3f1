"""Host cost of one HEADLINE step under host-side switches: enqueue time with an empty queue and the back-to-back step time, alternating.
python tools/host_ab.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
run = bench.Run(1)
step = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)['step']


def measure(tag):
	for _ in range(20):
		step()
	ts = []
	for _ in range(30):
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		step()
		ts.append(time.perf_counter() - t0)
	torch.cuda.synchronize()
	ts.sort()
	t0 = time.perf_counter()
	for _ in range(100):
		step()
	torch.cuda.synchronize()
	print(f'{tag:28s} enqueue median {ts[len(ts) // 2] * 1e3:.3f} min {ts[0] * 1e3:.3f} ms   back to back {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms', flush=True)


for _ in range(rounds):
	measure('default')
	with torch.autograd.set_multithreading_enabled(False):
		measure('autograd single-threaded')
