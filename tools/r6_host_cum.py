"""cProfile of the headline step by CUMULATIVE time, find_amd's own functions only: where the host's 1.3 - 1.7 ms per step are.
python tools/r6_host_cum.py [steps]"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
run = bench.Run(1)
step = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)['step']
from find_amd.train_utils import backward_on_this_thread
with backward_on_this_thread():
	for _ in range(40):
		step()
	torch.cuda.synchronize()
	pr = cProfile.Profile()
	pr.enable()
	for _ in range(steps):
		step()
	torch.cuda.synchronize()
	pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(90)
for ln in out.getvalue().splitlines():
	if 'find_amd' in ln or 'bench.py' in ln or 'ncalls' in ln or 'built-in method' in ln and ('rand' in ln or 'empty' in ln or 'cat' in ln):
		print(ln[:190])
