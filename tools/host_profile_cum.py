import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
from find_amd.train_utils import backward_on_this_thread
run = bench.Run(1)
step = bench.train3d_setup(run, bench.N_FEET, bench.N_FEET, stage='net', labels=False, seed=0)['step']
with backward_on_this_thread():
	for _ in range(40): step()
	torch.cuda.synchronize()
	pr = cProfile.Profile(); pr.enable()
	for _ in range(100): step()
	torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(60)
