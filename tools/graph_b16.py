"""The headline step (train_3d, batch 16) eager against one HIP-graph replay per step (find_amd/graph.py).  python tools/graph_b16.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd.graph import GraphedStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
run = bench.Run(1)
su = bench.train3d_setup(run, 16, 16, stage='net', labels=False, dp=False, capturable=True)
for rep in range(2):
	print(f'eager (capturable Adam): {run.timed(su["step"], steps, 10):.3f} ms/step', flush=True)
gs = GraphedStep(su['mwl'], su['opts'], [su['opt']], **su['flags'])
state = dict(i=0)
def step():
	gs(su['batches'][state['i'] % len(su['batches'])])
	state['i'] += 1
for rep in range(3):
	print(f'graph replay: {run.timed(step, steps, 10):.3f} ms/step', flush=True)
