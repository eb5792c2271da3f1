#!/usr/bin/env python3
"""The headline step (train_3d network stage, 16 feet per step) eager against the same step captured as ONE HIP graph (find_amd/graph.py):
is what is left between the step time and the GPU-busy time the host's?  python tools/headline_graph.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd.graph import GraphedStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
run = bench.Run(1)
su = bench.train3d_setup(run, 16, 16, stage='net', labels=False, dp=False, capturable=True)
for name in ('eager', 'graph', 'eager', 'graph'):
	if name == 'graph':
		gs = GraphedStep(su['mwl'], su['opts'], [su['opt']], **su['flags'])
		step = lambda: gs(su['batches'][0])
	else:
		step = su['step']
	ms = run.timed(step, steps, 10)
	print(f'{name}: {ms:.4f} ms per step', flush=True)
