"""Eager training loops with and without the deferred weight-gradient join (find_amd.functional.DEFER_WGRAD_JOIN): largest parameter
distance after the same steps, against the distance between two runs of the SAME setting (float atomics in the sampling backward make
any two runs differ; Adam turns that into a visible fraction of a step on weights whose gradient is a cancellation)."""
import os, sys, pathlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import find_amd.functional as FN
import test_gpu_trainloop as T

def run(defer):
	FN.DEFER_WGRAD_JOIN = defer
	tr, params, modes = T._trainer_run(False, pathlib.Path(tempfile.mkdtemp()))
	return params

a, b, c, d = run(True), run(True), run(False), run(False)
def dist(x, y):
	return max((x[n] - y[n]).abs().max().item() for n in x)
print('defer vs defer   %.3e' % dist(a, b))
print('plain vs plain   %.3e' % dist(c, d))
print('defer vs plain   %.3e' % dist(a, c), ' %.3e' % dist(b, d))
