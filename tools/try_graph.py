"""Experiment: capture the bench step (forward + backward) in a HIP graph through torch.cuda.CUDAGraph and compare step times.
Result on MI355X (ROCm 7.2, torch 2.10): capture works and replays bit-identically (including the side-stream fork/join inside
find_mlp_bwd), but replay is SLOWER than eager launches (2.91 vs 2.68 ms/step): the eager step is already kernel-bound, and graph
replay loses the cross-stream overlap.  bench.py therefore launches eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda:0')
model, params, step = bench.build_step(dev, 0)

def timeit(fn, n=50):
	for _ in range(5):
		fn()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n):
		fn()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / n * 1e3

print('eager  %.4f ms/step' % timeit(step))

# static grads: keep .grad tensors alive and let autograd accumulate into them inside the graph
N_FEET = bench.N_FEET
idx = torch.arange(N_FEET, device=dev)
for p in params:
	p.grad = torch.zeros_like(p)

def graph_body():
	for p in params:
		p.grad.zero_()
	batch = dict(shapevec_train=model.shapevec[idx], texvec_train=model.texvec[idx], posevec_train=model.posevec[idx], reg_train=model.reg[idx])
	res = model.get_meshes_from_batch(batch, is_train=True)
	loss = (res['verts'] ** 2).sum() + (res['col'] ** 2).sum()
	loss.backward()
	return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
	for _ in range(3):
		graph_body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
	static_loss = graph_body()
torch.cuda.synchronize()
ref = [p.grad.clone() for p in params]
g.replay(); torch.cuda.synchronize()
print('loss', static_loss.item(), 'grad repeatable', all(torch.equal(a, p.grad) for a, p in zip(ref, params)))
print('graph  %.4f ms/step' % timeit(g.replay))
