"""Time the bench step under different find_ctx knob values (_lib.set_tuning): python tools/tune_step.py key v1 v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from find_amd import _lib

key = sys.argv[1].encode()
vals = [int(v) for v in sys.argv[2:]]
dev = torch.device('cuda:0')
model, params, step = bench.build_step(dev, 0)
L = _lib.lib()
for v in vals:
	_lib.set_tuning(key.decode() if isinstance(key, bytes) else key, v)
	for _ in range(5):
		step()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(30):
		step()
	e1.record()
	e1.synchronize()
	print(f'{key.decode()}={v}: {e0.elapsed_time(e1) / 30:.4f} ms/step')
