"""Reference point for the HBM-bound kernels: a plain device copy (torch) of the same bytes as one fp16-mode Linear launch
(rows x 1 KB read + rows x 1 KB written).  python tools/copy_bw.py [rows]"""
import sys

import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 6890
x = torch.randn(rows, 256, device='cuda')
y = torch.empty_like(x)
for _ in range(50):
	y.copy_(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
	y.copy_(x)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / 200
print(f'copy {rows} x 1 KB: {ms * 1e3:.1f} us = {2 * rows * 1024 / (ms * 1e-3) / 1e12:.2f} TB/s (read + written)')
