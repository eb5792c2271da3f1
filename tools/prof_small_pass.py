"""The small MLP calls alone (bench.time_small_pass): the texture pass (16 x 1000 per-foot points, colour head) and the batch-1 template pass, forward and
forward + backward, HIP events around the Python calls.  Under `rocprofv3 --kernel-trace --stats` with FIND_TUNING=bwd_streams=0 the kernel table gives the
isolated durations of fused6_kernel / fused_chain_kernel (FIND_TUNING=fused6=0), dw4_group, dwpe.   python tools/prof_small_pass.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
f, fb = bench.time_small_pass(dev, 1000, False)
print(f'texture pass (16 x 1000): forward {f * 1e3:.1f} us, forward + backward {fb * 1e3:.1f} us')
f, fb = bench.time_small_pass(dev, bench.N_VERTS, True)
print(f'batch-1 template pass (6890): forward {f * 1e3:.1f} us, forward + backward {fb * 1e3:.1f} us')
