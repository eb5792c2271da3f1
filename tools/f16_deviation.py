import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
from find_amd import functional as F
import test_gpu_mlp_f16 as T
for cfg in [(3, 1002, True), (2, 1002, False), (16, 6890, True)]:
	o32, g32 = T._run_model(*cfg)
	F.set_mlp_precision('fp16')
	o16, g16 = T._run_model(*cfg)
	F.set_mlp_precision('fp32')
	print(cfg, 'out', (o16 - o32).abs().max().item())
	worst = max(((g16[n] - g32[n]).abs().max().item() / max(1e-6, g32[n].abs().max().item()), n) for n in g32)
	print('   worst grad rel-to-max', worst)
