import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from find_amd import synthetic, _lib
dev = torch.device('cuda:0')
mode = sys.argv[1]
N, V = 16, 6890
model = synthetic.make_model(V, train_size=N, val_size=2, device=dev)
gen = torch.Generator().manual_seed(3)
pos1 = ((torch.rand(1, V, 3, generator=gen) * 2 - 1) * 0.1).cuda()
pos = pos1 if mode.startswith('s') else pos1.expand(N, -1, -1).contiguous()
lat = [(torch.randn(N, 100, generator=gen) * 0.1).cuda().requires_grad_(True) for _ in range(3)]
res = model(pos, shapevec=lat[0], texvec=lat[1], posevec=lat[2])
torch.cuda.synchronize(); print(mode, 'fwd ok', flush=True)
if mode.endswith('b'):
	(res['disp'].sum() + res['col'].sum()).backward()
	torch.cuda.synchronize(); print(mode, 'bwd ok', flush=True)
