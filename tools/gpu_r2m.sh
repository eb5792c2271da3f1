#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2m; mkdir -p $O
cat > /tmp/b1.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'] + '/tools')
import torch
from find_amd import _lib, synthetic
import bench
dev = torch.device('cuda:0')
model = synthetic.make_model(6890, train_size=1, val_size=1, device=dev)
lat = synthetic.latents(1, seed=0, device=dev)
params = [p for p in model.parameters() if p.requires_grad]
def step():
	for p in params: p.grad = None
	lv = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
	r = model.get_meshes(shapevec=lv['shapevec'], reg=lv['reg'], texvec=lv['texvec'], posevec=lv['posevec'])
	(r['verts'].sum() + r['col'].sum()).backward()
for _ in range(60): step()
c2 = bench.build_step(dev, 0)[2]
for _ in range(30): c2()
torch.cuda.synchronize()
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 /tmp/b1.py > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f=glob.glob('$O/trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
	print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
