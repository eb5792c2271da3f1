#!/usr/bin/env python3
"""C3 / C4 on the latitude-longitude meshes of the configuration against uniform triangulations (find_amd.synthetic.MESH_KIND): python tools/c3_uniform.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
run = bench.Run(1)
for c4 in (False, True):
	a = bench.c3_record(run, 12, 3, False, c4=c4)
	b = bench.c3_record_uniform_meshes(run, 12, 3, c4=c4)
	print(f'{"C4 rank share" if c4 else "C3"}: lat-long {a["ms_per_step"]:.3f} ms, uniform {b["ms_per_step"]:.3f} ms', flush=True)
