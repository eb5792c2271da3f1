#!/usr/bin/env python3
"""C3 / C4 on the latitude-longitude meshes of the configuration against uniform triangulations (find_amd.synthetic.MESH_KIND), in the order
bench.py takes its records: python tools/c3_uniform.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
run = bench.Run(1)
for kind in ('latlong', 'uniform', 'uniform', 'latlong'):
	for c4 in (False, True):
		a = bench.c3_record(run, 12, 3, False, c4=c4, mesh=kind)
		print(f'{"C4" if c4 else "C3"} {kind} {a["ms_per_step"]:.3f} ms', flush=True)
