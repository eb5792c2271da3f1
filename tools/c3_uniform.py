#!/usr/bin/env python3
"""C3 / C4 on the latitude-longitude meshes of the configuration against uniform triangulations (find_amd.synthetic.MESH_KIND), in the order
bench.py takes its records: python tools/c3_uniform.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
run = bench.Run(1)
a = bench.c3_record(run, 12, 3, False)
print(f'C3 lat-long {a["ms_per_step"]:.3f} ms', flush=True)
a = bench.c3_record(run, 12, 3, False, c4=True)
print(f'C4 lat-long {a["ms_per_step"]:.3f} ms', flush=True)
for i in range(2):
	b = bench.c3_record_uniform_meshes(run, 12, 3)
	print(f'C3 uniform {b["ms_per_step"]:.3f} ms', flush=True)
b = bench.c3_record_uniform_meshes(run, 12, 3, c4=True)
print(f'C4 uniform {b["ms_per_step"]:.3f} ms', flush=True)
b = bench.c3_record_uniform_meshes(run, 12, 3)
print(f'C3 uniform {b["ms_per_step"]:.3f} ms', flush=True)
