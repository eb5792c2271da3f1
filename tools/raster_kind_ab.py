#!/usr/bin/env python3
"""C3 / C4 with the candidate-list rasteriser, the band rasteriser and the default choice (band from 384^2 on), same process:
python tools/raster_kind_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from find_amd import _lib
run = bench.Run(1)
for c4 in (False, True):
	for name, bits in (('list', 4096), ('band', 2048), ('default', 0), ('list', 4096), ('band', 2048)):
		_lib.set_tuning('raster_ablate', bits)
		r = bench.c3_record(run, 12, 3, False, c4=c4)
		print(f'{"C4 rank share" if c4 else "C3"} [{name}]: {r["ms_per_step"]:.3f} ms', flush=True)
_lib.set_tuning('raster_ablate', 0)
