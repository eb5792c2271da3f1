"""The render records in bench.py's order (c3 with its CPU leg, c4, c3 latlong, c4 latlong), repeats listed: python tools/r6_seq_records.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
run = bench.Run(1)
for kind, c4, cpu in (('uniform', False, True), ('uniform', True, False), ('latlong', False, False), ('latlong', True, False), ('latlong', False, False)):
	a = bench.c3_record(run, 20 if not c4 else 10, 3, cpu, c4=c4, mesh=kind)
	print(kind, 'c4' if c4 else 'c3', round(a['ms_per_step'], 3), a.get('ms_per_step_repeats'), flush=True)
