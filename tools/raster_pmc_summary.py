#!/usr/bin/env python3
"""Summary of the rasteriser's profile (tools/prof_raster.sh): every derived figure of the header is COMPUTED here from the counter tables of
the same run -- nothing is typed (the round-3 header disagreed with its own table).
usage: python3 tools/raster_pmc_summary.py <dir written by prof_raster.sh> [wave_face_evaluations]
  wave_face_evaluations: (wave, face) evaluations of one forward raster launch, from `python bench.py --subpaths` (silhouette record); optional."""
import collections
import csv
import glob
import sys

O = sys.argv[1]
evals = float(sys.argv[2]) if len(sys.argv) > 2 else None
import os
KERNELS = ('raster_kernel', 'raster_band_kernel', 'bin_kernel', 'sil_bwd_kernel', 'tie_fix_kernel', 'face_setup_kernel')
dur = collections.defaultdict(list)
for f in glob.glob(O + '/trace/*/*kernel_trace.csv'):
	for r in csv.DictReader(open(f)):
		for k in KERNELS:
			if k in r['Kernel_Name']:
				dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/pmc_*/*/*counter_collection.csv'):
	for r in csv.DictReader(open(f)):
		for k in KERNELS:
			if k in r['Kernel_Name']:
				cnt[k][r['Counter_Name']].append(float(r['Counter_Value']))
avg = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in cnt.items()}
print(f"# Rasteriser evidence: 16 feet x 4 views @{os.environ.get('RENDER_SIZE', '256')}^2, V=6890, F=13776, silhouette only, {os.environ.get('FIND_MESH_KIND', 'latlong')} template, MI355X.")
print('# Command: bash tools/prof_raster.sh <out> full; every figure below is computed by tools/raster_pmc_summary.py from the tables that follow.')
for k in KERNELS:
	if k not in dur and k not in avg:
		continue
	a = avg.get(k, {})
	line = f'# {k}:'
	if dur.get(k):
		d = sorted(dur[k])
		line += f' duration min {d[0]:.1f} us, median {d[len(d) // 2]:.1f} us over {len(d)} launches;'
	if 'FETCH_SIZE' in a and 'WRITE_SIZE' in a:
		# FETCH_SIZE / WRITE_SIZE are in KB; gfx950: FETCH_SIZE counts 64-B requests as 32 B (MI355X_MICROARCH.md): x2
		tr = (2 * a['FETCH_SIZE'] + a['WRITE_SIZE']) * 1024
		line += f' HBM-side traffic 2 x {a["FETCH_SIZE"] * 1024 / 1e9:.3f} + {a["WRITE_SIZE"] * 1024 / 1e9:.3f} = {tr / 1e9:.3f} GB per launch'
		if dur.get(k):
			line += f' ({tr / (sorted(dur[k])[0] * 1e-6) / 1e12:.2f} TB/s at the minimum duration)'
		line += ';'
	if 'SQ_INSTS_VALU' in a:
		line += f' SQ_INSTS_VALU {a["SQ_INSTS_VALU"] / 1e6:.1f} M wave-instructions'
		if evals and k in ('raster_kernel', 'raster_band_kernel'):
			line += f' = {a["SQ_INSTS_VALU"] / evals:.0f} per (wave, face) evaluation ({evals / 1e6:.2f} M evaluations)'
		if 'GRBM_GUI_ACTIVE' in a:
			cyc = a['GRBM_GUI_ACTIVE'] / 8   # summed over the 8 XCDs
			line += f'; x 4 cycles / 1024 SIMDs = {a["SQ_INSTS_VALU"] * 4 / 1024 / cyc * 100:.0f} % of the one-per-four-cycles VALU issue rate over {cyc / 1e6:.2f} M cycles'
		line += ';'
	if 'SQ_THREAD_CYCLES_VALU' in a and 'SQ_INSTS_VALU' in a:
		line += f' execution mask SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64 = {a["SQ_THREAD_CYCLES_VALU"] / a["SQ_INSTS_VALU"] / 64 * 100:.0f} %;'
	if 'SQ_WAIT_ANY' in a and 'SQ_WAVE_CYCLES' in a:
		line += f' waves waiting {a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] * 100:.0f} % of their time;'
	if 'SQ_LDS_BANK_CONFLICT' in a and 'SQ_ACTIVE_INST_LDS' in a:
		line += f' LDS bank conflicts {a["SQ_LDS_BANK_CONFLICT"] / 1e6:.1f} M cycles against {a["SQ_ACTIVE_INST_LDS"] / 1e6:.1f} M of LDS instructions'
	print(line)
print('== kernel durations (us): min / median / launches')
for k in KERNELS:
	if dur.get(k):
		d = sorted(dur[k])
		print(f'{k:20s} {d[0]:9.1f} {d[len(d) // 2]:9.1f} {len(d):5d}')
print('== counters: average per launch')
for k in KERNELS:
	for c, v in sorted(avg.get(k, {}).items()):
		print(f'{k:20s} {c:28s} n={len(cnt[k][c]):3d} avg={v:16.1f}')
