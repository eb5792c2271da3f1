#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: average counter value per dispatch for kernels matching a substring."""
import collections
import csv
import glob
import sys

pat = sys.argv[1]
for d in sys.argv[2:]:
	for f in glob.glob(d + '/*counter_collection.csv'):
		agg = collections.defaultdict(list)
		for r in csv.DictReader(open(f)):
			if pat in r['Kernel_Name']:
				agg[r['Counter_Name']].append(float(r['Counter_Value']))
		for k, v in sorted(agg.items()):
			print(f'{k:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}')
