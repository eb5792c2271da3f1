#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs: average counter value per dispatch, per kernel NAME (template arguments kept) matching a substring."""
import collections, csv, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
	for f in glob.glob(d + '/*counter_collection.csv'):
		agg = collections.defaultdict(list)
		for r in csv.DictReader(open(f)):
			if pat in r['Kernel_Name']:
				agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
		for (k, c), v in sorted(agg.items()):
			print(f'{k:62s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}')
