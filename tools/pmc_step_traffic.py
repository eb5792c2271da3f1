"""Per-kernel HBM-side traffic of a bench run from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters in KB, FETCH doubled
as MI355X_MICROARCH.md prescribes for gfx950).  python tools/pmc_step_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/>"""
import collections
import csv
import glob
import sys

res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
	f = glob.glob(f'{sys.argv[1]}/{c}/*/*counter_collection.csv')[0]
	agg = collections.defaultdict(list)
	for r in csv.DictReader(open(f)):
		key = (r['Kernel_Name'][:58], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))
		agg[key].append(float(r['Counter_Value']))
	res[c] = agg
names = sorted(res['FETCH_SIZE'], key=lambda k: -sum(res['FETCH_SIZE'][k]) - sum(res['WRITE_SIZE'].get(k, [0])))
print('kernel'.ljust(60), 'grid'.rjust(9), 'n'.rjust(4), 'fetch x2 MB'.rjust(12), 'write MB'.rjust(10))
for k in names[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
	fv = res['FETCH_SIZE'][k]
	wv = res['WRITE_SIZE'].get(k, [0])
	print(k[0].ljust(60), str(k[1]).rjust(9), str(len(fv)).rjust(4), f'{2 * sum(fv) / len(fv) * 1024 / 1e6:12.1f}', f'{sum(wv) / len(wv) * 1024 / 1e6:10.1f}')
