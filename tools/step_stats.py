"""Per-step kernel statistics of a rocprofv3 --kernel-trace CSV, with the steps cut at a marker kernel (the optimiser's: 'adam_kernel' for
the network stage, 'sgd_kernel' for the registration stage): the last `count` complete steps before the marker stops appearing are
averaged -- launches per step, average / minimum duration, time per step -- plus the step period and the union of busy time.
python tools/step_stats.py <kernel_trace.csv> <marker> [count=100] [skip_last=2]"""
import collections
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 100
skip_last = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
if len(marks) < skip_last + 3:
	raise SystemExit(f'{marker}: only {len(marks)} launches in the trace')
hi = len(marks) - 1 - skip_last
lo = max(0, hi - count)
n = hi - lo
by = collections.defaultdict(list)
busy = 0
for k in range(lo, hi):
	seg = rows[marks[k] + 1:marks[k + 1] + 1]
	iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
	cs, ce = iv[0]
	for s, e in iv[1:]:
		if s > ce:
			busy += ce - cs
			cs, ce = s, e
		else:
			ce = max(ce, e)
	busy += ce - cs
	for r in seg:
		by[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
period = (int(rows[marks[hi]]['Start_Timestamp']) - int(rows[marks[lo]]['Start_Timestamp'])) / 1e3 / n
total = sum(sum(v) for v in by.values()) / n
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'calls_per_step', 'avg_us', 'min_us', 'max_us', 'us_per_step', 'pct_of_kernel_time'])
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
	w.writerow([name[:120], f'{len(v) / n:.2f}', f'{sum(v) / len(v):.1f}', f'{min(v):.1f}', f'{max(v):.1f}', f'{sum(v) / n:.1f}', f'{100 * sum(v) / n / total:.1f}'])
w.writerow([f'# {n} steps cut at {marker}', f'launches per step {sum(len(v) for v in by.values()) / n:.1f}', f'step period {period:.1f} us (under the tracer)',
			f'GPU busy (union) {busy / 1e3 / n:.1f} us', f'sum of kernel time {total:.1f} us'])
