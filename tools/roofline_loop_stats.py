"""The launches that bench.py's `roofline` times, read back from a rocprofv3 --kernel-trace of the same command: the longest run of consecutive
launches of the dominant kernel is the isolated loop of time_dominant_kernel (150 warm-up + 100 timed); statistics of its last 100 launches and
the span they cover (what the HIP events around them measure).  python tools/roofline_loop_stats.py <kernel_trace.csv> [kernel substring] [iters]"""
import csv
import sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'gemm4_kernel<1, 4, 8>'
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
best, cur = [], []
for r in rows:
	if pat in r['Kernel_Name']:
		cur.append(r)
	else:
		if len(cur) > len(best):
			best = cur
		cur = []
if len(cur) > len(best):
	best = cur
run = best[-iters:]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in run]
span = (int(run[-1]['End_Timestamp']) - int(run[0]['Start_Timestamp'])) / 1e3
print(f'kernel: {pat}')
print(f'longest run of consecutive launches: {len(best)}; statistics of its last {len(run)}:')
print(f'  duration per launch: avg {sum(d) / len(d):.2f} us, min {min(d):.2f}, max {max(d):.2f}')
print(f'  first start to last end: {span:.1f} us = {span / len(run):.2f} us per launch (what the HIP events around the loop see)')
