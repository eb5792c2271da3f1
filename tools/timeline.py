"""Print one steady-state step of a rocprofv3 --kernel-trace CSV of bench.py as a timeline (start, end, duration, queue, kernel) with
the GPU-busy union and the gaps.  python tools/timeline.py <kernel_trace.csv> [step index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'gather_fwd_kernel' in r['Kernel_Name']][::4]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
i0, i1 = starts[k], starts[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
seg = rows[i0:i1]
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
	if s > ce:
		busy += ce - cs
		cs, ce = s, e
	else:
		ce = max(ce, e)
busy += ce - cs
print(f'step span {(int(rows[i1]["Start_Timestamp"]) - t0) / 1e3:.1f} us, {len(seg)} kernels, GPU busy (union) {busy / 1e3:.1f} us')
for r in seg:
	s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
	print(f'{s:8.1f} {e:8.1f} {e - s:7.1f} q{r["Queue_Id"]} {r["Kernel_Name"][:72]}')
