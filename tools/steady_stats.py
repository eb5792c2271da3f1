"""Steady-state per-kernel statistics of a rocprofv3 --kernel-trace CSV of `bench.py --headline-only --steps K --warmup W`: the
launches of the W warm-up steps (GPU clocks still ramping, allocator still growing) are dropped, only the last K / (K + W) of every
kernel's launches are averaged.  Prints a CSV (kernel, calls per step, avg us, min us, max us, us per step, % of kernel time).
python tools/steady_stats.py <kernel_trace.csv> <steps> <warmup> [> profiles/rNN_headline_steady_kernel_stats.csv]"""
import collections
import csv
import sys

path, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in rows:
	by[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
out = []
total = 0.0
# the steady window: everything after the end of the W-th launch of the step's last kernel (a kernel launched exactly once per step; the
# optimiser's if there is one).  Kernels with no launch inside it are set-up (stream probes, topology tables) and listed apart.
per_step_once = [n for n, iv in by.items() if len(iv) == steps + warmup]
marker = next((n for n in per_step_once if 'adam' in n or 'sgd' in n), None) or max(per_step_once, key=lambda n: by[n][-1][1], default=None)
t0 = by[marker][warmup - 1][1] if (marker and warmup > 0) else min(s for iv in by.values() for s, _ in iv)
once = []
for name, iv in by.items():
	keep = [(s, e) for s, e in iv if s >= t0]
	if not keep:
		once.append((name, len(iv), sum(e - s for s, e in iv) / 1e3))
		continue
	d = [(e - s) / 1e3 for s, e in keep]
	us_step = sum(d) / steps
	out.append((name, len(keep) / steps, sum(d) / len(d), min(d), max(d), us_step))
	total += us_step
# span of the steady steps: first kept launch to last end
span = (max(e for iv in by.values() for _, e in iv) - t0) / 1e3
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'calls_per_step', 'avg_us', 'min_us', 'max_us', 'us_per_step', 'pct_of_kernel_time'])
for name, ps, avg, mn, mx, us in sorted(out, key=lambda t: -t[5]):
	w.writerow([name[:120], f'{ps:.2f}', f'{avg:.1f}', f'{mn:.1f}', f'{mx:.1f}', f'{us:.1f}', f'{100 * us / total:.1f}'])
for name, n, us in once:
	w.writerow(['# not in the steady steps (set-up)', name[:100], f'{n} launches', f'{us:.1f} us in total'])
w.writerow(['# steady steps', steps, 'warm-up steps dropped', warmup, 'sum of kernel time per step (us)', f'{total:.1f}', f'wall span per step (us): {span / steps:.1f}'])
