#!/bin/bash
# Round 6 evidence -> gpurun_out/r06/ (copy what is to be judged into profiles/).  Usage: tools/profile_round5.sh [headline|roofline|dw6 ...]
#   headline  rocprofv3 --kernel-trace --stats of the headline loop alone (bench.py --headline-only): kernel stats, steady-state statistics,
#             per-step statistics cut at the optimiser kernel, one step launch by launch
#   roofline  the bench command itself (headline + the isolated loop its `roofline` object times) under the tracer: the timed launches read back
#   dw6       the bf16x3 weight-gradient kernel alone (tools/prof_wgrad.py with mlp_f16=2 at the headline shape): per-launch statistics of dw6_kernel
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O="$R/gpurun_out/r06"; mkdir -p "$O"
what=${*:-headline roofline dw6}
cd /tmp; export TMPDIR=/tmp
for w in $what; do
  case $w in
  headline)
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/headline" -- python3 "$R/bench.py" --headline-only --steps 30 --warmup 10 > "$O/headline_line.json" 2> "$O/headline.err"
    python3 "$R/tools/steady_stats.py" "$O"/headline/*/*kernel_trace.csv 25 adam_kernel > "$O/headline_steady_kernel_stats.csv"
    python3 "$R/tools/step_stats.py" "$O"/headline/*/*kernel_trace.csv adam_kernel 25 > "$O/headline_step_stats.csv"
    python3 "$R/tools/step_timeline.py" "$O"/headline/*/*kernel_trace.csv adam_kernel 6 > "$O/headline_step_timeline.txt"
    cp "$O"/headline/*/*kernel_stats.csv "$O/headline_kernel_stats.csv"
    head -16 "$O/headline_steady_kernel_stats.csv" | cut -c1-150; tail -1 "$O/headline_step_stats.csv" ;;
  roofline)
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/roofline" -- python3 "$R/bench.py" --no-records --no-cpu-baseline > "$O/roofline_line.json" 2> "$O/roofline.err"
    python3 "$R/tools/roofline_loop_stats.py" "$O"/roofline/*/*kernel_trace.csv "gemm7_kernel<1, 0, false, false>" > "$O/roofline_loop_kernel_stats.txt"; cat "$O/roofline_loop_kernel_stats.txt" ;;
  dw6)
    export FIND_TUNING=mlp_f16=2
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/dw6" -- python3 "$R/tools/prof_wgrad.py" 100 16 6890 > "$O/dw6.log" 2>&1
    unset FIND_TUNING
    { echo "# rocprofv3 --kernel-trace of tools/prof_wgrad.py 100 16 6890 under FIND_TUNING=mlp_f16=2 (find_linear_wgrad at the headline shape, bf16x3): the isolated loop";
      python3 - "$O" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/dw6/*/*kernel_trace.csv'):
	rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for k in ('dw6_kernel', 'reduce_w_kernel'):
	d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if k in r['Kernel_Name']][-100:]
	if d:
		fl = 6 * 2.0 * 16 * 6890 * 256 * 256
		extra = f'; executed bf16 products {fl / 1e9:.1f} GFLOP per launch = {fl / (sum(d) / len(d) * 1e-6) / 1e12:.0f} TFLOP/s = {fl / (sum(d) / len(d) * 1e-6) / 2.5e15:.3f} of the 2.5 PFLOP/s dense bf16 peak' if k == 'dw6_kernel' else ''
		print(f'{k}: last {len(d)} launches: avg {sum(d) / len(d):.2f} us, min {min(d):.2f}, max {max(d):.2f}{extra}')
PY
      cat "$O/dw6.log" | tail -3; } > "$O/dw6_loop_kernel_stats.txt"; cat "$O/dw6_loop_kernel_stats.txt" ;;
  esac
done
# the raw traces are large: keep the summaries only
rm -rf "$O"/headline/ "$O"/roofline/ "$O"/dw6/ 2>/dev/null || true
