#!/bin/bash
# kernel statistics of the render loop (tools/prof_render.py <size> <rgb>) under rocprofv3: usage  prof_render_stats.sh <size> <0|1>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O="$R/gpurun_out/render_stats"; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t" -- python3 "$R/tools/prof_render.py" "$1" "$2" > "$O/log" 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$O/t/*/*kernel_stats.csv'):
	for r in list(csv.DictReader(open(f)))[:24]:
		print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  {r['Percentage']}%")
PY
