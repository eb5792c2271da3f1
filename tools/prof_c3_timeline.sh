#!/bin/bash
# one C3 step (bench.py --c3) launch by launch under the tracer -> gpurun_out/c3_timeline.txt
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O="$R/gpurun_out"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/c3t; rocprofv3 --kernel-trace --output-format csv -d /tmp/c3t -- python3 "$R/bench.py" --c3 --no-cpu-baseline --steps 12 --warmup 4 > /tmp/c3t.log 2>&1
python3 "$R/tools/step_timeline.py" /tmp/c3t/*/*kernel_trace.csv adam_kernel 4 > "$O/c3_timeline.txt"
python3 "$R/tools/step_stats.py" /tmp/c3t/*/*kernel_trace.csv adam_kernel 8 > "$O/c3_step_stats.csv"
tail -1 "$O/c3_step_stats.csv"
