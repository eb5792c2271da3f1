#!/bin/bash
# kernel statistics of the C4 rank share (16 feet x 4 views @512^2, silhouette + pixel + Chamfer) and of C3: usage tools/prof_c4.sh [c4|c3 ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=${PROF_OUT:-$R/gpurun_out/r06}; mkdir -p $O
for w in ${*:-c4 c3}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 $R/bench.py --$w --steps 12 --warmup 4 --no-cpu-baseline > $O/${w}_line.json 2> $O/$w.err
  python3 $R/tools/step_stats.py $O/$w/*/*kernel_trace.csv adam_kernel 8 > $O/${w}_step_stats.csv
  python3 $R/tools/step_timeline.py $O/$w/*/*kernel_trace.csv adam_kernel 3 > $O/${w}_step_timeline.txt
  head -24 $O/${w}_step_stats.csv | cut -c1-170; tail -1 $O/${w}_step_stats.csv
  rm -rf $O/$w/
done
