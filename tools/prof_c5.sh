#!/bin/bash
# kernel statistics of the C5 step (16 feet x 50 002-vertex template, MLP fwd + bwd), opt-in fp16 mode and default: usage tools/prof_c5.sh [fp16|fp32 ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=${PROF_OUT:-$R/gpurun_out/r06}; mkdir -p $O
for w in ${*:-fp16}; do
  flag=""; [ $w = fp16 ] && flag="--fp16"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5$w -- python3 $R/bench.py --c5 $flag --steps 10 --warmup 3 --no-cpu-baseline > $O/c5${w}_line.json 2> $O/c5$w.err
  python3 $R/tools/step_stats.py $O/c5$w/*/*kernel_trace.csv head_out_reduce_kernel 6 > $O/c5${w}_step_stats.csv
  python3 $R/tools/step_timeline.py $O/c5$w/*/*kernel_trace.csv head_out_reduce_kernel 3 > $O/c5${w}_step_timeline.txt
  head -30 $O/c5${w}_step_stats.csv | cut -c1-170; tail -1 $O/c5${w}_step_stats.csv
  rm -rf $O/c5$w/
done
