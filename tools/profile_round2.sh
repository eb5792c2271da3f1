#!/bin/bash
# Round 2 evidence -> gpurun_out/r02/ (copy what is to be judged into profiles/):
#   1. python bench.py (the driver's command) -> bench_line.json
#   2. rocprofv3 --kernel-trace --stats of the headline loop alone (bench.py --headline-only) -> kernel stats of that command, and the
#      steady-state statistics (warm-up launches dropped, tools/steady_stats.py)
#   2b. the same for the batch-1 records (bench.py --train3d-b1): per-step statistics cut at the optimiser kernel (tools/step_stats.py)
#   2c. tools/step_timeline.py: one step of the headline trace launch by launch (what runs beside what)
#   2d. the bench command itself (headline + the isolated loop its `roofline` times) under the tracer: tools/roofline_loop_stats.py reads the
#       timed launches back from the trace (their average must agree with roofline.avg_kernel_ms of the same run, roofline_line.json)
#   3. PMC passes of the dominant kernel (find_linear_relu_fwd at the C2 shape): MFMA busy / traffic
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python3 $R/bench.py --headline-only --steps 30 --warmup 10 > $O/headline_line.json 2> $O/headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python3 $R/bench.py --no-records --no-cpu-baseline > $O/roofline_line.json 2> $O/roofline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -- python3 $R/bench.py --train3d-b1 --no-graph --no-cpu-baseline > $O/b1_line.json 2> $O/b1.err
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/gemm4_pmc_$i -- python3 $R/tools/prof_linear.py 40 16 6890 4 0 > $O/gemm4_pmc_$i.log 2>&1
  i=$((i+1))
done
cd $R
python3 tools/steady_stats.py $O/headline/*/*kernel_trace.csv 30 10 > $O/headline_steady_kernel_stats.csv
python3 tools/step_stats.py $O/headline/*/*kernel_trace.csv adam_kernel 25 > $O/headline_step_stats.csv
python3 tools/step_timeline.py $O/headline/*/*kernel_trace.csv adam_kernel 6 > $O/headline_step_timeline.txt
python3 tools/step_stats.py $O/b1/*/*kernel_trace.csv adam_kernel 200 > $O/b1_step_stats.csv
python3 tools/roofline_loop_stats.py $O/roofline/*/*kernel_trace.csv > $O/roofline_loop_kernel_stats.txt
rm -rf $O/roofline
cp $O/headline/*/*kernel_stats.csv $O/headline_kernel_stats.csv
for i in 0 1 2; do python3 tools/pmc_summary.py gemm4_kernel $O/gemm4_pmc_$i/*/; done > $O/gemm4_pmc_summary.txt
head -12 $O/headline_steady_kernel_stats.csv | cut -c1-150
tail -1 $O/headline_step_stats.csv; tail -1 $O/b1_step_stats.csv
cat $O/gemm4_pmc_summary.txt
tail -1 $O/bench_line.json | cut -c1-400
