#!/bin/bash
# Round 4 evidence -> gpurun_out/r04/ (copy what is to be judged into profiles/).  Usage: tools/profile_round4.sh [headline|b1|roofline|pmc ...]
#   headline  rocprofv3 --kernel-trace --stats of the headline loop alone (bench.py --headline-only): kernel stats, steady-state statistics
#             (warm-up launches dropped), per-step statistics cut at the optimiser kernel, one step launch by launch
#   b1        the same for the batch-1 network-stage record (bench.py --train3d-b1 --b1-only train3d_b1 --no-graph)
#   roofline  the bench command itself (headline + the isolated loop its `roofline` object times) under the tracer: the timed launches read back
#   pmc       PMC passes of the dominant kernel (find_linear_relu_fwd at the C2 shape): MFMA busy / FETCH_SIZE / WRITE_SIZE, one counter set per run
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O="$R/gpurun_out/r04"; mkdir -p "$O"
what=${*:-headline b1 roofline pmc}
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
  b1)
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/b1" -- python3 "$R/bench.py" --train3d-b1 --b1-only train3d_b1 --no-graph --no-cpu-baseline > "$O/b1_line.json" 2> "$O/b1.err"
    python3 "$R/tools/step_stats.py" "$O"/b1/*/*kernel_trace.csv adam_kernel 200 > "$O/b1_step_stats.csv"; tail -1 "$O/b1_step_stats.csv" ;;
  roofline)
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/roofline" -- python3 "$R/bench.py" --no-records --no-cpu-baseline > "$O/roofline_line.json" 2> "$O/roofline.err"
    python3 "$R/tools/roofline_loop_stats.py" "$O"/roofline/*/*kernel_trace.csv > "$O/roofline_loop_kernel_stats.txt"; cat "$O/roofline_loop_kernel_stats.txt" ;;
  pmc)
    i=0
    for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/gemm4_pmc_$i" -- python3 "$R/tools/prof_linear.py" 40 16 6890 4 0 > "$O/gemm4_pmc_$i.log" 2>&1
      i=$((i+1))
    done
    for i in 0 1 2; do python3 "$R/tools/pmc_summary.py" gemm4_kernel "$O"/gemm4_pmc_$i/*/; done > "$O/gemm4_pmc_summary.txt"; cat "$O/gemm4_pmc_summary.txt" ;;
  esac
done
# the raw traces are large: keep the summaries only
rm -rf "$O"/headline/ "$O"/b1/ "$O"/roofline/ "$O"/gemm4_pmc_?/ 2>/dev/null || true
