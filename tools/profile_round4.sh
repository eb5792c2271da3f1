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
    python3 "$R/tools/roofline_loop_stats.py" "$O"/roofline/*/*kernel_trace.csv "gemm7_kernel<1, 0>" > "$O/roofline_loop_kernel_stats.txt"; cat "$O/roofline_loop_kernel_stats.txt" ;;
  pmc)
    # PMC passes of the dominant kernel (bf16x3 gemm7 at the C2 shape; tools/prof_x3.py runs gemm4 / gemm6 / gemm7 / gemm5 on the same inputs), one counter set per run
    i=0
    for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
      rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/gemm7_pmc_$i" -- python3 "$R/tools/prof_x3.py" 10 16 6890 > "$O/gemm7_pmc_$i.log" 2>&1
      i=$((i+1))
    done
    { echo "# rocprofv3 --pmc passes of tools/prof_x3.py 10 16 6890 (find_linear_relu_fwd at the C2 shape), averages per launch; FETCH_SIZE / WRITE_SIZE in KB";
      echo "# (gfx950: HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs";
      for k in gemm7_kernel gemm4_kernel; do echo "== $k"; for i in 0 1 2 3 4 5; do python3 "$R/tools/pmc_summary.py" $k "$O"/gemm7_pmc_$i/*/; done; done;
      python3 - "$O" <<'PY'
import csv, glob, sys
for k in ('gemm7_kernel', 'gemm4_kernel'):
	d = []
	for f in glob.glob(sys.argv[1] + '/gemm7_pmc_0/*/*kernel_trace.csv'):
		d += [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if k in r['Kernel_Name']]
	if d: print(f'== {k}: duration under the counters: n={len(d)} avg {sum(d)/len(d):.1f} us min {min(d):.1f} us')
PY
    } > "$O/gemm7_pmc_summary.txt"; cat "$O/gemm7_pmc_summary.txt" ;;
  esac
done
# the raw traces are large: keep the summaries only
rm -rf "$O"/headline/ "$O"/b1/ "$O"/roofline/ "$O"/gemm7_pmc_?/ 2>/dev/null || true
