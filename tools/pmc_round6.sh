#!/bin/bash
# Round 6 PMC evidence -> gpurun_out/r06/ : gemm7 at the headline shape (all counter sets of tools/pmc_x3.sh) and gemm5 (fp16 mode, fp16-stored
# activations off: find_linear_relu_fwd) at the C5 shape 16 x 50 002 rows: FETCH_SIZE / WRITE_SIZE, one counter per pass (MI355X_MICROARCH.md).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r06; mkdir -p $O
{ echo "# rocprofv3 --pmc passes of tools/prof_x3.py 10 16 6890 (find_linear_relu_fwd at the headline shape, bf16x3), averages per launch; FETCH_SIZE / WRITE_SIZE in KiB"
  echo "# (gfx950: HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs"
  bash $R/tools/pmc_x3.sh r06_pmc_tmp gemm7_kernel; } > $O/gemm7_pmc_summary.txt 2>&1
cd /tmp; export TMPDIR=/tmp
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/prof_linear.py 20 16 50002 4 0 under FIND_TUNING=mlp_f16=1: gemm5_kernel at the C5 shape (16 x 50 002 rows), KiB per launch"
  for c in FETCH_SIZE WRITE_SIZE; do
    export FIND_TUNING=mlp_f16=1
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/g5_$c -- python3 $R/tools/prof_linear.py 20 16 50002 4 0 > $O/g5_$c.log 2>&1
    export FIND_TUNING=
    python3 $R/tools/pmc_summary.py gemm5_kernel $O/g5_$c/*/
  done; } > $O/gemm5_c5_pmc_summary.txt 2>&1
rm -rf $O/g5_*/ $R/gpurun_out/r06_pmc_tmp
tail -30 $O/gemm7_pmc_summary.txt; cat $O/gemm5_c5_pmc_summary.txt
