#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of bench.py (same command the bench line comes from)
#   2. PMC passes for the dominant kernel (find_linear_relu_fwd at the C2 shape), one counter set per pass
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/round
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/tools/prof_linear.py 40 16 6890 4 0 > $O/pmc$i.log 2>&1
  i=$((i+1))
done
tail -1 $O/bench.log
