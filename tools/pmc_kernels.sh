#!/bin/bash
# Matrix-pipe / wait / LDS counters of dw2 (fp32 dW), gemm5 and dw3 (fp16 mode) at the C2 shape, one counter set per pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/pmc_kernels
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  export FIND_TUNING=
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/dw2_$i -- python3 $R/tools/prof_wgrad.py 30 16 6890 > $O/dw2_$i.log 2>&1
  export FIND_TUNING=mlp_f16=1
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/dw3_$i -- python3 $R/tools/prof_wgrad.py 30 16 6890 > $O/dw3_$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/gemm5_$i -- python3 $R/tools/prof_linear.py 30 16 6890 4 0 > $O/gemm5_$i.log 2>&1
  i=$((i+1))
done
cd $R
for k in dw2 dw3 gemm5; do echo "== ${k}_kernel"; for i in 0 1 2; do python3 tools/pmc_summary.py ${k}_kernel $O/${k}_$i/*/; done; done
