#!/bin/bash
# usage: tools/pmc_gemm.sh <mode> <outdir-name>; separate passes of counters for the linear kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$2/p$i -- python3 $R/tools/prof_linear.py 6 16 6890 $1 0 > $R/gpurun_out/$2/log$i.txt 2>&1
  i=$((i+1))
done
