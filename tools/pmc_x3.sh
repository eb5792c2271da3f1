#!/bin/bash
# PMC passes of the bf16x3 kernels (gemm7_kernel / dw6_kernel) at the C2 shape: tools/pmc_x3.sh <out-name> [kernel-substring]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; K=${2:-gemm7_kernel}
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/tools/prof_x3.py 10 16 6890 > $O/pmc_$i.log 2>&1
  python3 $R/tools/pmc_summary.py $K $O/pmc_$i/*/
  if [ $i = 0 ]; then python3 - "$O/pmc_0" "$K" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'):
	d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
	if d: print(f'kernel duration under the counters: n={len(d)} avg {sum(d)/len(d):.1f} us min {min(d):.1f} us')
PY
  fi
  i=$((i+1))
done
rm -rf $O/pmc_?/
