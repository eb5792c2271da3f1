#!/bin/bash
# PMC passes of the bf16x3 weight-gradient kernel alone (tools/prof_wgrad.py under FIND_TUNING=mlp_f16=2): usage tools/pmc_dw6.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
export FIND_TUNING=mlp_f16=2
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/prof_wgrad.py 10 16 6890 > $O/log$i.txt 2>&1
  i=$((i+1))
done
{ echo "# rocprofv3 --pmc passes of tools/prof_wgrad.py 10 16 6890 under FIND_TUNING=mlp_f16=2 (find_linear_wgrad, bf16x3, headline shape): averages per launch of dw6_kernel";
  echo "# GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_* over all SIMDs / CUs";
  for i in 0 1 2 3 4 5; do python3 $R/tools/pmc_summary.py dw6_kernel $O/p$i/*/; done
  python3 - "$O" <<'PY'
import csv, glob, sys
d = []
for f in glob.glob(sys.argv[1] + '/p0/*/*kernel_trace.csv'):
	d += [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'dw6_kernel' in r['Kernel_Name']]
if d: print(f'== dw6_kernel: duration under the counters: n={len(d)} avg {sum(d)/len(d):.1f} us min {min(d):.1f} us')
PY
} > $O/../$1_summary.txt
rm -rf $O/p?/
cat $O/../$1_summary.txt
