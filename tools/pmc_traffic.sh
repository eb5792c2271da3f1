#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the fp16-mode Linear (gemm5), and of the weight-gradient
# launches in both modes (dw2 / dw3 + slab reduce), at the C2 shape.  Run from the repo root through gpurun.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/pmc_traffic
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  export FIND_TUNING=mlp_f16=1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/gemm5_$c -- python3 $R/tools/prof_linear.py 40 16 6890 4 0 > $O/gemm5_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/dw3_$c -- python3 $R/tools/prof_wgrad.py 40 16 6890 > $O/dw3_$c.log 2>&1
  export FIND_TUNING=
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/dw2_$c -- python3 $R/tools/prof_wgrad.py 40 16 6890 > $O/dw2_$c.log 2>&1
done
cd $R
for k in gemm5 dw3 dw2; do for c in FETCH_SIZE WRITE_SIZE; do echo "== $k $c"; python3 tools/pmc_summary.py ${k%%[0-9]}${k##*[a-z]}_kernel $O/${k}_$c/*/ 2>/dev/null; python3 tools/pmc_summary.py reduce_w_kernel $O/${k}_$c/*/ 2>/dev/null; done; done
