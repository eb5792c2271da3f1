#!/bin/bash
O=gpurun_out/r4u; mkdir -p $O
for i in 1 2 3; do
timeout 300 python bench.py --headline-only >> $O/head_default.json 2>> $O/err.txt
FIND_TUNING="lds_exclusive=1" timeout 300 python bench.py --headline-only >> $O/head_ldsexcl.json 2>> $O/err.txt
FIND_TUNING="dw_lds_free=0" timeout 300 python bench.py --headline-only >> $O/head_dw2full.json 2>> $O/err.txt
done
