#!/bin/bash
set -x
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/fused_ab.py ablate > $O/ab.txt 2>&1
for i in 1 2; do
FIND_TUNING="ablate=32" timeout 300 python bench.py --headline-only >> $O/head_old.json 2>> $O/err.txt
timeout 300 python bench.py --headline-only >> $O/head_new.json 2>> $O/err.txt
done
