#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m find_amd.build -j 8 > /dev/null 2>&1
for T in "" "bwd_streams=0" "fwd_streams=0" "reduce_stream=0" "bwd_streams=0,fwd_streams=0" "lds_exclusive=0"; do
  echo "== FIND_TUNING=$T" >> $O/log.txt
  FIND_TUNING=$T timeout 120 python tools/debug_b1.py 300 10 2>&1 | tail -3 >> $O/log.txt
done
echo "== reg stage" >> $O/log.txt
timeout 120 python tools/debug_b1.py 300 10 reg 2>&1 | tail -3 >> $O/log.txt
echo "== graph bisect after nested-fork fix" >> $O/log.txt
for c in mlp_free_pts mlp_shared full_step; do timeout 120 python tools/graph_bisect.py $c 2>&1 | tail -1 >> $O/log.txt; done
cat $O/log.txt
