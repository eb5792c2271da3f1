#!/bin/bash
# same-box A/B of the isolated bf16x3 weight gradient (tools/prof_wgrad.py, FIND_TUNING=mlp_f16=2) between this tree and the worktrees under _ab/
for d in . _ab/*; do
  [ -f $d/tools/prof_wgrad.py ] || continue
  for sz in "16 6890" "8 6890" "16 50002"; do
    r=$(cd $d && GRAFT_REPO_ROOT=$PWD FIND_TUNING=mlp_f16=2 python tools/prof_wgrad.py 100 $sz 2>&1 | tail -1)
    echo "$d [$sz]: $r"
  done
done
