#!/bin/bash
# same-box A/B of the forward render (tools/prof_raster_ablate.py) between this tree and worktrees under _ab/ (git worktree add _ab/<commit> <commit>; python -m find_amd.build there)
for rep in 1 2; do
for d in . _ab/*; do
  [ -f $d/tools/prof_raster_ablate.py ] || continue
  a=$(cd $d && python tools/prof_raster_ablate.py 6890 256 0 0 2>&1 | grep ablate= | tail -1 | sed 's/.*: //;s/ ms.*//')
  b=$(cd $d && python tools/prof_raster_ablate.py 6890 512 0 0 2>&1 | grep ablate= | tail -1 | sed 's/.*: //;s/ ms.*//')
  echo "$d: 256^2 $a ms, 512^2 $b ms"
done
done
