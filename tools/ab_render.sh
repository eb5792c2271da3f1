#!/bin/bash
# A/B of two builds of libfind_hip.so (ab_libs/old.so, ab_libs/new.so) on one box, alternating: render sub-paths, the C3 and the C4 step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do for v in old new; do
  cp $R/ab_libs/$v.so $R/find_amd/lib/libfind_hip.so
  python bench.py --subpaths --no-cpu-baseline 2>/dev/null | head -3 | python3 -c "
import sys,json
for ln in sys.stdin:
    d=json.loads(ln); print('$v', d['workload'][:28], round(d['ms_fwd'],3), round(d['ms_fwd_bwd'],3))"
  python bench.py --c3 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print('$v c3', round(l['ms_per_step'],3))"
  python bench.py --c4 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print('$v c4', round(l['ms_per_step'],3))"
done; done
cp $R/ab_libs/new.so $R/find_amd/lib/libfind_hip.so
