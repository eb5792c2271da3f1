#!/bin/bash
# same-box A/B of the C3 / C4 end-to-end steps and their kernel times between this tree and the worktrees under _ab/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for d in $R $R/_ab/*; do
  [ -f $d/bench.py ] || continue
  n=$(basename $d)
  (cd $d && python bench.py --c3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n c3 ms_per_step', round(d['ms_per_step'],3))")
  rm -rf /tmp/c3prof_$n; (cd $d && rocprofv3 --kernel-trace --output-format csv -d /tmp/c3prof_$n -- python3 bench.py --c3 --steps 6 --warmup 2 --no-cpu-baseline --no-prime > /dev/null 2>&1)
  f=$(ls /tmp/c3prof_$n/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('find::', '')
    d[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(d, key=lambda k: -sum(d[k])):
    if 'render' in k or 'geom' in k:
        v = d[k][len(d[k]) // 2:]   # (the later half: past the warm-up)
        print(f'   {sys.argv[2]:8s} {k[:60]:60s} n={len(v):3d} mean {sum(v) / len(v):8.1f} us  min {min(v):8.1f}  max {max(v):8.1f}')
PY
done
