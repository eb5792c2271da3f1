#!/bin/bash
# A/B on ONE box, alternating: library builds (tools/ab_libs/*.so) x environment switches, the headline loop each time.
# usage: tools/ab_step.sh <rounds> "<lib name or ->:<ENV or ->" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; rounds=$1; shift
cp $R/find_amd/lib/libfind_hip.so /tmp/ab_keep.so
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    lib=${v%%:*}; e=${v#*:}
    if [ "$lib" != "-" ]; then cp $R/tools/ab_libs/$lib.so $R/find_amd/lib/libfind_hip.so; else cp /tmp/ab_keep.so $R/find_amd/lib/libfind_hip.so; fi
    if [ "$e" = "-" ]; then e=""; fi
    env $e python3 $R/bench.py --headline-only --steps 40 --warmup 5 --repeats 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4f ms' % d['ms_per_step'], ['%.4f' % x for x in d['ms_per_step_repeats']])"
  done
done
cp /tmp/ab_keep.so $R/find_amd/lib/libfind_hip.so
