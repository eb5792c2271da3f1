#!/bin/bash
# A/B of environment switches on ONE box, alternating: tools/ab_env.sh <out-name> <rounds> "<bench args>" "ENV_A" "ENV_B" ...
# (each ENV is a space-separated list of VAR=value; "-" = nothing).  Prints ms_per_step of every run.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; rounds=$2; args=$3; shift 3
for r in $(seq 1 $rounds); do
  i=0
  for e in "$@"; do
    if [ "$e" = "-" ]; then e=""; fi
    env $e python3 $R/bench.py $args > $O/run_${i}_$r.json 2> $O/run_${i}_$r.err
    python3 - "$O/run_${i}_$r.json" "$i" "$e" <<'PY'
import json, sys
try:
	d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
	print(f'variant {sys.argv[2]} [{sys.argv[3]}]: {d["ms_per_step"]:.4f} ms  repeats {["%.4f" % x for x in d.get("ms_per_step_repeats", [])]}')
except Exception as ex:
	print(f'variant {sys.argv[2]} [{sys.argv[3]}]: FAILED {ex}')
PY
    i=$((i+1))
  done
done
