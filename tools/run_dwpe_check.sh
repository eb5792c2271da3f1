#!/bin/bash
# A/B of the Fourier layer's weight-gradient kernel (dwpe_kernel vs dw_kernel<AMODE_PE>): MLP parity tests, then headline steps under the tracer.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/dwpe; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_mlp.py tests/test_gpu_pins.py -q -x --tb=short 2>&1 | tail -5
for k in 1 0; do
  FIND_TUNING=dw_pe_lds_free=$k python bench.py --steps 30 --warmup 10 --no-cpu-baseline --headline-only 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.readlines()[-1]); print('dw_pe_lds_free=$k ms_per_step', l['ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --headline-only > $O/trace.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$O/trace/*/*kernel_stats.csv'):
	for r in list(csv.DictReader(open(f)))[:22]:
		print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
