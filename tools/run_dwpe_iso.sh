#!/bin/bash
# isolated durations of the Fourier weight-gradient kernel: serial backward (no side streams) under the tracer; usage: run_dwpe_iso.sh [extra FIND_TUNING ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/dwpe; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for k in "${@:-dw_pe_lds_free=1}"; do
export FIND_TUNING=$k,bwd_streams=0 FIND_DEFER_WGRADS=0 FIND_OVERLAP_CHAMFER=0
rm -rf $O/iso
rocprofv3 --kernel-trace --output-format csv -d $O/iso -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --headline-only > $O/iso.log 2>&1
echo "== $k"
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$O/iso/*/*kernel_trace.csv'):
	d=collections.defaultdict(list)
	for r in csv.DictReader(open(f)):
		n=r['Kernel_Name']
		if 'dwpe' in n or 'dw_kernel' in n or 'dw4' in n or 'reduce_w' in n:
			d[(n[:50],r['Grid_Size_X'],r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
	for k,v in sorted(d.items()):
		v=sorted(v); print(k, len(v), 'median %.1f min %.1f'%(v[len(v)//2], v[0]))
PY
done
