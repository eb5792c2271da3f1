#!/bin/bash
# A/B of two builds of libfind_hip.so (ab_libs/old.so, ab_libs/new.so) on one box: the dominant kernel's isolated loop (tools/prof_linear.py) under the tracer, alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do for v in old new; do
  cp $R/ab_libs/$v.so $R/find_amd/lib/libfind_hip.so
  rm -rf /tmp/ab; rocprofv3 --kernel-trace --output-format csv -d /tmp/ab -- python3 $R/tools/prof_linear.py 200 16 6890 4 0 > /tmp/ab.log 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob('/tmp/ab/*/*kernel_trace.csv'):
	d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'gemm4_kernel' in r['Kernel_Name']]
	d=d[len(d)//2:]; d.sort()
	print('$v', 'n', len(d), 'median %.2f us  min %.2f' % (d[len(d)//2], d[0]))
PY
done; done
cp $R/ab_libs/new.so $R/find_amd/lib/libfind_hip.so
