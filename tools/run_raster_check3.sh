cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -- python3 $GRAFT_REPO_ROOT/tools/prof_raster_ablate.py 6890 256 $1 > /tmp/rp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/rp/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print('%-70s calls %s avg %.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
