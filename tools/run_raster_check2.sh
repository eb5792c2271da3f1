cd /tmp; export TMPDIR=/tmp
for ab in 0 16 32; do
rm -rf /tmp/rp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -- python3 $GRAFT_REPO_ROOT/tools/prof_render.py 256 0 $ab > /tmp/rp.log 2>&1
echo "== ablate $ab"
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/rp/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    if 'render' in r['Name']: print('%-60s calls %s avg %.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
