python -m pytest tests/test_gpu_render.py -q --tb=short -x 2>&1 | tail -3
python tools/bench_paths.py 2>&1 | grep render | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    print(d['workload'], '| fwd %.3f ms  fwd+bwd %.3f ms | tests %d cands %d over %d | early tiles %s fix %s pool %s' % (d['ms_fwd'], d['ms_fwd_bwd'], d['pixel_face_tests'], d['silhouette_candidates'], d['pixels_over_K'], d['tiles_left_early'], d['tie_fixup_pixels'], d['pool_entries']))
"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -- python3 $GRAFT_REPO_ROOT/tools/prof_render.py 256 0 > /tmp/rp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/rp/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print('%-70s calls %s avg %.1f us total %.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
