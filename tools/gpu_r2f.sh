#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
cd $GRAFT_REPO_ROOT
for c in mlp_free_pts mlp_shared full_step; do timeout 120 python tools/graph_bisect.py $c 2>&1 | tail -1 >> $O/log.txt; done
cat $O/log.txt
timeout 600 python -m pytest tests/test_gpu_train3d.py tests/test_gpu_mlp.py -q -m gpu -x 2>&1 | tail -5
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python3 bench.py --train3d-b1 --no-cpu-baseline --no-graph > $O/b1.json 2> $O/b1.err
find $O/prof_b1 -name "*kernel_stats.csv" | head -2
