#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/graph_bisect.py > $O/bisect.log 2>&1
cat $O/bisect.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_train3d.py::test_graphed_step_equals_eager_steps 2>&1 | tail -40 > $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py --no-graph > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python3 bench.py --train3d-b1 --no-cpu-baseline --no-graph > $O/b1.json 2> $O/b1.err
find $O/prof_b1 -name "*kernel_stats.csv" | head -2
