#!/bin/bash
# round 2: full GPU suite after the find_ctx refactor, the new bench line, kernel trace of the batch-1 step
O=gpurun_out/r2b; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python3 bench.py --train3d-b1 --no-cpu-baseline > $O/b1.json 2> $O/b1.err
find $O/prof_b1 -name "*kernel_stats.csv" | head -2
