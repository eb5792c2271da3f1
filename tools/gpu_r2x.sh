#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2x; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --headline-only --steps 20 --warmup 10 > $O/line.json 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 tools/steady_stats.py $O/trace/*/*kernel_trace.csv 20 10 > $O/steady.csv
head -32 $O/steady.csv | cut -c1-170
tail -1 $O/steady.csv
tail -1 $O/line.json | cut -c1-300
