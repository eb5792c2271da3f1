#!/bin/bash
set -x
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/tests_all.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
