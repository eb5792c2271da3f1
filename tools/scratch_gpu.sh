#!/bin/bash
set -x
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/bench_paths.py > $O/subpaths.jsonl 2> $O/subpaths.err
