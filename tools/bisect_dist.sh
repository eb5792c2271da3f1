#!/bin/bash
# run one test in this tree and in the worktrees under _ab/ (same box)
T=${1:-tests/test_gpu_distributed.py::test_two_ranks_of_the_find_model_equal_one_process_on_16_feet}
for d in _ab/* .; do
  [ -d $d/tests ] || continue
  for i in 1 2; do
    (cd $d && GRAFT_REPO_ROOT=$PWD python -m pytest $T -x -q 2>&1 | grep -E "passed|failed|AssertionError: " | cut -c1-200 | sed "s|^|$d: |")
  done
done
