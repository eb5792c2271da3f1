#!/bin/bash
# round 2, first GPU pass: new tests, the new bench line, kernel trace of the batch-1 step
O=gpurun_out/r2a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train3d.py tests/test_gpu_distributed.py tests/test_gpu_optim.py "tests/test_gpu_mlp_f16.py" -q -m gpu -s 2>&1 | tail -60 > $O/pytest_new.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python3 bench.py --train3d-b1 --no-cpu-baseline > $O/b1.json 2> $O/b1.err
ls -R $O | head -30
tail -5 $O/pytest_new.log
