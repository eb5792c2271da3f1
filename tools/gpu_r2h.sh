#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 300 python tools/prof_raster_ablate.py 6890 256 0 1 2 3 4 7 0 2>&1 | grep ablate
timeout 300 python tools/check_flags.py 2>&1 | grep -v amdgpu
