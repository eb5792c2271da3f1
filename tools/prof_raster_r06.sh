#!/bin/bash
# Round-6 rasteriser evidence: the kernel each image size takes (list kernel @256^2, band kernel @512^2) on both template triangulations
# (uniform = what the c3 / c4 records run on; latlong = the pole-sliver stress meshes): durations, HBM-side traffic (FETCH / WRITE in separate passes),
# VALU issue and execution mask.  usage: bash tools/prof_raster_r06.sh   -> gpurun_out/r06/raster_<kind>_<size>.txt
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
for kind in uniform latlong; do for size in 256 512; do
  export FIND_MESH_KIND=$kind RENDER_SIZE=$size
  O=$R/gpurun_out/r06/raster_${kind}_$size; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_render.py $size 0 > $O/trace.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/tools/prof_render.py $size 0 > $O/pmc_$i.log 2>&1
    i=$((i+1))
  done
  python3 $R/tools/raster_pmc_summary.py $O > $R/gpurun_out/r06/raster_${kind}_$size.txt
  head -8 $R/gpurun_out/r06/raster_${kind}_$size.txt | cut -c1-400
  rm -rf $O
done; done
