#!/bin/bash
# Rasteriser evidence at the C3 render shape (16 feet x 4 views @256^2, silhouette only): kernel durations, HBM-side traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), VALU / wave occupancy / LDS counters of raster_tile_kernel and sil_bwd_kernel.
# usage: bash tools/prof_raster.sh <outdir-name> [quick|full] [wave_face_evaluations]   (the summary is computed by tools/raster_pmc_summary.py)
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/trace $O/pmc_*; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_render.py 256 0 > $O/trace.log 2>&1
if [ "${2:-}" != "quick" ]; then
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/tools/prof_render.py 256 0 > $O/pmc_$i.log 2>&1
  i=$((i+1))
done
fi
cd $R
python3 $R/tools/raster_pmc_summary.py $O ${3:-}
