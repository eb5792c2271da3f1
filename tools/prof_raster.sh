#!/bin/bash
# Rasteriser evidence at the C3 render shape (16 feet x 4 views @256^2, silhouette only): kernel durations, HBM-side traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), VALU / wave occupancy / LDS counters of raster_tile_kernel and sil_bwd_kernel.
# usage: bash tools/prof_raster.sh <outdir-name> [quick]
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_render.py 256 0 > $O/trace.log 2>&1
if [ "${2:-}" != "quick" ]; then
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/tools/prof_render.py 256 0 > $O/pmc_$i.log 2>&1
  i=$((i+1))
done
fi
cd $R
python3 - <<PY
import csv, glob, collections
O='$O'
f=glob.glob(O+'/trace/*/*kernel_stats.csv')
if f:
	print('== kernel stats (6 fwd+bwd passes incl. the first)')
	for r in list(csv.DictReader(open(f[0])))[:8]:
		print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f} us")
for d in sorted(glob.glob(O+'/pmc_*/')):
	for f in glob.glob(d+'*/*counter_collection.csv'):
		agg=collections.defaultdict(list)
		for r in csv.DictReader(open(f)):
			for pat in ('raster_kernel','bin_kernel','sil_bwd_kernel','tie_fix_kernel','face_setup_kernel'):
				if pat in r['Kernel_Name']:
					agg[(pat,r['Counter_Name'])].append(float(r['Counter_Value']))
		for (pat,c),v in sorted(agg.items()):
			print(f'{pat:20s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):16.1f}')
PY
