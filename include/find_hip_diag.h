/* libfind_hip_diag.so: the laboratory build of libfind_hip.so (gfx950), made by find_amd/build.py from the same sources with -DFIND_DIAG.
 * It exports everything include/find_hip.h declares, with the same behaviour, plus what is listed here -- none of which the product carries,
 * neither as host entry points nor in its code objects (csrc/common.h: FIND_ABL / FIND_DBG are compile-time 0 / nullptr there).  Only tools/
 * load it (FIND_DIAG=1 in the environment of find_amd._lib); tests, bench.py and __graft_entry__ run the product library.
 *
 * Additional find_ctx_set keys:
 *   "gemm7"        bf16x3 Linear kernel: 1 = gemm7 (weights in registers, activations through LDS; the product's only one), 0 = gemm6 (weight
 *                  planes in LDS; superseded in round 4, kept for A/B runs: tools/ablate_x3.py)
 *   "x3_abl"       ablation variant of gemm6 (bits 1 no split arithmetic, 2 no LDS fragment reads after a unit's first, 4 no A loads after the
 *                  prologue) / gemm7 (1 no split / LDS writes, 2 no fragment reads, 4 no A loads, 8 no MFMAs): compile-time variants, results WRONG
 *   "dw_lds_free"  additionally 2 / 3 = dw4_wide_kernel / dw2_repro_kernel: waves of 328 / 312 registers, the reproducers of the co-residence
 *                  fault (mlp.hip: wrong weight-gradient elements whenever waves of another kernel share their SIMD); with
 *                  "reduce_exclusive" = 2 every backward pass has wrong elements (tools/probe_lds_fault.py)
 *   "ablate"       every bit: 1 no W staging, 2 no MFMAs, 4 no epilogue, 8 no Fourier features in the fused chains; 1 no DMA issue, 2 no epilogue
 *                  stores in gemm3; 512 / 1024 gemm7 reads / writes one unit's rows only -- results are WRONG under any of these
 *   "dbg"          device pointer to >= 4 * grid uint64: per-workgroup timers of gemm3 / gemm7 (s_memtime ticks; tools/prof_x3.py)
 *   "dw2_verify"   device pointer to 8 + 64 * 8 uint64: dw2_kernel compares every published LDS ring stage with its source in HBM
 * find_ctx_get(ctx, "diag") reads 1. */
#ifndef FIND_HIP_DIAG_H
#define FIND_HIP_DIAG_H
#include "find_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* find_render_switches without the restriction to result-preserving bits.  Further bits (the render is WRONG under each but 64):
 * 1 no candidate lists, 2 no K-nearest pass, 4 no fragment math, 32 no bbox scan in the binning pass, 64 work counters in the flags words
 * (tools/render_stats.py). */
int find_debug_raster_ablate(int64_t bits);

#ifdef __cplusplus
}
#endif
#endif
