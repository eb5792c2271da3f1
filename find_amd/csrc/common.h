// Shared helpers for libfind_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

#include "find_hip.h"

namespace find {

void set_error(const char* fmt, ...);
extern int g_raster_ablate;  // render.hip / geom.hip switches (find_render_switches; the diagnostics build's find_debug_raster_ablate)

// Product and laboratory are two libraries built from the same sources (find_amd/build.py): libfind_hip.so carries the kernels the path runs
// and the switches that leave results unchanged; libfind_hip_diag.so (-DFIND_DIAG, include/find_hip_diag.h) adds the fault reproducers, the
// superseded A/B kernels, the per-workgroup timers and the ablation bits under which results are WRONG.  Device code asks through
// FIND_ABL / FIND_DBG, which are compile-time 0 / nullptr in the product, so none of that code is in its code objects.
#ifdef FIND_DIAG
#define FIND_DIAG_ON 1
#else
#define FIND_DIAG_ON 0
#endif
#define FIND_ABL(bits, mask) (FIND_DIAG_ON && ((bits) & (mask)))
#define FIND_DBG(ptr) (FIND_DIAG_ON ? (ptr) : nullptr)
// result-preserving switches, settable in both builds
constexpr int MLP_SWITCHES = 16 | 32 | 128;              // "ablate": no s_setprio in gemm4, every column block in the Fourier dW, 32-row fused tiles
constexpr int RASTER_SWITCHES = 8 | 16 | 256 | 512 | 1024 | 2048 | 4096;  // no early exit, unsorted tile lists, tiny list pool; Chamfer: all pairs / grid from 64 points; 2048 / 4096: band / list rasteriser at every size

inline int check_launch(const char* what) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		set_error("%s: %s", what, hipGetErrorString(e));
		return FIND_ELAUNCH;
	}
	return FIND_OK;
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace (256-byte aligned carves).
struct Carver {
	char* base;
	int64_t off;
	explicit Carver(void* p) : base(reinterpret_cast<char*>(p)), off(0) {}
	template <typename T>
	T* take(int64_t n) {
		T* r = reinterpret_cast<T*>(base + off);
		off += align_up(n * (int64_t)sizeof(T), 256);
		return r;
	}
};

}  // namespace find

#define FIND_REQUIRE(cond, ...)            \
	do {                                   \
		if (!(cond)) {                     \
			find::set_error(__VA_ARGS__);  \
			return FIND_EINVAL;            \
		}                                  \
	} while (0)

#define FIND_LAUNCH_CHECK(what)                    \
	do {                                           \
		int _rc = find::check_launch(what);        \
		if (_rc != FIND_OK) return _rc;            \
	} while (0)

// Workgroup barrier for data exchanged through LDS ONLY: waits for this wave's LDS operations, not for its global loads and stores.
// (__syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: in a loop that keeps global loads in flight across iterations -- the
// prefetch of a software pipeline -- it drains them at every barrier and exposes the whole memory latency once per iteration: a quarter
// of gemm7's time and more of dw6's before this.)
// Buffer stores of more than 64 bits go out WITHOUT an SGPR offset (the offset is folded into the VGPR one).  gfx950 reads the data registers
// of such a store after it has issued: a VALU instruction right behind it that overwrites one of them corrupts the stored dword -- 0.11 % of
// the stores under a backed-up memory pipeline with an SGPR offset, 21 % without (tools/store_hazard_probe.hip; one wait state cures both).
// LLVM's hazard recognizer inserts that wait state only for the form without an SGPR offset (GCNHazardRecognizer::createsVALUHazard follows
// the ISA manual's rule), so that is the form to emit; tools/check_store_hazard.py lints the built code objects for the unprotected one
// (tests/test_host_api.py runs it).  Found in round 5: gemm7's masked-dX stores, single elements wrong, differently from run to run.
typedef unsigned find_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_b128(find_u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voff, int off) {
	__builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff + off, 0, 0);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wave-level sum over 64 lanes (result valid in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}
