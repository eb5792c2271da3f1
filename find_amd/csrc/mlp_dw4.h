// dw4_kernel: weight gradient  dW[n,k] = sum_rows dZ[row,n] * X[row,k]  (256 x 256 output, exact fp32 MFMA) WITHOUT LDS and within 256
// registers per lane: the default kernel for every 256 x 256 weight gradient since round 2.  (dw2_kernel's 128 x 128 wave tiles need 312
// registers, and waves above 256 registers are corrupted when other kernels' waves share their SIMD: mlp.hip, "Co-residence fault".  This
// kernel may share its SIMDs with anything: 0 wrong tensors in 1450 stress passes.)
// The operand layout of dw2 read straight from global memory.  Lane l of a k-pair (two rows of dZ and X) loads 16 bytes of dZ at
// columns 4(l&31).. of row 2t + (l>>5) and 8 bytes of X at columns 2(l&31)..; component ja of the first feeds MFMA row-block ja,
// component jb of the second column-block jb, so accumulator (ja, jb) holds the outputs n = 4i + ja, k = 2j + jb (i, j = MFMA row /
// column).  Eight waves = two per SIMD: wave (wn, wk) owns n in [128 wn, +128), k in [64 wk, +64): 4 x 2 accumulator blocks.  No LDS,
// no barrier, no DMA: operands are prefetched DW4_PD k-pairs ahead through a ring of 8 register slots.  Same split geometry and slab
// layout as dw2 (Dw2Args), so reduce_w_kernel sums the slabs unchanged.
#pragma once
#include "mlp_dw2.h"

namespace find {
namespace mlp {

constexpr int DW4_PD = 6;   // prefetch distance in k-pairs (ring of 8)

template <int NB>   // 32-column blocks of k per wave: 2 (eight waves, the product) or 4 (four waves with dw2's 128 x 128 tiles: diagnosis)
__device__ __forceinline__ void dw4_body(const Dw2Args& g, const int split) {
	typedef float bvec __attribute__((ext_vector_type(NB)));
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wn = NB == 2 ? wave >> 2 : wave >> 1, wk = NB == 2 ? wave & 3 : wave & 1;
	const int li = lane & 31, fh = lane >> 5;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, g.chunks_per_foot);
	const int total = max(q1 - q0, 0);            // whole 16-row chunks
	const int tail = (sidx == g.spf - 1) ? g.tail_rows : 0;
	float* const pw = g.pw + (int64_t)split * 65536;
	float* const pb = g.pb ? g.pb + (int64_t)split * 256 : nullptr;
	const float* const zfoot = g.dz + (int64_t)foot * g.dz_foot_stride;
	const float* const xfoot = g.x + (int64_t)foot * g.x_foot_stride;

	f32x16 acc[4][NB];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < NB; ++b)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
	typedef float v2f __attribute__((ext_vector_type(2)));
	v2f bs01 = {0.f, 0.f}, bs23 = {0.f, 0.f};   // column sums of the dZ values this lane loads
	const bool do_bias = pb != nullptr && wk == 0;

	const float* zp = zfoot + ((int64_t)q0 * 16 + fh) * 256 + wn * 128 + 4 * li;
	const float* xp = xfoot + ((int64_t)q0 * 16 + fh) * 256 + wk * (32 * NB) + NB * li;
	const int nkp = total * 8;   // k-pairs of the whole chunks

	auto mfma8 = [&](const float4& a, const bvec& b) {
#pragma unroll
		for (int ja = 0; ja < 4; ++ja) {
			const float av = ja == 0 ? a.x : (ja == 1 ? a.y : (ja == 2 ? a.z : a.w));
#pragma unroll
			for (int jb = 0; jb < NB; ++jb) acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[jb], acc[ja][jb], 0, 0, 0);
		}
	};

	if (nkp > 0) {
		float4 ra[8];
		bvec rb[8];
		// k-pair t lives at row offset 2t: 512 floats further.  Loads past the end of the run re-read its last k-pair (never used).
		const int last = nkp - 1;
#pragma unroll
		for (int t = 0; t < DW4_PD; ++t) {
			const int tt = min(t, last);
			ra[t] = *reinterpret_cast<const float4*>(zp + (int64_t)tt * 512);
			rb[t] = *reinterpret_cast<const bvec*>(xp + (int64_t)tt * 512);
			// (in program order: left to the scheduler, the slots the loop reads first were loaded last, and the wait counter of the loop's first step
			//  -- the merge of this path and the back edge -- became vmcnt(2): every load in flight drained once per eight k-pairs.  Round 3.)
			__builtin_amdgcn_sched_barrier(0);
		}
		for (int t0 = 0; t0 < nkp; t0 += 8) {
#pragma unroll
			for (int s = 0; s < 8; ++s) {
				{
					const int tt = min(t0 + s + DW4_PD, last);
					ra[(s + DW4_PD) & 7] = *reinterpret_cast<const float4*>(zp + (int64_t)tt * 512);
					rb[(s + DW4_PD) & 7] = *reinterpret_cast<const bvec*>(xp + (int64_t)tt * 512);
				}
				__builtin_amdgcn_sched_barrier(0);
				const float4 a = ra[s];
				bs01 += v2f{a.x, a.y}; bs23 += v2f{a.z, a.w};   // (every wave, packed: cheaper than a select; only the wk = 0 waves store it)
				mfma8(a, rb[s]);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}

	// ---- leftover rows [16 * chunks_per_foot, V) of the foot (last split only): un-pipelined; a row past the end contributes zeros
	if (tail > 0) {
		const float* tz = zfoot + ((int64_t)g.chunks_per_foot * 16) * 256 + wn * 128 + 4 * li;
		const float* tx = xfoot + ((int64_t)g.chunks_per_foot * 16) * 256 + wk * (32 * NB) + NB * li;
		const int steps = (tail + 1) >> 1;
		for (int t = 0; t < steps; ++t) {
			const int row = 2 * t + fh;
			const bool ok = row < tail;
			const int rr = min(row, tail - 1);
			float4 a = *reinterpret_cast<const float4*>(tz + (int64_t)rr * 256);
			const bvec b = *reinterpret_cast<const bvec*>(tx + (int64_t)rr * 256);
			if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
			bs01 += v2f{a.x, a.y}; bs23 += v2f{a.z, a.w};
			mfma8(a, b);
		}
	}

	// ---- epilogue: accumulator (ja, jb) element (i, j) is output n = 128 wn + 4 i + ja, k = 64 wk + 2 j + jb;
	// i = (r & 3) + 8 (r >> 2) + 4 fh, j = lane & 31: a half-wave stores 256 contiguous bytes of one output row.
	{
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(pw)), 0, 65536 * 4, 0x00020000);
		const int voff = ((wn * 128 + 16 * fh) * 256 + wk * (32 * NB) + NB * li) * 4;
#pragma unroll
		for (int ja = 0; ja < 4; ++ja)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int nrow = 4 * ((r & 3) + 8 * (r >> 2)) + ja;
				// (value copies: __builtin_bit_cast on a vector-element lvalue reads element 0)
				if constexpr (NB == 2) {
					typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
					u32x2 v;
					const float f0 = acc[ja][0][r], f1 = acc[ja][1][r];
					v.x = __float_as_uint(f0); v.y = __float_as_uint(f1);
					__builtin_amdgcn_raw_buffer_store_b64(v, rsrc, voff, nrow * 1024, 0);
				} else {
					typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
					u32x4 v;
					const float f0 = acc[ja][0][r], f1 = acc[ja][1][r], f2 = acc[ja][2 % NB][r], f3 = acc[ja][3 % NB][r];
					v.x = __float_as_uint(f0); v.y = __float_as_uint(f1); v.z = __float_as_uint(f2); v.w = __float_as_uint(f3);
					store_b128(v, rsrc, voff, nrow * 1024);
				}
			}
	}
	if (do_bias) {
		// the two row parities of a k-pair sit in the two halves of the wave
		float4 bsum = make_float4(bs01.x, bs01.y, bs23.x, bs23.y);
		bsum.x += __shfl_xor(bsum.x, 32, 64); bsum.y += __shfl_xor(bsum.y, 32, 64);
		bsum.z += __shfl_xor(bsum.z, 32, 64); bsum.w += __shfl_xor(bsum.w, 32, 64);
		if (fh == 0) *reinterpret_cast<float4*>(pb + wn * 128 + 4 * li) = bsum;
	}
}

__global__ __launch_bounds__(512) void dw4_kernel(const Dw2Args g) { dw4_body<2>(g, blockIdx.x); }
#ifdef FIND_DIAG
// (diagnosis, "dw_lds_free" = 2: dw2's register shape without its LDS ring -- four waves, one per SIMD, 128 x 128 tiles, 328 registers with 256
// accumulators in AGPRs.  It shows the fault of dw2 without any LDS in the kernel: what matters is a wave above 256 registers, mlp_kernels.h)
__global__ __launch_bounds__(256, 1) void dw4_wide_kernel(const Dw2Args g) { dw4_body<4>(g, blockIdx.x); }
#endif

// Several weight gradients of the same geometry in ONE launch (blockIdx.y = job), as dw2_group_kernel.
__global__ __launch_bounds__(512) void dw4_group_kernel(const Dw2Group grp) {
	const int j = blockIdx.y;
	Dw2Args g;   // (fields copied one by one: a reference into the kernel-argument array makes the compiler copy the array to scratch)
	g.dz = grp.job[j].dz; g.dz_foot_stride = grp.job[j].dz_foot_stride; g.x = grp.job[j].x; g.x_foot_stride = grp.job[j].x_foot_stride;
	g.chunks_per_foot = grp.job[j].chunks_per_foot; g.tail_rows = grp.job[j].tail_rows; g.spf = grp.job[j].spf; g.cps = grp.job[j].cps;
	g.pw = grp.job[j].pw; g.pb = grp.job[j].pb; g.dbg = nullptr;
	dw4_body<2>(g, blockIdx.x);
}

}  // namespace mlp
}  // namespace find
