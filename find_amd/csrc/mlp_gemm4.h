// gemm4_kernel: third-generation Linear kernel for K = 256 (every hidden layer, forward and dX), built around what
// bounded gemm3 (profiles/: 71+ cycles per MFMA, one wave per SIMD, W re-streamed for every tile):
//   * W STAYS IN LDS.  A workgroup owns one 128-column half of the output and keeps that half of W -- 128 n x 256 k fp32
//     = 128 KB -- resident for its whole life (one LDS-DMA pass in the prologue).  No operand is re-streamed per tile
//     and the main loop has NO barrier and NO DMA;
//   * A NEVER TOUCHES LDS.  The MFMA A operand of v_mfma_f32_32x32x2_f32 is one value per lane (row l&31, k-half l>>5),
//     so a lane reads its own row straight from global memory: 4 x 16 B = k 16 consecutive floats per 32-k chunk, all
//     32 loads of a 32-row unit off ONE 64-bit base register with immediate offsets, prefetched 3 chunks ahead
//     (across unit boundaries).  Two lanes x 4 loads consume each 128-B line exactly once;
//   * TWO WAVES PER SIMD.  512 threads, 64 accumulator AGPRs + < 192 VGPRs per lane: while one wave of a SIMD waits on
//     LDS / memory or runs its epilogue, the other keeps the matrix pipe issuing.  Waves are independent (unit = 32 rows
//     x 128 columns x K 256 = 512 MFMAs, dealt round-robin inside the workgroup's row range);
//   * the two column halves of the same rows run on the same XCD (blockIdx b and b+8), so the second read of A hits L2.
// LDS layout of the W half: row R = 4*(n & 31) + (n >> 5) (so the four 32-column blocks of a lane are 1 KB apart: immediate
// offsets), 1 KB per row, 16-B slot index XORed with (n & 15): any 16 consecutive lanes of a ds_read_b128 hit 16 distinct
// slots of a 256-B bank row.  The swizzle is applied on the DMA source address; the LDS side stays lane-linear.
//
// Measured (profiles/r01_gemm4_notes.md): 315-300 K shader cycles per launch at the C2 shape against 221 K for the MFMAs alone
// (SQ_VALU_MFMA_BUSY_CYCLES / cycles = 0.74), s_waitcnt stalls ~4%.  Variants tried and dropped, all within +-3% of this one:
// hand-counted vmcnt for the A loads (asm-issued) so the next unit never waits for the previous unit's stores; 16-byte stores
// with the operand roles swapped and the bias folded into the accumulator init; one wave per SIMD with 64x128 wave tiles (main
// loop at 96% of the MFMA issue bound in isolation, but the epilogue is then dead time) and the same with two accumulator sets
// draining the previous unit under the MFMAs (the slices are not absorbed: every non-MFMA instruction of the wave costs
// matrix-pipe issue time); column quarters (NI = 2) on the large launches at two and at three waves per SIMD (123 / 127 us
// against 121); delaying the second wave of every SIMD by 8-64 K cycles so that the two waves' epilogues cannot coincide (+1 us).
// Also dropped: cutting the units of an under-filled last round (C2: 26.9 units per row range = 3.36 rounds of 8 waves) into two
// 64-column halves so that all 8 waves work for half a unit time -- 121 us against 120: a wave whose SIMD partner has run out of
// units already gets the whole matrix pipe, so the round is not the loss it looks like on paper.
// What is left is instruction issue: ~0.6 non-MFMA instructions per MFMA.
#pragma once
#include "mlp_gemm3.h"

namespace find {
namespace mlp {

constexpr int GEMM4_LDS = 128 * 1024;
constexpr int GEMM4_PD = 3;  // A prefetch distance in 32-k chunks (ring of 4)

// NI = 32-column blocks per wave unit: 4 (column halves, the large launches) or 2 (column quarters, 64 KB of W per workgroup:
// launches of a few hundred units -- the shared trunk's V rows -- spread over four times as many SIMDs as gemm3's 64-row tiles)
template <int EPI, int NI, int NW = 8>
__global__ __launch_bounds__(NW * 64) void gemm4_kernel(const Gemm2Args g) {
	constexpr int NCG = 8 / NI;  // column groups
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const unsigned lds_base = (unsigned)(uintptr_t)smem;

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
	const int li = lane & 31, fh = lane >> 5;
	const int b = blockIdx.x;
	const int npairs = gridDim.x / NCG;
	const int pair = (b / (8 * NCG)) * 8 + (b & 7);   // the column groups of the same rows are 8 blocks apart: same XCD
	const int col0 = ((b >> 3) % NCG) * (NI * 32);
	const int V = g.V, lda = g.lda, ldy = g.ldy, upf = g.tiles_per_foot;

	// ---- prologue: this column group of W -> LDS (LDS row R = NI * (n & 31) + (n >> 5); wave w: rows 4 NI w .. 4 NI (w+1) - 1)
	{
		const float* wb = uniform_ptr(g.w0);
		if (wave < 8)
#pragma unroll
		for (int j = 0; j < NI; ++j) {
			const int r0 = wave * 4 * NI + 4 * j;
			unsigned o[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int l2 = (r0 + i) / NI, ni = (r0 + i) % NI;  // n & 31, n >> 5
				o[i] = (unsigned)(((col0 + ni * 32 + l2) * g.ldw + ((lane ^ (l2 & 15)) * 4)) * 4);
			}
			dma4(wb, __builtin_amdgcn_readfirstlane(lds_base + r0 * 1024), o[0], o[1], o[2], o[3]);
		}
		FIND_WAIT_VMCNT(0);
		__syncthreads();
	}

	const int u0 = (int)((int64_t)pair * g.ntiles / npairs);
	const int u1 = (int)((int64_t)(pair + 1) * g.ntiles / npairs);
	int u = u0 + wave;
	if (u >= u1) return;

	// B fragment byte offsets: LDS row NI*li (+ni), slot ((c&1)*8 + fh*4 + q) ^ (li & 15); (c>>1)*256 and ni*1024 are immediates
	unsigned boff[2][4];
#pragma unroll
	for (int c1 = 0; c1 < 2; ++c1)
#pragma unroll
		for (int q = 0; q < 4; ++q) boff[c1][q] = (unsigned)((li * NI) * 1024 + (((c1 * 8 + fh * 4 + q) ^ (li & 15)) * 16));

	auto unit_rows = [&](int uu, int& foot, int& v0) -> const float4* {
		foot = uu / upf;
		v0 = (uu - foot * upf) * 32;
		const int row = min(v0 + li, V - 1);  // rows past the end of a foot re-read its last row (never stored)
		return reinterpret_cast<const float4*>(g.a0 + (int64_t)foot * g.a_foot_stride + (int64_t)row * lda + fh * 16);
	};

	int foot, v0;
	const float4* cur = unit_rows(u, foot, v0);
	float4 areg[4][4];
#pragma unroll
	for (int c = 0; c < GEMM4_PD; ++c)
#pragma unroll
		for (int q = 0; q < 4; ++q) areg[c][q] = cur[c * 8 + q];

	float4 bf[2][NI];
	auto load_b = [&](int c, int q, float4 (&f)[NI]) {
#pragma unroll
		for (int ni = 0; ni < NI; ++ni) f[ni] = *reinterpret_cast<const float4*>(smem + boff[c & 1][q] + (c >> 1) * 256 + ni * 1024);
	};
	load_b(0, 0, bf[0]);

	for (; u < u1; u += NW) {
		int nfoot = foot, nv0 = v0;
		const float4* nxt = (u + NW < u1) ? unit_rows(u + NW, nfoot, nv0) : cur + (8 - GEMM4_PD) * 8;  // (no next unit: re-read lines just fetched, not chunks 0..2 from HBM)

		float bv[NI];
		if constexpr (EPI == EPI_BIAS_RELU) {
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) bv[ni] = g.bias[(int64_t)foot * g.bias_foot_stride + col0 + ni * 32 + li];
		}

		f32x16 acc[NI];
#pragma unroll
		for (int ni = 0; ni < NI; ++ni)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

		// (epilogue descriptors up front: the first mask block is requested under the last chunk's MFMAs)
		const int valid_rows = min(32, V - v0);
		float* ytile = g.y + (int64_t)foot * g.y_foot_stride + (int64_t)v0 * ldy;
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(ytile)), 0, valid_rows * ldy * 4, 0x00020000);
		const int voff = ((4 * fh) * ldy + col0 + li) * 4;
		__amdgpu_buffer_rsrc_t msrc = rsrc;
		if constexpr (EPI == EPI_MASK) {
			const float* mtile = g.mask + (int64_t)foot * g.mask_foot_stride + (int64_t)v0 * ldy;
			msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(mtile)), 0, valid_rows * ldy * 4, 0x00020000);
		}
		// EPI_MASK: the mask values of block ni + 1 are requested before block ni is stored -- the compiler keeps a buffer load behind
		// the buffer stores in front of it (two descriptors: it cannot rule out aliasing), so block by block the epilogue paid the
		// load latency four times per unit -- and those of block 0 before the unit's last chunk is multiplied.
		float mv[2][16];
		auto load_mask = [&](int ni, float (&m)[16]) {
#pragma unroll
			for (int r = 0; r < 16; ++r)
				m[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(msrc, voff + ((r & 3) * ldy + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0));
		};

		// (the wave in its MFMA loop outranks the SIMD's other wave while that one runs its epilogue: 122.7 -> 121.0 us per launch at the C2
		// shape in isolation, no difference inside the step; ablate bit 16 switches it off)
		if (!(g.ablate & 16)) __builtin_amdgcn_s_setprio(2);
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			if constexpr (EPI == EPI_MASK) {
				if (c == 7) load_mask(0, mv[0]);
			}
			// A prefetch: chunk c+PD of this unit, or chunk c+PD-8 of the wave's next unit
			{
				const int pc = c + GEMM4_PD;
				const float4* src = (pc < 8) ? cur + pc * 8 : nxt + (pc - 8) * 8;
#pragma unroll
				for (int q = 0; q < 4; ++q) areg[pc & 3][q] = src[q];
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int s = c * 4 + q;
				// B fragments of the next k-group (wraps to (0,0): the next unit multiplies the same W)
				load_b(((s + 1) >> 2) & 7, (s + 1) & 3, bf[(s + 1) & 1]);
				const float4 a = areg[c & 3][q];
				const float4(&f)[NI] = bf[s & 1];
#pragma unroll
				for (int kk = 0; kk < 4; ++kk) {
					const float av = kk == 0 ? a.x : (kk == 1 ? a.y : (kk == 2 ? a.z : a.w));
#pragma unroll
					for (int ni = 0; ni < NI; ++ni) {
						const float bvv = kk == 0 ? f[ni].x : (kk == 1 ? f[ni].y : (kk == 2 ? f[ni].z : f[ni].w));
						acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv, acc[ni], 0, 0, 0);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}

		if (!(g.ablate & 16)) __builtin_amdgcn_s_setprio(0);
		// ---- epilogue: buffer stores; the SRD's size is the number of valid bytes of the unit, so rows past the end of a
		// foot are dropped by the bounds check.  Element (r, lane) of block ni = row (r&3) + 8(r>>2) + 4fh, column 32ni + li.
		{
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) {
				if constexpr (EPI == EPI_MASK) {
					if (ni + 1 < NI) load_mask(ni + 1, mv[(ni + 1) & 1]);
				}
				// (bias: one packed add for two accumulator registers -- a non-MFMA instruction costs the SIMD's other wave a matrix-pipe issue slot;
				//  there is no packed maximum for the ReLU)
				typedef float v2f __attribute__((ext_vector_type(2)));
				float vals[16];
#pragma unroll
				for (int r = 0; r < 16; r += 2) {
					if constexpr (EPI == EPI_BIAS_RELU) {
						const v2f t = v2f{acc[ni][r], acc[ni][r + 1]} + v2f{bv[ni], bv[ni]};
						vals[r] = fmaxf(t.x, 0.f); vals[r + 1] = fmaxf(t.y, 0.f);
					} else {
						vals[r] = acc[ni][r]; vals[r + 1] = acc[ni][r + 1];
					}
				}
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					float val = vals[r];
					if constexpr (EPI == EPI_MASK) val = (mv[ni & 1][r] > 0.f) ? val : 0.f;
					__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rsrc, voff + ((r & 3) * ldy + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0);
				}
			}
		}
		cur = nxt; foot = nfoot; v0 = nv0;
	}
}

}  // namespace mlp
}  // namespace find
