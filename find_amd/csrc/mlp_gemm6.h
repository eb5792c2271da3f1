// gemm6_kernel: the K = 256 Linear layer in fp32-FAITHFUL arithmetic on the bf16 matrix pipe ("bf16x3").
//
// gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  An fp32 number is, exactly, the sum of three bf16
// numbers: a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (round to nearest; both differences are exact
// in fp32, the last one has at most 8 significant bits, and bf16 has fp32's exponent range, so nothing overflows, underflows or is
// lost: |a2| <= 2^-9 |a|, |a3| <= 2^-18 |a|).  A product then is nine bf16 x bf16 products, each exact in the matrix pipe's fp32
// accumulator; the six of relative size >= 2^-18
//        a1 b1   +   a1 b2 + a2 b1   +   a1 b3 + a2 b2 + a3 b1
// leave out a2 b3 + a3 b2 + a3 b3 <= 2^-26 |a b| -- a quarter of the rounding error of ONE fp32 multiplication (2^-24) -- so the
// layer's results carry the error of fp32 accumulation and nothing else: they are as close to the float64 product as the exact-fp32
// MFMA kernel's (tests/test_gpu_mlp_bf16x3.py measures both), at 6/16 of its matrix-pipe time.  This is not a reduced-precision
// mode (that is gemm5's fp16: operands rounded to 11 bits); tensors stay fp32 in HBM, operands are split on their way into the pipe.
//
// Structure: gemm5's (no barrier and no DMA after the prologue, A never touches LDS) with the weight operand in three planes:
//   * a workgroup (8 waves, two per SIMD) owns a 64-column quarter of the output and keeps that quarter of W in LDS as three bf16
//     planes (64 n x 256 k x 3, rows padded to 528 B: conflict-free ds_read_b128), split once in the prologue;
//   * a wave unit is 32 rows x 64 columns: lane (row l&31, half l>>5) reads 16 consecutive floats of its row per 32-k chunk straight
//     from global memory (three chunks ahead, across units), splits them into the three 8 x bf16 operands of the chunk's two MFMA
//     k-steps (v_cvt_pk_bf16_f32 + packed subtracts: ~36 VALU instructions per k-step against 12 MFMAs = 96 issue slots), and runs
//     the six products per column block, smallest terms first;
//   * the four column quarters of the same rows sit 8 blocks apart (same XCD): A comes from HBM once and from L2 three times.
// Epilogues as gemm4 / gemm5.
#pragma once
#include "mlp_gemm5.h"

namespace find {
namespace mlp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int G6_ROW = 528;                       // bytes per W row of one plane in LDS: 256 bf16 + 16 B of padding
constexpr int G6_NI = 2;                          // 32-column blocks per workgroup
constexpr int G6_PLANE = G6_NI * 32 * G6_ROW;     // 33 792 B
constexpr int GEMM6_LDS = 3 * G6_PLANE;           // 101 376 B
constexpr int GEMM6_NW = 8;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Two floats -> their three bf16 pieces, packed (element 0 in the low half): a = p1 + p2 + p3 exactly.  Written pair by pair so that
// the compiler emits the 9-instruction form (v_cvt_pk_bf16_f32; v_lshlrev_b32 + v_and_b32 to widen the pair back; v_pk_add_f32 with
// negated second operand; twice; v_cvt_pk_bf16_f32): over 8-wide vectors it converts and subtracts element by element (60 for 36).
struct Split2 { unsigned p1, p2, p3; };
__device__ __forceinline__ Split2 split_pair(const f32x2 a) {
	Split2 o;
	o.p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
	const f32x2 r1 = a - f32x2{__uint_as_float(o.p1 << 16), __uint_as_float(o.p1 & 0xffff0000u)};
	o.p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
	const f32x2 r2 = r1 - f32x2{__uint_as_float(o.p2 << 16), __uint_as_float(o.p2 & 0xffff0000u)};
	o.p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
	return o;
}

// 8 floats -> the three 8 x bf16 MFMA operands
__device__ __forceinline__ void split3(const float4& lo, const float4& hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
	const Split2 q0 = split_pair(f32x2{lo.x, lo.y}), q1 = split_pair(f32x2{lo.z, lo.w}), q2 = split_pair(f32x2{hi.x, hi.y}), q3 = split_pair(f32x2{hi.z, hi.w});
	p1 = __builtin_bit_cast(bf16x8, u32x4{q0.p1, q1.p1, q2.p1, q3.p1});
	p2 = __builtin_bit_cast(bf16x8, u32x4{q0.p2, q1.p2, q2.p2, q3.p2});
	p3 = __builtin_bit_cast(bf16x8, u32x4{q0.p3, q1.p3, q2.p3, q3.p3});
}

#ifdef FIND_DIAG   // (the kernel itself: diagnostics build only, "gemm7" = 0; the product keeps this header's split helpers)
// ABL: profiling only (tools/ablate_x3.py): 1 = no split arithmetic, 2 = no LDS fragment reads after a unit's first, 4 = no A loads after the
// prologue -- compile-time, so that the measured loop keeps its basic blocks
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(GEMM6_NW * 64) void gemm6_kernel(const Gemm2Args g) {
	constexpr int NI = G6_NI;
	constexpr int NCG = 8 / NI;  // column groups
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & 31, fh = lane >> 5;
	const int b = blockIdx.x;
	const int npairs = gridDim.x / NCG;
	const int pair = (b / (8 * NCG)) * 8 + (b & 7);   // the column groups of the same rows are 8 blocks apart: same XCD
	const int col0 = ((b >> 3) % NCG) * (NI * 32);
	const int V = g.V, lda = g.lda, ldy = g.ldy, upf = g.tiles_per_foot;

	const int u0 = (int)((int64_t)pair * g.ntiles / npairs);
	const int u1 = (int)((int64_t)(pair + 1) * g.ntiles / npairs);
	if (u0 >= u1) return;

	auto unit_rows = [&](int uu, int& foot, int& v0) -> const float4* {
		foot = uu / upf;
		v0 = (uu - foot * upf) * 32;
		const int row = min(v0 + li, V - 1);  // rows past the end of a foot re-read its last row (never stored)
		return reinterpret_cast<const float4*>(g.a0 + (int64_t)foot * g.a_foot_stride + (int64_t)row * lda + fh * 16);
	};

	// the first A chunks are on their way while W is split
	int u = u0 + wave;
	const bool active = u < u1;
	int foot = 0, v0 = 0;
	const float4* cur = unit_rows(active ? u : u0, foot, v0);
	float4 areg[4][4];
#pragma unroll
	for (int c = 0; c < GEMM4_PD; ++c)
#pragma unroll
		for (int q = 0; q < 4; ++q) areg[c][q] = cur[c * 8 + q];

	// ---- prologue: this column group of W (64, ldw) fp32 -> three bf16 planes in LDS; item = (row n, 8-k group): 2048 items over 512 threads
	{
		const float* wb = g.w0 + (int64_t)col0 * g.ldw;
#pragma unroll
		for (int it = 0; it < (NI * 32 * 32) / (GEMM6_NW * 64); ++it) {
			const int item = it * (GEMM6_NW * 64) + tid;
			const int n = item >> 5, kg = item & 31;
			const float4* src = reinterpret_cast<const float4*>(wb + (int64_t)n * g.ldw + kg * 8);
			bf16x8 p1, p2, p3;
			split3(src[0], src[1], p1, p2, p3);
			char* dst = smem + n * G6_ROW + kg * 16;
			*reinterpret_cast<bf16x8*>(dst) = p1;
			*reinterpret_cast<bf16x8*>(dst + G6_PLANE) = p2;
			*reinterpret_cast<bf16x8*>(dst + 2 * G6_PLANE) = p3;
		}
		__syncthreads();
	}
	if (!active) return;

	// B fragment of (plane p, chunk c, step m, column block ni): W row 32 ni + li, 8 bf16 at k = 32c + 16fh + 8m
	const char* const bbase = smem + li * G6_ROW + fh * 32;
	auto load_b = [&](int p, int c, int m, int ni) -> bf16x8 {
		return *reinterpret_cast<const bf16x8*>(bbase + p * G6_PLANE + ni * (32 * G6_ROW) + c * 64 + m * 16);
	};

	for (; u < u1; u += GEMM6_NW) {
		int nfoot = foot, nv0 = v0;
		const float4* nxt = (u + GEMM6_NW < u1) ? unit_rows(u + GEMM6_NW, nfoot, nv0) : cur + (8 - GEMM4_PD) * 8;

		f32x16 acc[NI];
#pragma unroll
		for (int ni = 0; ni < NI; ++ni)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

		// Software pipeline over the unit's 16 k-steps (step s = chunk s >> 1, half s & 1): while the matrix pipe runs the twelve products
		// of step s, the wave splits the A values of step s + 1 and its B fragments arrive from LDS.  The instruction ORDER is prescribed
		// (sched_group_barrier: LDS reads first, then one MFMA followed by three VALU instructions, twelve times): left alone, the
		// scheduler emits the 36 VALU instructions of a split in one run and the MFMAs in another, and neither hides the other.
		bf16x8 a1[2], a2[2], a3[2], b1[2][NI], b2[2][NI], b3[2][NI];
		split3(areg[0][0], areg[0][1], a1[0], a2[0], a3[0]);
#pragma unroll
		for (int ni = 0; ni < NI; ++ni) { b1[0][ni] = load_b(0, 0, 0, ni); b2[0][ni] = load_b(1, 0, 0, ni); b3[0][ni] = load_b(2, 0, 0, ni); }
#pragma unroll
		for (int s = 0; s < 16; ++s) {
			const int c = s >> 1, m = s & 1, cu = s & 1, nx = cu ^ 1;
			if (m == 0 && !(ABL & 4)) {  // A prefetch: chunk c+PD of this unit, or chunk c+PD-8 of the wave's next unit
				const int pc = c + GEMM4_PD;
				const float4* src = (pc < 8) ? cur + pc * 8 : nxt + (pc - 8) * 8;
#pragma unroll
				for (int q = 0; q < 4; ++q) areg[pc & 3][q] = src[q];
			}
			if (s + 1 < 16) {
				const int c1 = (s + 1) >> 1, m1 = (s + 1) & 1;
				if constexpr (!(ABL & 2)) {
#pragma unroll
					for (int ni = 0; ni < NI; ++ni) { b1[nx][ni] = load_b(0, c1, m1, ni); b2[nx][ni] = load_b(1, c1, m1, ni); b3[nx][ni] = load_b(2, c1, m1, ni); }
				} else {
#pragma unroll
					for (int ni = 0; ni < NI; ++ni) { b1[nx][ni] = b1[cu][ni]; b2[nx][ni] = b2[cu][ni]; b3[nx][ni] = b3[cu][ni]; }
				}
				if constexpr (!(ABL & 1)) split3(areg[c1 & 3][2 * m1], areg[c1 & 3][2 * m1 + 1], a1[nx], a2[nx], a3[nx]);
				else { a1[nx] = __builtin_bit_cast(bf16x8, areg[c1 & 3][2 * m1]); a2[nx] = __builtin_bit_cast(bf16x8, areg[c1 & 3][2 * m1 + 1]); a3[nx] = a1[cu]; }
			}
			// smallest terms first: what the accumulator rounds away is then the least it can be
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[cu], b1[cu][ni], acc[ni], 0, 0, 0);
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[cu], b3[cu][ni], acc[ni], 0, 0, 0);
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[cu], b2[cu][ni], acc[ni], 0, 0, 0);
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[cu], b1[cu][ni], acc[ni], 0, 0, 0);
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[cu], b2[cu][ni], acc[ni], 0, 0, 0);
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[cu], b1[cu][ni], acc[ni], 0, 0, 0);
			if (m == 0) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // the four global loads of the chunk prefetch
			__builtin_amdgcn_sched_group_barrier(0x100, 3 * NI, 0);            // next step's B fragments
#pragma unroll
			for (int i = 0; i < 6 * NI; ++i) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
				__builtin_amdgcn_sched_group_barrier(0x002, 3, 0);             // three VALU instructions of the next step's split
			}
			__builtin_amdgcn_sched_barrier(0);
		}

		// ---- epilogue (as gemm4): element (r, lane) of block ni = row (r&3) + 8(r>>2) + 4fh, column col0 + 32ni + li
		{
			const int valid_rows = min(32, V - v0);
			float* ytile = g.y + (int64_t)foot * g.y_foot_stride + (int64_t)v0 * ldy;
			const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(ytile)), 0, valid_rows * ldy * 4, 0x00020000);
			const int voff = ((4 * fh) * ldy + col0 + li) * 4;
			__amdgpu_buffer_rsrc_t msrc = rsrc;
			if constexpr (EPI == EPI_MASK) {
				const float* mtile = g.mask + (int64_t)foot * g.mask_foot_stride + (int64_t)v0 * ldy;
				msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(mtile)), 0, valid_rows * ldy * 4, 0x00020000);
			}
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) {
				float bv = 0.f;
				if constexpr (EPI == EPI_BIAS_RELU) bv = g.bias[(int64_t)foot * g.bias_foot_stride + col0 + ni * 32 + li];
				float mv[16];
				if constexpr (EPI == EPI_MASK) {
#pragma unroll
					for (int r = 0; r < 16; ++r)
						mv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(msrc, voff + ((r & 3) * ldy + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0));
				}
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					float val = acc[ni][r];
					if constexpr (EPI == EPI_BIAS_RELU) val = fmaxf(val + bv, 0.f);
					if constexpr (EPI == EPI_MASK) val = (mv[r] > 0.f) ? val : 0.f;
					__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rsrc, voff + ((r & 3) * ldy + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0);
				}
			}
		}
		cur = nxt; foot = nfoot; v0 = nv0;
	}
}
#endif

}  // namespace mlp
}  // namespace find
