// gemm5_kernel: the K = 256 Linear layer on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, fp32 accumulation) -- the opt-in
// reduced-precision mode of BASELINE.json configs[4] ("fp16 MLP with MFMA tiles").  Activations and weights stay fp32 in HBM; both
// MFMA operands are rounded to fp16 (round to nearest even) on their way into the matrix pipe, the products are exact and the sums
// fp32, outputs are written as fp32.  At 16x the fp32 MFMA rate the layer is bound by HBM (2 KB per row in + out) instead of by the
// matrix pipe, so the kernel is built around the streams, with gemm4's skeleton:
//   * a workgroup (8 waves, two per SIMD) keeps the WHOLE weight matrix in LDS as fp16 -- 256 n x 256 k, rows padded to 528 B so
//     that 16 consecutive lanes of a ds_read_b128 (one W row each) cover all 64 banks -- converted once in the prologue; a wave
//     unit is 32 rows x all 256 columns, so A is read from HBM exactly once (gemm4 reads it once per column half);
//   * A never touches LDS: lane (row l&31, half l>>5) reads 16 consecutive floats of its row per 32-k chunk (the two halves
//     share a 128-B line), three chunks ahead across unit boundaries, and packs them into the two 8 x fp16 operands of the
//     chunk's two MFMA k-steps.  The k order inside a chunk is therefore permuted (k = 32c + 16h + 8m + j for step m, element j);
//     the B fragments read LDS in the same order, and a sum does not care;
//   * no barrier and no DMA after the prologue; epilogue as gemm4 (bias + ReLU, or the ReLU mask of the backward, or plain).
// Rounding both operands to 11 significant bits gives relative errors of ~1e-3 per layer output (tests/test_gpu_mlp_f16.py states
// the tolerance); the fp32 kernels remain the default and the parity path.
#pragma once
#include "mlp_gemm4.h"

namespace find {
namespace mlp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int G5_ROW = 528;                 // bytes per W row in LDS: 256 fp16 + 16 B of padding
constexpr int GEMM5_LDS = 256 * G5_ROW;     // 135 168 B
constexpr int GEMM5_NW = 8;

__device__ __forceinline__ f16x8 pack_f16x8(const float4& lo, const float4& hi) {
	f16x8 v;
	v[0] = (_Float16)lo.x; v[1] = (_Float16)lo.y; v[2] = (_Float16)lo.z; v[3] = (_Float16)lo.w;
	v[4] = (_Float16)hi.x; v[5] = (_Float16)hi.y; v[6] = (_Float16)hi.z; v[7] = (_Float16)hi.w;
	return v;
}

// H16 ("act16", mlp.hip): A, y and the mask are STORED as fp16 (ld in elements as before): the lane's 16 k-values of a chunk are its two
// MFMA operands as they lie in memory, the epilogue writes / tests 2-byte elements -- half the HBM bytes of a layer that is bound by them.
// VIRT ("bcast_fold", mlp.hip; H16 only): the layer behind a head's broadcast first layer never reads that layer's output h1 =
// fp16(relu(P[v] + bias[foot])) from HBM -- n_feet x V x 512 B written once and read three times per step -- but forms it from the V x 256
// fp32 product P (a0 / mask, rows shared by all feet: L2) and the foot's bias row on the way in.  EPI_BIAS_RELU: the A operand
// (g.va_bias; the bias row of the wave's current foot sits in LDS behind W); EPI_MASK: the ReLU mask (g.vm_bias).  The values are the
// stored path's bit for bit (same add, same max, same round-to-nearest-even).  Units run tile-major (g.tile_major: the 8 waves of a
// workgroup take the same 32 rows of consecutive feet), so P is pulled from HBM once.
constexpr int GEMM5_LDS_VIRT = GEMM5_LDS + GEMM5_NW * 1024;
template <int EPI, bool H16 = false, bool VIRT = false>
__global__ __launch_bounds__(GEMM5_NW * 64) void gemm5_kernel(const Gemm2Args g) {
	static_assert(!VIRT || (H16 && EPI != EPI_NONE), "gemm5_kernel: the virtual operand exists for the fp16-stored heads only");
	constexpr bool VA = VIRT && EPI == EPI_BIAS_RELU, VM = VIRT && EPI == EPI_MASK;
	constexpr int ES = H16 ? 2 : 4;          // bytes per stored element of y / mask (and of A unless it is virtual)
	constexpr int AES = VA ? 4 : ES;         // bytes per element of what the A loads read
	constexpr int CH = 32 * AES;             // bytes of a 32-k chunk of one row
	constexpr int NQ = CH / 2 / 16;          // 16-byte pieces of a lane's half chunk: 4 (fp32) or 2 (fp16)
	typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & 31, fh = lane >> 5;
	const int V = g.V, lda = g.lda, ldy = g.ldy, upf = g.tiles_per_foot;

	const int u0 = (int)((int64_t)blockIdx.x * g.ntiles / gridDim.x);
	const int u1 = (int)((int64_t)(blockIdx.x + 1) * g.ntiles / gridDim.x);
	if (u0 >= u1) return;

	const int nf = g.ntiles / upf;
	auto unit_rows = [&](int uu, int& foot, int& v0) -> const char* {
		if (VIRT) {   // (tile-major units: g.tile_major says so to the reader of a launch, the template decides)
			const int t = uu / nf;
			foot = uu - t * nf;
			v0 = t * 32;
		} else {
			foot = uu / upf;
			v0 = (uu - foot * upf) * 32;
		}
		const int row = min(v0 + li, V - 1);  // rows past the end of a foot re-read its last row (never stored)
		return reinterpret_cast<const char*>(g.a0) + ((int64_t)foot * g.a_foot_stride + (int64_t)row * lda + fh * 16) * AES;
	};
	auto piece = [](const char* base, int chunk, int q) -> u32x4g { return *reinterpret_cast<const u32x4g*>(base + chunk * CH + q * 16); };

	// the first A chunks are on their way while W is converted
	int u = u0 + wave;
	const bool active = u < u1;
	int foot = 0, v0 = 0;
	const char* cur = unit_rows(active ? u : u0, foot, v0);
	u32x4g areg[4][NQ];
#pragma unroll
	for (int c = 0; c < GEMM4_PD; ++c)
#pragma unroll
		for (int q = 0; q < NQ; ++q) areg[c][q] = piece(cur, c, q);

	// ---- prologue: W (256, ldw) fp32 -> LDS fp16; item = (row n, 8-k group): 8192 items over 512 threads
	{
		const float* wb = g.w0;
#pragma unroll 4
		for (int it = 0; it < 16; ++it) {
			const int item = it * (GEMM5_NW * 64) + tid;
			const int n = item >> 5, kg = item & 31;
			const float4* src = reinterpret_cast<const float4*>(wb + (int64_t)n * g.ldw + kg * 8);
			const float4 lo = src[0], hi = src[1];
			*reinterpret_cast<f16x8*>(smem + n * G5_ROW + kg * 16) = pack_f16x8(lo, hi);
		}
		__syncthreads();
	}
	if (!active) return;

	// B fragment of (chunk c, step m, column block ni): W row 32 ni + li, 8 fp16 at k = 32c + 16fh + 8m
	const char* const bbase = smem + li * G5_ROW + fh * 32;
	auto load_b = [&](int c, int m, int ni) -> f16x8 { return *reinterpret_cast<const f16x8*>(bbase + ni * (32 * G5_ROW) + c * 64 + m * 16); };

	float* const brow = reinterpret_cast<float*>(smem + GEMM5_LDS + wave * 1024);   // VA: the bias row of the wave's current foot
	int bfoot = -1;
	for (; u < u1; u += GEMM5_NW) {
		if constexpr (VA) {
			if (foot != bfoot) {   // (wave-uniform; the wave's LDS instructions execute in order: no barrier between this write and the reads below)
				const float4 b4 = *reinterpret_cast<const float4*>(g.va_bias + (int64_t)foot * g.va_bias_stride + lane * 4);
				asm volatile("" ::: "memory");
				*reinterpret_cast<float4*>(brow + lane * 4) = b4;
				asm volatile("" ::: "memory");
				bfoot = foot;
			}
		}
		int nfoot = foot, nv0 = v0;
		// (no next unit: the run-ahead loads re-read chunks 5..7 of this unit -- lines the wave has just fetched, served by L2 -- instead
		// of pulling chunks 0..2 in from HBM a second time: 24 MB per launch at the C2 shape, profiles/r01_traffic_pmc_summary.txt)
		const char* nxt = (u + GEMM5_NW < u1) ? unit_rows(u + GEMM5_NW, nfoot, nv0) : cur + (8 - GEMM4_PD) * CH;

		f32x16 acc[8];
#pragma unroll
		for (int ni = 0; ni < 8; ++ni)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

#pragma unroll
		for (int c = 0; c < 8; ++c) {
			{  // A prefetch: chunk c+PD of this unit, or chunk c+PD-8 of the wave's next unit
				const int pc = c + GEMM4_PD;
#pragma unroll
				for (int q = 0; q < NQ; ++q) areg[pc & 3][q] = (pc < 8) ? piece(cur, pc, q) : piece(nxt, pc - 8, q);
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int m = 0; m < 2; ++m) {
				f16x8 a;
				if constexpr (VA) {
					const float4 b0 = *reinterpret_cast<const float4*>(brow + c * 32 + fh * 16 + m * 8), b1 = *reinterpret_cast<const float4*>(brow + c * 32 + fh * 16 + m * 8 + 4);
					float4 x0 = __builtin_bit_cast(float4, areg[c & 3][2 * m]), x1 = __builtin_bit_cast(float4, areg[c & 3][2 * m + 1]);
					x0.x = fmaxf(x0.x + b0.x, 0.f); x0.y = fmaxf(x0.y + b0.y, 0.f); x0.z = fmaxf(x0.z + b0.z, 0.f); x0.w = fmaxf(x0.w + b0.w, 0.f);
					x1.x = fmaxf(x1.x + b1.x, 0.f); x1.y = fmaxf(x1.y + b1.y, 0.f); x1.z = fmaxf(x1.z + b1.z, 0.f); x1.w = fmaxf(x1.w + b1.w, 0.f);
					a = pack_f16x8(x0, x1);
				} else if constexpr (H16) a = __builtin_bit_cast(f16x8, areg[c & 3][m]);
				else a = pack_f16x8(__builtin_bit_cast(float4, areg[c & 3][2 * m]), __builtin_bit_cast(float4, areg[c & 3][2 * m + 1]));
				f16x8 bf[8];
#pragma unroll
				for (int ni = 0; ni < 8; ++ni) bf[ni] = load_b(c, m, ni);
#pragma unroll
				for (int ni = 0; ni < 8; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[ni], acc[ni], 0, 0, 0);
			}
		}

		// ---- epilogue (as gemm4): element (r, lane) of block ni = row (r&3) + 8(r>>2) + 4fh, column 32ni + li
		{
			const int valid_rows = min(32, V - v0);
			char* ytile = reinterpret_cast<char*>(g.y) + ((int64_t)foot * g.y_foot_stride + (int64_t)v0 * ldy) * ES;
			const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(static_cast<const char*>(ytile)), 0, valid_rows * ldy * ES, 0x00020000);
			const int voff = ((4 * fh) * ldy + li) * ES;
			__amdgpu_buffer_rsrc_t msrc = rsrc;
			if constexpr (EPI == EPI_MASK) {
				constexpr int MES = VM ? 4 : ES;
				const char* mtile = reinterpret_cast<const char*>(g.mask) + ((int64_t)foot * g.mask_foot_stride + (int64_t)v0 * ldy) * MES;
				msrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(mtile), 0, valid_rows * ldy * MES, 0x00020000);
			}
#pragma unroll
			for (int ni = 0; ni < 8; ++ni) {
				if constexpr (VM) { if (ni & 1) __builtin_amdgcn_sched_barrier(0); }   // (the 4-byte mask loads of all eight blocks hoisted at once do not fit the register budget)
				float bv = 0.f;
				if constexpr (EPI == EPI_BIAS_RELU) bv = g.bias[(int64_t)foot * g.bias_foot_stride + ni * 32 + li];
				float mv[16];
				if constexpr (EPI == EPI_MASK) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						if constexpr (VM) mv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(msrc, (((4 * fh) + (r & 3)) * ldy + li + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0));
						else if constexpr (H16) mv[r] = (float)(short)__builtin_amdgcn_raw_buffer_load_b16(msrc, voff + ((r & 3) * ldy + ni * 32) * ES, (8 * (r >> 2) * ldy) * ES, 0);   // (the sign is all that is asked)
						else mv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(msrc, voff + ((r & 3) * ldy + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0));
					}
					if constexpr (VM) {   // h1 = fp16(relu(P + bias)) is positive where P + bias rounds to a positive fp16: above half the smallest subnormal
						const float bvm = g.vm_bias[(int64_t)foot * g.vm_bias_stride + ni * 32 + li];
#pragma unroll
						for (int r = 0; r < 16; ++r) mv[r] = (mv[r] + bvm > 0x1p-25f) ? 1.f : 0.f;
					}
				}
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					float val = acc[ni][r];
					if constexpr (EPI == EPI_BIAS_RELU) val = fmaxf(val + bv, 0.f);
					if constexpr (EPI == EPI_MASK) val = (mv[r] > 0.f) ? val : 0.f;
					if constexpr (H16) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)val), rsrc, voff + ((r & 3) * ldy + ni * 32) * ES, (8 * (r >> 2) * ldy) * ES, 0);
					else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rsrc, voff + ((r & 3) * ldy + ni * 32) * 4, (8 * (r >> 2) * ldy) * 4, 0);
				}
			}
		}
		cur = nxt; foot = nfoot; v0 = nv0;
	}
}

}  // namespace mlp
}  // namespace find
