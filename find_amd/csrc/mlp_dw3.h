// dw3_kernel: weight gradient  dW[n,k] = sum_rows dZ[row,n] * X[row,k]  on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, fp32
// accumulation) -- the weight-gradient half of the opt-in fp16 mode (mlp_gemm5.h has the forward / dX half).  The contraction runs
// over ROWS, so both MFMA operands need 8 consecutive rows of one column per lane: the kernel transposes on the way into LDS.
//   * one 4-wave workgroup owns the full 256 x 256 output (256 accumulator registers per lane) over a run of 64-row chunks of ONE
//     foot, as dw2 does; partial tiles go to the same slabs and reduce_w_kernel, per-foot bias sums likewise (summed in fp32 from
//     the un-rounded dZ values while they pass through registers);
//   * staging: thread (c = tid & 63, g = tid >> 6) loads the float4 at columns 4c..4c+3 of rows 8(g + 4u) .. +7, u = 0, 1, of both
//     operands (a row is read by 64 lanes as one contiguous KB), rounds to fp16 and writes, per column, the 8 rows as ONE 16-byte
//     LDS word: LDS layout [column][64 rows] fp16 = 128 B per column, 16-byte slot s (rows 8s..8s+7) stored at slot s ^ ((col>>2)&7)
//     so the 8 lanes of a write phase hit 8 different bank groups;
//   * an MFMA k-step is 16 rows: lane (i, h) of the A fragment reads slot 2t + h of column n = 32 ti + i with one ds_read_b128;
//   * the chunk after the one being multiplied is already in flight in registers (32 float4 per thread): at 16x the fp32 MFMA rate
//     the kernel is bound by the HBM stream (2 KB per row), not by the matrix pipe, LDS or the transposition.
// Rows past the end of a foot are zero-filled at load time (no tail path).
#pragma once
#include "mlp_gemm5.h"

namespace find {
namespace mlp {

struct Dw3Args {
	const float* dz;         // rows (foot, v), ld 256
	int64_t dz_foot_stride;
	const float* x;          // rows (foot, v) or shared (x_foot_stride 0), ld 256
	int64_t x_foot_stride;
	int V;                   // rows per foot
	int chunks_per_foot;     // ceil(V / 64)
	int spf;                 // splits per foot (>= 1)
	int cps;                 // 64-row chunks per split
	float* pw;               // [n_feet*spf][256][256]
	float* pb;               // [n_feet*spf][256] or nullptr
	const float* xbias;      // dw3_h16v_kernel (bcast_fold): x is the shared fp32 product P (x_foot_stride 0) and the operand is
	int64_t xbias_stride;    // fp16(relu(P[v] + xbias[foot])) -- what bias_relu_bcast would have stored, formed on the way in
};

constexpr int DW3_OPER = 256 * 128;            // one operand of one chunk in LDS: 256 columns x 64 rows fp16
constexpr int DW3_BUF = 2 * DW3_OPER;          // dZ then X
constexpr int DW3_LDS = 2 * DW3_BUF + 4 * 256 * 4;  // double buffer + bias reduction scratch = 135 168 B

// H16 ("act16", mlp.hip): both operands are STORED as fp16 (strides in elements as before): 8 bytes per thread and row instead of 16.
template <bool H16, bool VX = false>
__device__ __forceinline__ void dw3_body(const Dw3Args& g) {
	static_assert(!VX || H16, "dw3_body: the virtual operand exists beside fp16-stored gradients only");
	FIND_CLAIM_WHOLE_REGISTER_FILE();   // 432 registers by itself (256 fp32 accumulators in AGPRs): see the macro
	constexpr int ES = H16 ? 2 : 4;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* red = reinterpret_cast<float*>(smem + 2 * DW3_BUF);

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wn = wave >> 1, wk = wave & 1;
	const int li = lane & 31, fh = lane >> 5;
	const int split = blockIdx.x;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, g.chunks_per_foot);
	float* const pw = g.pw + (int64_t)split * 65536;
	float* const pb = g.pb ? g.pb + (int64_t)split * 256 : nullptr;
	const char* const zfoot = reinterpret_cast<const char*>(g.dz) + (int64_t)foot * g.dz_foot_stride * ES;
	const char* const xfoot = reinterpret_cast<const char*>(g.x) + (int64_t)foot * g.x_foot_stride * (VX ? 4 : ES);
	auto ld4 = [](const char* base, int64_t elem) -> float4 {
		if constexpr (H16) {
			typedef _Float16 h4 __attribute__((ext_vector_type(4)));
			const h4 h = *reinterpret_cast<const h4*>(base + elem * 2);
			return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
		} else {
			return *reinterpret_cast<const float4*>(base + elem * 4);
		}
	};

	f32x16 acc[4][4];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < 4; ++b)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
	float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

	// staging registers: [operand][u][row j of the group]
	float4 st[2][2][8];
	const int c4 = lane * 4;  // first of this thread's four columns
	auto load_chunk = [&](int q) {
		const int r0 = q * 64;
#pragma unroll
		for (int u = 0; u < 2; ++u)
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int row = r0 + 8 * (wave + 4 * u) + j;
				const bool ok = row < g.V;
				const int rc = ok ? row : g.V - 1;
				const float4 z = ld4(zfoot, (int64_t)rc * 256 + c4);
				const float4 x = ld4(xfoot, (int64_t)rc * 256 + c4);
				st[0][u][j] = ok ? z : make_float4(0.f, 0.f, 0.f, 0.f);
				st[1][u][j] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
			}
	};
	auto comp = [](const float4& v, int e) -> float { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
	auto store_chunk = [&](char* buf) {
#pragma unroll
		for (int op = 0; op < 2; ++op)
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const int slot = (wave + 4 * u) ^ (lane & 7);   // ((col >> 2) & 7) == lane & 7 for every one of the four columns
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					f16x8 v;
#pragma unroll
					for (int j = 0; j < 8; ++j) v[j] = (_Float16)comp(st[op][u][j], e);
					*reinterpret_cast<f16x8*>(buf + op * DW3_OPER + (c4 + e) * 128 + slot * 16) = v;
				}
			}
		if (pb) {
#pragma unroll
			for (int u = 0; u < 2; ++u)
#pragma unroll
				for (int j = 0; j < 8; ++j) { bsum.x += st[0][u][j].x; bsum.y += st[0][u][j].y; bsum.z += st[0][u][j].z; bsum.w += st[0][u][j].w; }
		}
	};
	// fragment addresses: column wn*128 + 32 ti + li (dZ) / wk*128 + 32 tj + li (X); slot (2t + fh) ^ ((col >> 2) & 7)
	const int swz = (li >> 2) & 7;   // 32 ti and the 128-column wave offsets are multiples of 32: they do not change (col >> 2) & 7
	const int za = (wn * 128 + li) * 128;
	const int xa = DW3_OPER + (wk * 128 + li) * 128;
	auto multiply = [&](const char* buf) {
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			const int so = ((2 * t + fh) ^ swz) * 16;
			f16x8 a[4], b[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				a[i] = *reinterpret_cast<const f16x8*>(buf + za + i * (32 * 128) + so);
				b[i] = *reinterpret_cast<const f16x8*>(buf + xa + i * (32 * 128) + so);
			}
#pragma unroll
			for (int ti = 0; ti < 4; ++ti)
#pragma unroll
				for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
		}
	};

	if constexpr (H16) {
		// fp16-stored operands: a chunk is half the registers, so TWO chunks are kept in flight (as dw6 does) -- with one, the memory pipeline
		// idles while a chunk is transposed into LDS and the launch reads at ~3 TB/s; the halves go to LDS as they are (no conversion)
		typedef unsigned u2 __attribute__((ext_vector_type(2)));
		u2 hs[2][2][2][8];   // [set][operand][u][row j]: the thread's four columns of a row as two dwords
		// VX: the x operand arrives as fp32 products; they stay raw in flight (xs) and become relu(P + bias) -> fp16 when the chunk is stored
		float4 xs[VX ? 2 : 1][2][VX ? 8 : 1];
		float4 xb = make_float4(0.f, 0.f, 0.f, 0.f);
		if constexpr (VX) xb = *reinterpret_cast<const float4*>(g.xbias + (int64_t)foot * g.xbias_stride + c4);
		auto load_h = [&](int q, u2 (&set)[2][2][8], int si) {
			const int r0 = q * 64;
#pragma unroll
			for (int u = 0; u < 2; ++u)
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const int row = r0 + 8 * (wave + 4 * u) + j;
					const bool ok = row < g.V;
					const int rc = ok ? row : g.V - 1;
					const u2 z = *reinterpret_cast<const u2*>(zfoot + ((int64_t)rc * 256 + c4) * 2);
					set[0][u][j] = ok ? z : u2{0u, 0u};
					if constexpr (VX) {
						// (rows past the end of a foot: dZ is zero there, whatever x is)
						xs[si][u][j] = *reinterpret_cast<const float4*>(xfoot + ((int64_t)rc * 256 + c4) * 4);
					} else {
						const u2 x = *reinterpret_cast<const u2*>(xfoot + ((int64_t)rc * 256 + c4) * 2);
						set[1][u][j] = ok ? x : u2{0u, 0u};
					}
				}
		};
		auto store_h = [&](char* buf, u2 (&set)[2][2][8], int si) {
			if constexpr (VX) {
				typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
				for (int u = 0; u < 2; ++u)
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						const float4 v = xs[si][u][j];
						h2v lo, hi;
						lo[0] = (_Float16)fmaxf(v.x + xb.x, 0.f); lo[1] = (_Float16)fmaxf(v.y + xb.y, 0.f);
						hi[0] = (_Float16)fmaxf(v.z + xb.z, 0.f); hi[1] = (_Float16)fmaxf(v.w + xb.w, 0.f);
						set[1][u][j] = u2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
					}
			}
#pragma unroll
			for (int op = 0; op < 2; ++op)
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const int slot = (wave + 4 * u) ^ (lane & 7);
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						u32x4 v;   // rows 2 jj, 2 jj + 1 of column e as one dword
#pragma unroll
						for (int jj = 0; jj < 4; ++jj) {
							const unsigned a0 = (e < 2) ? set[op][u][2 * jj].x : set[op][u][2 * jj].y, a1 = (e < 2) ? set[op][u][2 * jj + 1].x : set[op][u][2 * jj + 1].y;
							v[jj] = (e & 1) ? ((a0 >> 16) | (a1 & 0xffff0000u)) : ((a0 & 0xffffu) | (a1 << 16));
						}
						*reinterpret_cast<u32x4*>(buf + op * DW3_OPER + (c4 + e) * 128 + slot * 16) = v;
					}
				}
			if (pb) {
				typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
				for (int u = 0; u < 2; ++u)
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						const unsigned wlo = set[0][u][j].x, whi = set[0][u][j].y;   // (value copies: __builtin_bit_cast on a vector-element lvalue reads element 0)
						const h2 lo = __builtin_bit_cast(h2, wlo), hi = __builtin_bit_cast(h2, whi);
						bsum.x += (float)lo[0]; bsum.y += (float)lo[1]; bsum.z += (float)hi[0]; bsum.w += (float)hi[1];
					}
			}
		};
		if (q0 < q1) {
			load_h(q0, hs[0], 0);
			load_h(min(q0 + 1, q1 - 1), hs[1], 1);
			store_h(smem, hs[0], 0);
			load_h(min(q0 + 2, q1 - 1), hs[0], 0);
			__syncthreads();
			int cb = 0;
			// chunk q is in LDS buffer cb; `set` holds chunk q + 1 (stored under this chunk's MFMAs), then takes chunk q + 3.  (A repeated
			// load re-reads the run's last chunk; a repeated store goes to the buffer nobody reads any more.)
			auto body = [&](int q, u2 (&set)[2][2][8], int si) {
				multiply(smem + cb * DW3_BUF);
				if (q + 1 < q1) store_h(smem + (cb ^ 1) * DW3_BUF, set, si);
				load_h(min(q + 3, q1 - 1), set, si);
				__syncthreads();
				cb ^= 1;
			};
			for (int q = q0; q < q1; q += 2) {
				body(q, hs[1], 1);
				if (q + 1 < q1) body(q + 1, hs[0], 0);
			}
		}
	} else
	if (q0 < q1) {
		load_chunk(q0);
		store_chunk(smem);
		__syncthreads();
		int cb = 0;
		for (int q = q0; q < q1; ++q) {
			const bool more = q + 1 < q1;
			if (more) load_chunk(q + 1);            // in flight under the MFMAs
			multiply(smem + cb * DW3_BUF);
			if (more) store_chunk(smem + (cb ^ 1) * DW3_BUF);   // the other buffer: its readers finished before the last barrier
			__syncthreads();
			cb ^= 1;
		}
	}

	// ---- epilogue: tile (ti, tj) element (r, lane) is n = wn*128 + 32ti + (r&3) + 8(r>>2) + 4fh, k = wk*128 + 32tj + li
	{
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(pw)), 0, 65536 * 4, 0x00020000);
		const int voff = ((wn * 128 + 4 * fh) * 256 + wk * 128 + li) * 4;
#pragma unroll
		for (int ti = 0; ti < 4; ++ti)
#pragma unroll
			for (int tj = 0; tj < 4; ++tj)
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const float f = acc[ti][tj][r];
					__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f), rsrc, voff + ((r & 3) * 256 + tj * 32) * 4, ((ti * 32 + 8 * (r >> 2)) * 256) * 4, 0);
				}
	}
	if (pb) {
		__syncthreads();
		*reinterpret_cast<float4*>(&red[wave * 256 + lane * 4]) = bsum;
		__syncthreads();
		pb[tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
	}
}

__global__ __launch_bounds__(256, 1) void dw3_kernel(const Dw3Args g) { dw3_body<false>(g); }
__global__ __launch_bounds__(256, 1) void dw3_h16_kernel(const Dw3Args g) { dw3_body<true>(g); }
__global__ __launch_bounds__(256, 1) void dw3_h16v_kernel(const Dw3Args g) { dw3_body<true, true>(g); }

}  // namespace mlp
}  // namespace find
