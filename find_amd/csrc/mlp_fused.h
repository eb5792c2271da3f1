// fused_chain_kernel: a whole chain of 256-wide Linear layers on ONE 32-row tile per workgroup, in one launch.
//
// Why: when a call has few rows -- the reference's own batch size 1 (6890 template vertices = 216 32-row tiles), the 1000 texture
// samples, the shared trunk of a 16-foot batch -- every layer of the MLP was its own launch over fewer tiles than the chip has CUs,
// each with its prologue, its tail and a ~1.5 us boundary to the next: thirteen dependent launches of ~20 us for 7 us of matrix-pipe
// work each.  Here a workgroup takes a tile through the whole chain: the activation tile lives in LDS between layers (32 x 256 fp32),
// the weights stream from L2 through a double-buffered LDS chunk ring (all workgroups read the same 256 KB per layer), each layer's
// output is written to HBM once (the backward needs it) and handed to the next layer through LDS.  The chain is a list of steps
// in the kernel arguments, so the same kernel runs the forward (Fourier features -> trunk -> heads -> 3-wide outputs) and the dX
// chain of the backward (masked transposed products, two-operand sums).
//
// MFMA: v_mfma_f32_32x32x2_f32, exact fp32.  A wave owns 64 output columns (two 32 x 32 accumulators) of the tile's 32 rows.
// Operand reads: one ds_read_b128 per operand per EIGHT k -- lanes 0-31 take k = 8g .. 8g+3, lanes 32-63 k = 8g+4 .. 8g+7, and MFMA s
// pairs component s of both halves (the contraction order inside a chunk is permuted; the sum is the same set of products).
// LDS: X 32 x 260 floats (rows padded by 16 B: conflict-free ds_read_b128 across 32 rows), two W chunks of 256 x 36 floats.
#pragma once
#include "mlp_kernels.h"

namespace find {
namespace mlp {

enum { FS_SRC_LDS = 0, FS_SRC_PE = 1, FS_SRC_GLOBAL = 2 };
enum { FS_GEMM = 0, FS_OUT = 1 };

struct FusedStep {
	const float* w;       // GEMM: (256, ldw) rows n, k contiguous.  OUT: (3, 256)
	const float* bias;    // GEMM + relu: (.., 256) at bias + foot * bias_foot_stride.  OUT: (3)
	const float* src;     // SRC_GLOBAL: rows (foot, v), ld 256, loaded into the X tile first
	const float* aux;     // mask: activation rows (ld 256) whose sign gates the result.  OUT: avg_col or null
	float* dst;           // GEMM: output rows (ld 256) or null.  OUT: activated output (rows, 3)
	float* dst2;          // OUT: pre-activation z (rows, 3) or null
	int ldw;
	int nchunk;           // K = 32 * nchunk (PE: 8 chunks per regenerated 256-wide k-tile)
	int bias_foot_stride;
	unsigned char kind;   // FS_GEMM / FS_OUT
	unsigned char src_kind;
	unsigned char relu;   // epilogue: + bias, ReLU
	unsigned char mask;   // epilogue: zero where aux <= 0
	unsigned char keep;   // 1: leave the accumulators to the next step (a two-operand sum), no epilogue here
	unsigned char accum;  // 1: start from the previous step's accumulators
	unsigned char to_lds; // 1: the result becomes the X tile of the next step
	unsigned char head;   // OUT: 0 = displacement (0.1 tanh), 1 = colour (0.5 (1 + tanh))
};

constexpr int FUSED_MAX_STEPS = 26;
struct FusedArgs {
	FusedStep step[FUSED_MAX_STEPS];
	int n_steps;
	const float* pos;         // PE: (pos_batch, V, 3)
	int64_t pos_foot_stride;
	const float* Bm;          // (3, pe)
	int pe;
	int V;                    // rows per foot
	int tiles_per_foot;
	int ntiles;
};

constexpr int FX_LD = 260;                       // X row stride (floats)
constexpr int FW_LD = 36;                        // W chunk row stride (floats)
constexpr int FUSED_X_BYTES = 32 * FX_LD * 4;    // 33 280
constexpr int FUSED_W_BYTES = 256 * FW_LD * 4;   // 36 864 per stage
constexpr int FUSED_LDS = FUSED_X_BYTES + 2 * FUSED_W_BYTES + 3 * 256 * 4;   // + Fourier matrix

__global__ __launch_bounds__(256, 1) void fused_chain_kernel(const FusedArgs g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* const X = reinterpret_cast<float*>(smem);
	float* const Wc0 = reinterpret_cast<float*>(smem + FUSED_X_BYTES);
	float* const Wc1 = reinterpret_cast<float*>(smem + FUSED_X_BYTES + FUSED_W_BYTES);
	float* const Bl = reinterpret_cast<float*>(smem + FUSED_X_BYTES + 2 * FUSED_W_BYTES);

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & 31, lh = lane >> 5;
	const int V = g.V;

	if (g.pe > 0) {
		for (int i = tid; i < 3 * g.pe; i += 256) Bl[i] = g.Bm[i];
	}

	// W chunk staging: thread -> 8 x 16 B of the 256 x 32 chunk (row = idx / 8, part = idx % 8: 8 threads read one 128-B row piece)
	// (kept in eight named registers and moved by unconditional code: behind conditionals the compiler parks them in scratch memory)
	float4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7;
	const int w_row = tid >> 3, w_part = tid & 7;   // + 32 rows per q
#define FUSED_W_LOAD(wp, ldw_, c_)                                                                              \
	do {                                                                                                        \
		const float* _b = (wp) + (int64_t)w_row * (ldw_) + (c_) * 32 + w_part * 4;                              \
		const int64_t _s = (int64_t)32 * (ldw_);                                                                \
		wr0 = *reinterpret_cast<const float4*>(_b); wr1 = *reinterpret_cast<const float4*>(_b + _s);            \
		wr2 = *reinterpret_cast<const float4*>(_b + 2 * _s); wr3 = *reinterpret_cast<const float4*>(_b + 3 * _s); \
		wr4 = *reinterpret_cast<const float4*>(_b + 4 * _s); wr5 = *reinterpret_cast<const float4*>(_b + 5 * _s); \
		wr6 = *reinterpret_cast<const float4*>(_b + 6 * _s); wr7 = *reinterpret_cast<const float4*>(_b + 7 * _s); \
	} while (0)
#define FUSED_W_STORE(Wc_)                                                                                      \
	do {                                                                                                        \
		float* _d = (Wc_) + w_row * FW_LD + w_part * 4;                                                         \
		*reinterpret_cast<float4*>(_d) = wr0; *reinterpret_cast<float4*>(_d + 32 * FW_LD) = wr1;                \
		*reinterpret_cast<float4*>(_d + 64 * FW_LD) = wr2; *reinterpret_cast<float4*>(_d + 96 * FW_LD) = wr3;   \
		*reinterpret_cast<float4*>(_d + 128 * FW_LD) = wr4; *reinterpret_cast<float4*>(_d + 160 * FW_LD) = wr5; \
		*reinterpret_cast<float4*>(_d + 192 * FW_LD) = wr6; *reinterpret_cast<float4*>(_d + 224 * FW_LD) = wr7; \
	} while (0)

	for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
		const int foot = tile / g.tiles_per_foot;
		const int v0 = (tile - foot * g.tiles_per_foot) * 32;
		const int valid = min(32, V - v0);
		const int64_t row0 = (int64_t)foot * V + v0;   // first global row of the tile

		f32x16 acc[2];
		int stage = 0;        // W ring stage holding the chunk about to be multiplied
		bool primed = false;  // chunk 0 of the current step already sits in the ring (prefetched under the previous step)

		for (int si = 0; si < g.n_steps; ++si) {
			// the step's fields as scalars (indexing the kernel-argument array through a reference makes the compiler copy it to scratch)
			struct { const float *w, *bias, *src, *aux; float *dst, *dst2; int ldw, nchunk, bias_foot_stride, kind, src_kind, relu, mask, keep, accum, to_lds, head; } s;
			s.w = g.step[si].w; s.bias = g.step[si].bias; s.src = g.step[si].src; s.aux = g.step[si].aux; s.dst = g.step[si].dst; s.dst2 = g.step[si].dst2;
			s.ldw = g.step[si].ldw; s.nchunk = g.step[si].nchunk; s.bias_foot_stride = g.step[si].bias_foot_stride; s.kind = g.step[si].kind;
			s.src_kind = g.step[si].src_kind; s.relu = g.step[si].relu; s.mask = g.step[si].mask; s.keep = g.step[si].keep; s.accum = g.step[si].accum;
			s.to_lds = g.step[si].to_lds; s.head = g.step[si].head;
			const bool next_gemm = si + 1 < g.n_steps && g.step[si + 1].kind == FS_GEMM;
			const float* const next_w = next_gemm ? g.step[si + 1].w : nullptr;
			const int next_ldw = next_gemm ? g.step[si + 1].ldw : 0;
			if (s.kind == FS_OUT) {
				// final 256 -> 3 layer + tanh scaling on the X tile: 8 lanes per row, 32 columns each
				__syncthreads();
				const int row = tid >> 3, seg = tid & 7;
				float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
				for (int c4 = 0; c4 < 8; ++c4) {
					const float4 x = *reinterpret_cast<const float4*>(X + row * FX_LD + seg * 32 + c4 * 4);
					const float4 a = *reinterpret_cast<const float4*>(s.w + 0 * W + seg * 32 + c4 * 4);
					const float4 b = *reinterpret_cast<const float4*>(s.w + 1 * W + seg * 32 + c4 * 4);
					const float4 c = *reinterpret_cast<const float4*>(s.w + 2 * W + seg * 32 + c4 * 4);
					p0 += x.x * a.x + x.y * a.y + x.z * a.z + x.w * a.w;
					p1 += x.x * b.x + x.y * b.y + x.z * b.z + x.w * b.w;
					p2 += x.x * c.x + x.y * c.y + x.z * c.z + x.w * c.w;
				}
#pragma unroll
				for (int d = 1; d < 8; d <<= 1) { p0 += __shfl_xor(p0, d, 64); p1 += __shfl_xor(p1, d, 64); p2 += __shfl_xor(p2, d, 64); }
				if (seg < 3 && row < valid) {
					const float zz = (seg == 0 ? p0 : (seg == 1 ? p1 : p2)) + s.bias[seg];
					const float t = tanhf(zz);
					const int64_t o = (row0 + row) * 3 + seg;
					if (s.dst2) s.dst2[o] = zz;
					s.dst[o] = s.head ? ((s.aux ? s.aux[seg] : 0.f) + 0.5f * (1.0f + t)) : 0.1f * t;
				}
				primed = false;
				continue;
			}
			// ---- GEMM step
			if (!primed) { FUSED_W_LOAD(s.w, s.ldw, 0); }
			if (s.src_kind == FS_SRC_GLOBAL) {
				__syncthreads();   // everyone is done with the previous X
#pragma unroll
				for (int q = 0; q < 8; ++q) {
					const int idx = tid + 256 * q;
					const int row = idx >> 6, part = idx & 63;
					const float4 v = *reinterpret_cast<const float4*>(s.src + (row0 + min(row, valid - 1)) * W + part * 4);
					*reinterpret_cast<float4*>(X + row * FX_LD + part * 4) = v;
				}
			}
			if (!primed) { FUSED_W_STORE(stage ? Wc1 : Wc0); }
			if (!s.accum) {
#pragma unroll
				for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
			}
			__syncthreads();

			const int nchunk = s.nchunk;
			for (int c = 0; c < nchunk; ++c) {
				if (s.src_kind == FS_SRC_PE && (c & 7) == 0) {
					// regenerate the X tile with the Fourier features of k-tile c / 8 (padded order of pe_value); 32 values per thread
					if (c > 0) __syncthreads();   // the previous k-tile has been consumed
					const int row = tid >> 3, seg = tid & 7;
					const float* pp = g.pos + (int64_t)foot * g.pos_foot_stride + (int64_t)(v0 + min(row, valid - 1)) * 3;
					const float px = pp[0], py = pp[1], pz = pp[2];
					// this thread's 32 columns are chunk cc of the padded order (mlp_kernels.h: pe_value): sin of 32 features, cos of the same
					// 32, ..., then [x y z 0 ...], then zeros
					const int cc = (c >> 3) * 8 + seg, nsc = g.pe >> 4;
					float* xr = X + row * FX_LD + seg * 32;
					if (cc < nsc) {
						const float* b0 = Bl + (cc >> 1) * 32;
						const bool is_cos = cc & 1;
#pragma unroll 4
						for (int j = 0; j < 32; ++j) {
							const float t = 2.0f * fmaf(pz, b0[2 * g.pe + j], fmaf(py, b0[g.pe + j], px * b0[j]));
							xr[j] = is_cos ? cospif(t) : sinpif(t);
						}
					} else {
#pragma unroll
						for (int j = 0; j < 32; j += 4) *reinterpret_cast<float4*>(xr + j) = make_float4(0.f, 0.f, 0.f, 0.f);
						if (cc == nsc) { xr[0] = px; xr[1] = py; xr[2] = pz; }
					}
					__syncthreads();
				}
				// prefetch the next chunk of the stream: of this step, or chunk 0 of the next GEMM step
				// (when nothing follows, the same chunk is fetched again into the idle stage: unconditional code keeps the registers registers)
				const bool more = c + 1 < nchunk;
				const bool ns = more || next_gemm;
				{
					const float* pw = more ? s.w : (next_gemm ? next_w : s.w);
					const int pl = more ? s.ldw : (next_gemm ? next_ldw : s.ldw);
					const int pc = more ? c + 1 : (next_gemm ? 0 : c);
					FUSED_W_LOAD(pw, pl, pc);
				}

				__builtin_amdgcn_sched_barrier(0);   // the prefetch is issued BEFORE the MFMAs (left alone, the compiler sinks it next to its store)
				const float* Wc = stage ? Wc1 : Wc0;
				const int xk = (s.src_kind == FS_SRC_PE ? (c & 7) : c) * 32;
				const float* xa = X + li * FX_LD + xk + lh * 4;
				const float* wb = Wc + (wave * 64 + li) * FW_LD + lh * 4;
				// fragments of k-group q+1 are requested before the MFMAs of group q (two register sets)
				float4 fa[2], fb0[2], fb1[2];
				fa[0] = *reinterpret_cast<const float4*>(xa);
				fb0[0] = *reinterpret_cast<const float4*>(wb);
				fb1[0] = *reinterpret_cast<const float4*>(wb + 32 * FW_LD);
#pragma unroll
				for (int gq = 0; gq < 4; ++gq) {
					const int cur = gq & 1, nxt = cur ^ 1;
					if (gq < 3) {
						fa[nxt] = *reinterpret_cast<const float4*>(xa + (gq + 1) * 8);
						fb0[nxt] = *reinterpret_cast<const float4*>(wb + (gq + 1) * 8);
						fb1[nxt] = *reinterpret_cast<const float4*>(wb + 32 * FW_LD + (gq + 1) * 8);
					}
					__builtin_amdgcn_sched_barrier(0);
					const float4 a = fa[cur], b0 = fb0[cur], b1 = fb1[cur];
					acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
					acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
					acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
					acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
					acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
					acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
					acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
					acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
					__builtin_amdgcn_sched_barrier(0);
				}
				FUSED_W_STORE(stage ? Wc0 : Wc1);
				(void)ns;
				stage ^= 1;
				__syncthreads();
			}
			primed = next_gemm;
			if (s.keep) continue;

			// ---- epilogue: bias + ReLU / mask, store to HBM, hand the tile to the next step through LDS
			// (every wave has passed the barrier that ended the last chunk: nobody reads X any more)
#pragma unroll
			for (int nt = 0; nt < 2; ++nt) {
				const int col = wave * 64 + nt * 32 + li;
				const float bv = s.relu ? s.bias[(int64_t)foot * s.bias_foot_stride + col] : 0.f;
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
					float val = acc[nt][r];
					if (s.relu) val = fmaxf(val + bv, 0.f);
					if (s.mask) val = (s.aux[(row0 + min(row, valid - 1)) * W + col] > 0.f) ? val : 0.f;
					if (s.dst && row < valid) s.dst[(row0 + row) * W + col] = val;
					if (s.to_lds) X[row * FX_LD + col] = val;
				}
			}
			// (the next step starts with a barrier before anyone reads X)
		}
		__syncthreads();   // the next tile overwrites X and the ring
	}
}

}  // namespace mlp
}  // namespace find
