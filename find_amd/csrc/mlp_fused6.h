// fused6_kernel: fused_chain_kernel's layer chains (mlp_fused.h: one 32- or 64-row tile per workgroup through a list of steps, the
// activation tile in LDS between layers) in bf16x3 arithmetic (mlp_gemm6.h: every fp32 operand split exactly into three bf16 pieces, the
// six products of relative size >= 2^-18 on the bf16 matrix pipe, fp32 accumulation -- fp32-level error at 16/6 of the fp32 pipe's rate).
//
// What changes against the fp32 chain:
//   * THE WEIGHTS ARRIVE SPLIT AND IN FRAGMENT ORDER.  split_w_kernel turns every GEMM step's weight matrix into its three bf16 planes once
//     per call (a few MB, a few us), laid out so that the whole chain's weights for one wave -- wave w owns output columns 32 w .. 32 w + 31
//     of every layer -- are ONE contiguous stream: [wave][16-k step t of the chain][plane][lane] x 16 B.  A wave loads a step's three operands
//     as three contiguous KB straight into MFMA operand registers (no LDS, no split arithmetic in the chain), four steps ahead through a
//     ring of named registers; the stream runs on across layer boundaries, so the next layer's first weights are in flight during the
//     epilogue;
//   * THE ACTIVATION TILE IS THREE bf16 PLANES in LDS (rows padded to 528 B: the 32-row x 2-half fragment reads are conflict-free,
//     tools/lds_b128_probe.hip), written by whoever produces it -- the Fourier-feature generator, the loader of a step's global source,
//     the previous layer's epilogue -- which splits what it writes: every activation is split ONCE per layer;
//   * the weights are the MFMA's ROW operand (v_mfma_f32_32x32x16_bf16: D[n][row] += W[n][k] X[row][k]), so a lane ends up with four
//     consecutive output COLUMNS of its row per accumulator quad: the epilogue is 16-byte bias / mask loads and stores and 8-byte LDS
//     plane writes (gemm7's transposed block);
//   * barriers that only hand the LDS tile over are LDS-only (lds_barrier: __syncthreads would drain the weight stream's prefetch twice
//     per layer); a step that reloads rows this workgroup stored itself (the colour head re-reading the trunk output) keeps the full one.
// Fourier features, 256 -> 3 output steps (on the reconstructed fp32 tile: p1 + p2 + p3 is exact), two-operand sums, masks: as in mlp_fused.h.
#pragma once
#include "mlp_dwpe.h"
#include "mlp_fused.h"
#include "mlp_gemm6.h"

namespace find {
namespace mlp {

constexpr int F6_ROW = 528;                                   // bytes per activation row of one plane: 256 bf16 + 16 B
constexpr int f6_plane(int nt) { return 32 * nt * F6_ROW; }   // 16 896 B per 32 rows
constexpr int fused6_lds(int nt) { return 3 * f6_plane(nt) + 3 * 256 * 4; }   // + Fourier matrix: 53 760 / 104 448 B
constexpr int64_t F6_STEP_BYTES = 3 * 64 * 16;                // one wave's operands of one 16-k step: three planes x 1 KB

// ---- weights -> fragment-ordered bf16 planes (once per chain launch)
struct SplitWJob { const float* w; int ldw; int nsteps; int cum; int mode; int pe; int in_dim; };   // mode: FusedStep::wmode
struct SplitWArgs {
	SplitWJob job[FUSED_MAX_STEPS];
	int njobs;
	int total;      // 16-k steps of the whole chain
	u32x4* dst;     // [8 waves][total][3 planes][64 lanes]
};

__global__ __launch_bounds__(256) void split_w_kernel(const SplitWArgs g) {
	const SplitWJob j = g.job[blockIdx.y];
	const int idx = blockIdx.x * 256 + threadIdx.x;
	const int lane = idx & 63, s = (idx >> 6) % j.nsteps, nb = (idx >> 6) / j.nsteps;
	if (nb >= 8) return;
	// MFMA row operand of step s: lane (n = lane & 31, half = lane >> 5) holds W[32 nb + n][16 s + 8 half .. + 8]
	const int n = nb * 32 + (lane & 31), k0 = s * 16 + (lane >> 5) * 8;
	float4 lo, hi;
	if (j.mode == 0) {
		const float4* src = reinterpret_cast<const float4*>(j.w + (int64_t)n * j.ldw + k0);
		lo = src[0]; hi = src[1];
	} else {
		// the weights as the model holds them: transposed (a dX step; the 32 lanes of a half read 32 consecutive floats of a row) or through
		// the Fourier layer's padded column order -- what repack_kernel used to materialise first, one launch per call on the step's critical path
		float v[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (j.mode == 1) v[i] = j.w[(int64_t)(k0 + i) * j.ldw + n];
			else { const int ko = pe_col_to_orig(k0 + i, j.pe, j.in_dim); v[i] = ko >= 0 ? j.w[(int64_t)n * j.ldw + ko] : 0.f; }
		}
		lo = make_float4(v[0], v[1], v[2], v[3]); hi = make_float4(v[4], v[5], v[6], v[7]);
	}
	bf16x8 p1, p2, p3;
	split3(lo, hi, p1, p2, p3);
	u32x4* d = g.dst + ((int64_t)nb * g.total + j.cum + s) * 192 + lane;
	d[0] = __builtin_bit_cast(u32x4, p1);
	d[64] = __builtin_bit_cast(u32x4, p2);
	d[128] = __builtin_bit_cast(u32x4, p3);
}

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int NT>
__global__ __launch_bounds__(512, 2) void fused6_kernel(const FusedArgs g) {
	constexpr int ROWS = 32 * NT, PLANE = f6_plane(NT);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	char* const XP = smem;   // plane p at XP + p * PLANE
	float* const Bl = reinterpret_cast<float*>(smem + 3 * PLANE);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & 31, lh = lane >> 5;
	const int V = g.V;
	const int total = g.total_steps;
	const u32x4* const wstream = reinterpret_cast<const u32x4*>(g.w6) + (int64_t)wave * total * 192 + lane;

	if (g.pe > 0) {
		for (int i = tid; i < 3 * g.pe; i += 512) Bl[i] = g.Bm[i];
	}

	// one (row pair) of values -> the three planes: 4 bytes each at byte offset `off` of a plane
#define F6_PUT4(off_, q0_, q1_)                                                                                   \
	do {                                                                                                          \
		*reinterpret_cast<uint2*>(XP + (off_)) = make_uint2((q0_).p1, (q1_).p1);                                  \
		*reinterpret_cast<uint2*>(XP + PLANE + (off_)) = make_uint2((q0_).p2, (q1_).p2);                          \
		*reinterpret_cast<uint2*>(XP + 2 * PLANE + (off_)) = make_uint2((q0_).p3, (q1_).p3);                      \
	} while (0)

	{   // one tile per workgroup (launch_chain starts ntiles of them)
		const int tile = blockIdx.x;
		const int foot = tile / g.tiles_per_foot;
		const int v0 = (tile - foot * g.tiles_per_foot) * ROWS;
		const int valid = min(ROWS, V - v0);
		const int64_t row0 = (int64_t)foot * V + v0;   // first global row of the tile

		// ---- the weight stream: step t of the chain in ring slot t & 3 (named registers), requested four steps ahead
		int t_pf = 0;
		u32x4 r0a, r0b, r0c, r1a, r1b, r1c, r2a, r2b, r2c, r3a, r3b, r3c;
#define F6_WLOAD(slot)                                                                          \
		do {                                                                                    \
			const u32x4* _p = wstream + (int64_t)min(t_pf, total - 1) * 192;                    \
			slot##a = _p[0]; slot##b = _p[64]; slot##c = _p[128];                               \
			++t_pf;                                                                             \
		} while (0)
		// (g.ablate -- profiling only, results are wrong under either bit: 4 no epilogue, 8 no Fourier features; tools/fused_micro.py.  Bits inside
		// the k-step loop -- no weight loads, no MFMAs -- were taken out again: their tests alone cost the step 2.5 %)
		F6_WLOAD(r0); F6_WLOAD(r1); F6_WLOAD(r2); F6_WLOAD(r3);
		__builtin_amdgcn_sched_barrier(0);

		// ONE loop over the chain's 16-k steps, four per turn; a step of the chain is set up when the previous one has ended and closed
		// (epilogue) when its last k-step is through.  (Written as nested loops -- steps of the chain outside, k-steps inside -- the
		// compiler kept two copies of the weight ring and of the accumulators and moved them back and forth at every step boundary:
		// 236 registers and spills at NT = 2.)
		f32x16 acc[NT];
		struct { const float *w, *bias, *src, *aux; float *dst, *dst2; int nchunk, bias_foot_stride, kind, src_kind, relu, mask, keep, accum, to_lds, head; } s;
		int si = 0, c = 0, nsteps = 0;
		bool setup = true;

		while (true) {
			if (setup) {
				bool done = false;
				while (true) {
					if (si >= g.n_steps) { done = true; break; }
					s.w = g.step[si].w; s.bias = g.step[si].bias; s.src = g.step[si].src; s.aux = g.step[si].aux; s.dst = g.step[si].dst; s.dst2 = g.step[si].dst2;
					s.nchunk = g.step[si].nchunk; s.bias_foot_stride = g.step[si].bias_foot_stride; s.kind = g.step[si].kind;
					s.src_kind = g.step[si].src_kind; s.relu = g.step[si].relu; s.mask = g.step[si].mask; s.keep = g.step[si].keep; s.accum = g.step[si].accum;
					s.to_lds = g.step[si].to_lds; s.head = g.step[si].head;
					if (s.kind != FS_OUT) break;
					// final 256 -> 3 layer + tanh scaling on the tile (p1 + p2 + p3 = the fp32 activation, exactly): 16 lanes per row, 16 columns each
					lds_barrier();
					const int seg = tid & 15;
#pragma unroll 1
					for (int rt = 0; rt < NT; ++rt) {   // (rolled: unrolled, this small step set the kernel's register count)
						const int row = rt * 32 + (tid >> 4);
						float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll 1
						for (int c8 = 0; c8 < 2; ++c8) {
							const int off = row * F6_ROW + (seg * 16 + c8 * 8) * 2;
							const u32x4 a1 = *reinterpret_cast<const u32x4*>(XP + off);
							const u32x4 a2 = *reinterpret_cast<const u32x4*>(XP + PLANE + off);
							const u32x4 a3 = *reinterpret_cast<const u32x4*>(XP + 2 * PLANE + off);
							float x[8];
#pragma unroll
							for (int e = 0; e < 4; ++e) {
								x[2 * e] = (bf16_lo(a1[e]) + bf16_lo(a2[e])) + bf16_lo(a3[e]);
								x[2 * e + 1] = (bf16_hi(a1[e]) + bf16_hi(a2[e])) + bf16_hi(a3[e]);
							}
#pragma unroll
							for (int e4 = 0; e4 < 2; ++e4) {
								const float4 a = *reinterpret_cast<const float4*>(s.w + 0 * W + seg * 16 + c8 * 8 + e4 * 4);
								const float4 b = *reinterpret_cast<const float4*>(s.w + 1 * W + seg * 16 + c8 * 8 + e4 * 4);
								const float4 cw = *reinterpret_cast<const float4*>(s.w + 2 * W + seg * 16 + c8 * 8 + e4 * 4);
								const float* xx = x + e4 * 4;
								p0 += xx[0] * a.x + xx[1] * a.y + xx[2] * a.z + xx[3] * a.w;
								p1 += xx[0] * b.x + xx[1] * b.y + xx[2] * b.z + xx[3] * b.w;
								p2 += xx[0] * cw.x + xx[1] * cw.y + xx[2] * cw.z + xx[3] * cw.w;
							}
						}
						p0 = row16_sum(p0); p1 = row16_sum(p1); p2 = row16_sum(p2);   // (DPP: bit for bit the xor-1, 2, 4, 8 butterfly through ds_bpermute it replaces)
						if (seg < 3 && row < valid) {
							const float zz = (seg == 0 ? p0 : (seg == 1 ? p1 : p2)) + s.bias[seg];
							const float t = tanhf(zz);
							const int64_t o = (row0 + row) * 3 + seg;
							if (s.dst2) s.dst2[o] = zz;
							s.dst[o] = s.head ? ((s.aux ? s.aux[seg] : 0.f) + 0.5f * (1.0f + t)) : 0.1f * t;
						}
					}
					++si;
				}
				if (done) break;
				// ---- a GEMM step begins
				if (s.src_kind == FS_SRC_GLOBAL) {
					__syncthreads();   // everyone is done with the previous tile, and rows this workgroup stored itself have landed
#pragma unroll
					for (int q = 0; q < 4 * NT; ++q) {
						const int idx = tid + 512 * q;
						const int row = idx >> 6, part = idx & 63;
						const float4 v = *reinterpret_cast<const float4*>(s.src + (row0 + min(row, valid - 1)) * W + part * 4);
						const Split2 q0 = split_pair(f32x2{v.x, v.y}), q1 = split_pair(f32x2{v.z, v.w});
						F6_PUT4(row * F6_ROW + part * 8, q0, q1);
					}
				}
				if (!s.accum) {
#pragma unroll
					for (int rt = 0; rt < NT; ++rt)
#pragma unroll
						for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
				}
				lds_barrier();   // the tile is in place
				nsteps = 2 * s.nchunk;   // 16-k steps: a multiple of four (chunk counts are even)
				c = 0;
				setup = false;
			}

			if (s.src_kind == FS_SRC_PE && (c & 15) == 0 && !FIND_ABL(g.ablate, 8)) {
				// regenerate the tile with the Fourier features of k-tile c / 16; a thread fills 16 columns of one row
				if (c > 0) lds_barrier();   // the previous k-tile has been consumed by every wave
				// The padded order (mlp_kernels.h: pe_value) alternates 32-column chunks: sin of 32 features, cos of the SAME 32, ..., then
				// [x y z 0 ...], zeros.  A thread takes 8 features of one (sin, cos) chunk pair: one argument and one branch-free evaluation
				// (sincospi_poly: the device library's sinpif / cospif polynomials) give both values.
				const int seg = tid & 15, m = seg >> 2, jf = (seg & 3) * 8;
#pragma unroll
				for (int rt = 0; rt < NT; ++rt) {
					const int row = rt * 32 + (tid >> 4);
					const float* pp = g.pos + (int64_t)foot * g.pos_foot_stride + (int64_t)(v0 + min(row, valid - 1)) * 3;
					const float px = pp[0], py = pp[1], pz = pp[2];
					const int cs = (c >> 4) * 8 + 2 * m, nsc = g.pe >> 4;   // the pair's sin chunk (the cos chunk follows it); nsc sin / cos chunks in all
					const int offs = row * F6_ROW + (2 * m * 32 + jf) * 2, offc = offs + 64;
					if (cs < nsc) {
						const float* b0 = Bl + (cs >> 1) * 32 + jf;
#pragma unroll 1
						for (int j = 0; j < 8; j += 4) {   // (four at a time: register pressure)
							float sv[4], cv[4];
#pragma unroll
							for (int e = 0; e < 4; ++e) {
								const float t = 2.0f * fmaf(pz, b0[2 * g.pe + j + e], fmaf(py, b0[g.pe + j + e], px * b0[j + e]));
								sincospi_poly(t, sv[e], cv[e]);
							}
							const Split2 s0 = split_pair(f32x2{sv[0], sv[1]}), s1 = split_pair(f32x2{sv[2], sv[3]});
							F6_PUT4(offs + j * 2, s0, s1);
							const Split2 c0 = split_pair(f32x2{cv[0], cv[1]}), c1 = split_pair(f32x2{cv[2], cv[3]});
							F6_PUT4(offc + j * 2, c0, c1);
						}
					} else {
						const bool xyz = cs == nsc && jf == 0;
						const Split2 q0 = split_pair(f32x2{xyz ? px : 0.f, xyz ? py : 0.f}), q1 = split_pair(f32x2{xyz ? pz : 0.f, 0.f});
						const Split2 z = split_pair(f32x2{0.f, 0.f});
						F6_PUT4(offs, q0, q1); F6_PUT4(offs + 8, z, z);
						F6_PUT4(offc, z, z); F6_PUT4(offc + 8, z, z);
					}
				}
				lds_barrier();
			}

			// ---- four 16-k steps: the six products per 32-row block, smallest terms first
			{
				const char* const xb = XP + li * F6_ROW + lh * 16;
#define F6_FRAG(buf, s_)                                                                                                \
				do {                                                                                                    \
					const char* _x = xb + ((s.src_kind == FS_SRC_PE) ? ((s_) & 15) : (s_)) * 32;                        \
					_Pragma("unroll") for (int rt = 0; rt < NT; ++rt) {                                                 \
						x1[buf][rt] = *reinterpret_cast<const bf16x8*>(_x + rt * 32 * F6_ROW);                          \
						x2[buf][rt] = *reinterpret_cast<const bf16x8*>(_x + PLANE + rt * 32 * F6_ROW);                  \
						x3[buf][rt] = *reinterpret_cast<const bf16x8*>(_x + 2 * PLANE + rt * 32 * F6_ROW);              \
					}                                                                                                   \
				} while (0)
#define F6_MMA(buf, slot)                                                                                               \
				do {                                                                                                    \
					const bf16x8 w1 = __builtin_bit_cast(bf16x8, slot##a), w2 = __builtin_bit_cast(bf16x8, slot##b),    \
								 w3 = __builtin_bit_cast(bf16x8, slot##c);                                              \
					_Pragma("unroll") for (int rt = 0; rt < NT; ++rt) {                                                 \
						acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x1[buf][rt], acc[rt], 0, 0, 0);           \
						acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2[buf][rt], acc[rt], 0, 0, 0);           \
						acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x3[buf][rt], acc[rt], 0, 0, 0);           \
						acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1[buf][rt], acc[rt], 0, 0, 0);           \
						acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2[buf][rt], acc[rt], 0, 0, 0);           \
						acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1[buf][rt], acc[rt], 0, 0, 0);           \
					}                                                                                                   \
				} while (0)
				bf16x8 x1[2][NT], x2[2][NT], x3[2][NT];
				F6_FRAG(0, c);
				F6_FRAG(1, c + 1);
				__builtin_amdgcn_sched_barrier(0);
				F6_MMA(0, r0);
				F6_WLOAD(r0);
				F6_FRAG(0, c + 2);
				__builtin_amdgcn_sched_barrier(0);
				F6_MMA(1, r1);
				F6_WLOAD(r1);
				F6_FRAG(1, c + 3);
				__builtin_amdgcn_sched_barrier(0);
				F6_MMA(0, r2);
				F6_WLOAD(r2);
				__builtin_amdgcn_sched_barrier(0);
				F6_MMA(1, r3);
				F6_WLOAD(r3);
				__builtin_amdgcn_sched_barrier(0);
#undef F6_FRAG
#undef F6_MMA
			}
			c += 4;
			if (c < nsteps) continue;

			// ---- the step's last k-step is through
			if (!s.keep && !FIND_ABL(g.ablate, 4)) {
				// epilogue: lane (row li of block rt, half lh) holds columns wave * 32 + 8 q + 4 lh .. + 3 in acc[rt][4 q .. 4 q + 3]
				lds_barrier();   // every wave has multiplied its last step: nobody reads the tile any more
				const int n0 = wave * 32 + 4 * lh;
				// buffer loads / stores: the descriptor's size = the tile's valid bytes, so rows past the end of a foot are dropped by the bounds check
				const int voff = (li * W + n0) * 4;
				const float* bp = s.bias + (int64_t)foot * s.bias_foot_stride + n0;   // (requested at the step's start instead: no gain, 16 registers)
				const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr((s.mask ? s.aux : s.w) + row0 * W)), 0, s.mask ? valid * W * 4 : 0, 0x00020000);
				const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr((s.dst ? s.dst : s.w) + row0 * W)), 0, s.dst ? valid * W * 4 : 0, 0x00020000);
				// one 32-row block at a time
#pragma unroll
				for (int rt = 0; rt < NT; ++rt) {
					u32x4 mv[4];
					if (s.mask) {
#pragma unroll
						for (int q = 0; q < 4; ++q) mv[q] = __builtin_amdgcn_raw_buffer_load_b128(msrc, voff + q * 32, rt * 32 * W * 4, 0);
					}
					if (s.relu) {
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * q);
							acc[rt][4 * q + 0] = fmaxf(acc[rt][4 * q + 0] + bv.x, 0.f);
							acc[rt][4 * q + 1] = fmaxf(acc[rt][4 * q + 1] + bv.y, 0.f);
							acc[rt][4 * q + 2] = fmaxf(acc[rt][4 * q + 2] + bv.z, 0.f);
							acc[rt][4 * q + 3] = fmaxf(acc[rt][4 * q + 3] + bv.w, 0.f);
						}
					}
					if (s.mask) {
#pragma unroll
						for (int q = 0; q < 4; ++q)
#pragma unroll
							for (int e = 0; e < 4; ++e) acc[rt][4 * q + e] = (__uint_as_float(mv[q][e]) > 0.f) ? acc[rt][4 * q + e] : 0.f;   // (a row past the end reads 0: never stored)
					}
					if (s.dst) {
#pragma unroll
						for (int q = 0; q < 4; ++q)
							store_b128(u32x4{__float_as_uint(acc[rt][4 * q]), __float_as_uint(acc[rt][4 * q + 1]), __float_as_uint(acc[rt][4 * q + 2]), __float_as_uint(acc[rt][4 * q + 3])},
									   drs, voff + q * 32, rt * 32 * W * 4);
					}
					if (s.to_lds) {
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							const Split2 q0 = split_pair(f32x2{acc[rt][4 * q], acc[rt][4 * q + 1]}), q1 = split_pair(f32x2{acc[rt][4 * q + 2], acc[rt][4 * q + 3]});
							F6_PUT4((rt * 32 + li) * F6_ROW + (n0 + 8 * q) * 2, q0, q1);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
				// (the next step starts with a barrier before anyone reads the tile)
			}
			++si;
			setup = true;
		}
#undef F6_WLOAD
	}
#undef F6_PUT4
}

}  // namespace mlp
}  // namespace find
