// fused_chain_kernel: a whole chain of 256-wide Linear layers on ONE 32-row (or 64-row: NT = 2) tile per workgroup, in one launch.
//
// Why: when a call has few rows -- the reference's own batch size 1 (6890 template vertices = 216 32-row tiles), the 1000 texture
// samples, the shared trunk of a 16-foot batch -- every layer of the MLP was its own launch over fewer tiles than the chip has CUs,
// each with its prologue, its tail and a ~1.5 us boundary to the next: thirteen dependent launches of ~20 us for 7 us of matrix-pipe
// work each.  Here a workgroup takes a tile through the whole chain: the activation tile lives in LDS between layers (32 x 256 fp32),
// the weights stream from L2 through a double-buffered LDS chunk ring (all workgroups read the same 256 KB per layer), each layer's
// output is written to HBM once (the backward needs it) and handed to the next layer through LDS.  The chain is a list of steps
// in the kernel arguments, so the same kernel runs the forward (Fourier features -> trunk -> heads -> 3-wide outputs) and the dX
// chain of the backward (masked transposed products, two-operand sums).
// (Measured and dropped: an up-front L2 warm-up of the chain's weights -- every workgroup of an XCD requesting its share of the 128-byte
// lines once through LDS-DMA into a sink -- on the theory that the per-tile time, 12 us per layer whatever the tile count, was a chain of
// cold L2 misses; it changed nothing (0.642 against 0.634 ms for a batch-1 forward + backward): the tile time is the matrix pipe's 6.8 us
// per layer for a 32 x 256 x 256 tile on one CU plus the staging around it, not the weights' latency.)
//
// MFMA: v_mfma_f32_32x32x2_f32, exact fp32.  Eight waves (two per SIMD); a wave owns 32 output columns (one 32 x 32 accumulator) of
// the tile's 32 rows and stages ITS OWN 32 weight rows through a private two-stage LDS ring: no workgroup barrier inside a layer, the
// waves drift apart and one wave's staging / epilogue runs under the other's MFMAs (with four lock-stepped waves every phase of a chunk
// -- L2 loads, LDS stores, barrier, fragment reads -- added to the MFMA time: 235 us per 12-layer chain against 84 us of matrix pipe;
// tools/fused_micro.py).  Only the activation tile X is shared: two barriers per layer.
// Operand reads: one ds_read_b128 per operand per EIGHT k -- lanes 0-31 take k = 8g .. 8g+3, lanes 32-63 k = 8g+4 .. 8g+7, and MFMA s
// pairs component s of both halves (the contraction order inside a chunk is permuted; the sum is the same set of products).
// LDS: X 32 x 260 floats (rows padded by 16 B: conflict-free ds_read_b128 across 32 rows), per wave two W chunks of 32 x 36 floats.
// NT = 2 (64-row tiles, two accumulators per wave, every B fragment multiplied with two A fragments) for calls of more 32-row blocks than
// CUs: the 16 x 1000 texture samples 243 -> ~200 us per pass.  Epilogue through buffer loads / stores (one VGPR offset for the sixteen
// rows of a lane instead of sixteen 64-bit addresses): 256 -> 193 registers at NT = 1, and what lets NT = 2 fit in 256 without scratch;
// train_3d step 3.39 -> 3.30 (epilogue) -> 3.25 ms (NT = 2).
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
	int nchunk;           // K = 32 * nchunk, EVEN (PE: 8 chunks per regenerated 256-wide k-tile; the padded weight's extra chunk is zeros)
	int bias_foot_stride;
	unsigned char kind;   // FS_GEMM / FS_OUT
	unsigned char src_kind;
	unsigned char relu;   // epilogue: + bias, ReLU
	unsigned char mask;   // epilogue: zero where aux <= 0
	unsigned char keep;   // 1: leave the accumulators to the next step (a two-operand sum), no epilogue here
	unsigned char accum;  // 1: start from the previous step's accumulators
	unsigned char to_lds; // 1: the result becomes the X tile of the next step
	unsigned char head;   // OUT: 0 = displacement (0.1 tanh), 1 = colour (0.5 (1 + tanh))
	unsigned char wmode;  // GEMM, bf16x3 chains only (split_w_kernel reads the weights once per call anyway): 0 = w as described above; 1 = w is the
	                      // layer's weight as the MODEL holds it and the step multiplies by its transpose (a dX step: W_eff[n][k] = w[k * ldw + n]);
	                      // 2 = w is base.0.weight in the reference's column order, read through the padded Fourier order (pe_col_to_orig)
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
	int ablate;               // profiling only: 1 no W staging (loads + LDS stores), 2 no MFMAs, 4 no epilogue, 8 no Fourier features
	const void* w6;           // fused6_kernel (mlp_fused6.h): the chain's weights as fragment-ordered bf16 planes, [8 waves][total_steps][3][64] x 16 B
	int total_steps;          //   16-k steps of all GEMM steps of the chain
};

constexpr int FX_LD = 260;                       // X row stride (floats)
constexpr int FW_LD = 36;                        // W chunk row stride (floats)
constexpr int FUSED_NW = 8;                      // waves per workgroup
constexpr int FUSED_WS_BYTES = 32 * FW_LD * 4;   // one wave's chunk: 32 rows x 32 k, padded (4 608)
// NT = 32-row blocks per tile: 1, or 2 for calls of more 32-row blocks than the chip has CUs (the 16 x 1000 texture samples: 512 blocks
// were two rounds of workgroups; as 256 64-row tiles they are one, and every staged weight chunk feeds twice the MFMAs)
constexpr int fused_x_bytes(int nt) { return 32 * nt * FX_LD * 4; }   // 33 280 per 32 rows
constexpr int fused_lds(int nt) { return fused_x_bytes(nt) + FUSED_NW * 2 * FUSED_WS_BYTES + 3 * 256 * 4; }   // + Fourier matrix

// A wave's W staging: lane -> row = lane / 8 (+ 8 q), 16-byte part = lane % 8: 8 lanes read one 128-B row piece.  The chunk stream of a
// tile -- every 32-k chunk of every GEMM step, in order -- runs two chunks ahead of the MFMAs: chunk k+2 is requested into one of two
// register sets while chunk k is multiplied, and chunk k+1 moves from its registers into the wave's idle LDS stage.
#define FUSED_W_LOAD(set, wp, ldw_, c_)                                                                            \
	do {                                                                                                           \
		const float* _b = (wp) + (int64_t)(wave * 32 + w_row) * (ldw_) + (c_) * 32 + w_part * 4;                   \
		const int64_t _s = (int64_t)8 * (ldw_);                                                                    \
		set##0 = *reinterpret_cast<const float4*>(_b); set##1 = *reinterpret_cast<const float4*>(_b + _s);         \
		set##2 = *reinterpret_cast<const float4*>(_b + 2 * _s); set##3 = *reinterpret_cast<const float4*>(_b + 3 * _s); \
	} while (0)
#define FUSED_W_STORE(set, Wc_)                                                                                    \
	do {                                                                                                           \
		float* _d = (Wc_) + w_row * FW_LD + w_part * 4;                                                            \
		*reinterpret_cast<float4*>(_d) = set##0; *reinterpret_cast<float4*>(_d + 8 * FW_LD) = set##1;              \
		*reinterpret_cast<float4*>(_d + 16 * FW_LD) = set##2; *reinterpret_cast<float4*>(_d + 24 * FW_LD) = set##3; \
	} while (0)

template <int NT>
__global__ __launch_bounds__(512, 2) void fused_chain_kernel(const FusedArgs g) {
	constexpr int ROWS = 32 * NT, FUSED_X_BYTES = fused_x_bytes(NT);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* const X = reinterpret_cast<float*>(smem);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	float* const Wc0 = reinterpret_cast<float*>(smem + FUSED_X_BYTES + wave * 2 * FUSED_WS_BYTES);   // this wave's ring
	float* const Wc1 = Wc0 + 32 * FW_LD;
	float* const Bl = reinterpret_cast<float*>(smem + FUSED_X_BYTES + FUSED_NW * 2 * FUSED_WS_BYTES);
	const int li = lane & 31, lh = lane >> 5;
	const int V = g.V;
	const int w_row = lane >> 3, w_part = lane & 7;

	if (g.pe > 0) {
		for (int i = tid; i < 3 * g.pe; i += 512) Bl[i] = g.Bm[i];
	}
	int first_gemm = 0;
	while (first_gemm < g.n_steps && g.step[first_gemm].kind != FS_GEMM) ++first_gemm;

	for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
		const int foot = tile / g.tiles_per_foot;
		const int v0 = (tile - foot * g.tiles_per_foot) * ROWS;
		const int valid = min(ROWS, V - v0);
		const int64_t row0 = (int64_t)foot * V + v0;   // first global row of the tile

		// ---- prefetch cursor over the tile's chunk stream (wave-uniform scalars)
		int pf_si = first_gemm, pf_c = 0;
		const float* pf_w = g.step[pf_si < g.n_steps ? pf_si : 0].w;
		int pf_ldw = g.step[pf_si < g.n_steps ? pf_si : 0].ldw, pf_n = g.step[pf_si < g.n_steps ? pf_si : 0].nchunk;
		auto pf_advance = [&]() {   // past the end the cursor stays on the last chunk (re-fetched, never used)
			if (pf_c + 1 < pf_n) { ++pf_c; return; }
			int nx = pf_si + 1;
			while (nx < g.n_steps && g.step[nx].kind != FS_GEMM) ++nx;
			if (nx < g.n_steps) { pf_si = nx; pf_c = 0; pf_w = g.step[nx].w; pf_ldw = g.step[nx].ldw; pf_n = g.step[nx].nchunk; }
		};
		float4 wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3;   // (named registers: a struct or an array ends up in scratch)
		FUSED_W_LOAD(wa, pf_w, pf_ldw, pf_c); pf_advance();   // chunk 0
		FUSED_W_LOAD(wb, pf_w, pf_ldw, pf_c); pf_advance();   // chunk 1
		FUSED_W_STORE(wa, Wc0);                                // (the ring is this wave's own: its previous tile is done with it)
		// chunk k of the stream sits in ring stage k & 1 when its turn comes, chunk k + 1 in register set (k + 1) & 1

		f32x16 acc[NT];
		for (int si = 0; si < g.n_steps; ++si) {
			// the step's fields as scalars (indexing the kernel-argument array through a reference makes the compiler copy it to scratch)
			struct { const float *w, *bias, *src, *aux; float *dst, *dst2; int nchunk, bias_foot_stride, kind, src_kind, relu, mask, keep, accum, to_lds, head; } s;
			s.w = g.step[si].w; s.bias = g.step[si].bias; s.src = g.step[si].src; s.aux = g.step[si].aux; s.dst = g.step[si].dst; s.dst2 = g.step[si].dst2;
			s.nchunk = g.step[si].nchunk; s.bias_foot_stride = g.step[si].bias_foot_stride; s.kind = g.step[si].kind;
			s.src_kind = g.step[si].src_kind; s.relu = g.step[si].relu; s.mask = g.step[si].mask; s.keep = g.step[si].keep; s.accum = g.step[si].accum;
			s.to_lds = g.step[si].to_lds; s.head = g.step[si].head;
			if (s.kind == FS_OUT) {
				// final 256 -> 3 layer + tanh scaling on the X tile: 16 lanes per row, 16 columns each
				__syncthreads();
				const int seg = tid & 15;
#pragma unroll
				for (int rt = 0; rt < NT; ++rt) {
				const int row = rt * 32 + (tid >> 4);
				float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
				for (int c4 = 0; c4 < 4; ++c4) {
					const float4 x = *reinterpret_cast<const float4*>(X + row * FX_LD + seg * 16 + c4 * 4);
					const float4 a = *reinterpret_cast<const float4*>(s.w + 0 * W + seg * 16 + c4 * 4);
					const float4 b = *reinterpret_cast<const float4*>(s.w + 1 * W + seg * 16 + c4 * 4);
					const float4 c = *reinterpret_cast<const float4*>(s.w + 2 * W + seg * 16 + c4 * 4);
					p0 += x.x * a.x + x.y * a.y + x.z * a.z + x.w * a.w;
					p1 += x.x * b.x + x.y * b.y + x.z * b.z + x.w * b.w;
					p2 += x.x * c.x + x.y * c.y + x.z * c.z + x.w * c.w;
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
				continue;
			}
			// ---- GEMM step
			if (s.src_kind == FS_SRC_GLOBAL) {
				__syncthreads();   // everyone is done with the previous X
#pragma unroll
				for (int q = 0; q < 4 * NT; ++q) {
					const int idx = tid + 512 * q;
					const int row = idx >> 6, part = idx & 63;
					const float4 v = *reinterpret_cast<const float4*>(s.src + (row0 + min(row, valid - 1)) * W + part * 4);
					*reinterpret_cast<float4*>(X + row * FX_LD + part * 4) = v;
				}
			}
			if (!s.accum) {
#pragma unroll
				for (int rt = 0; rt < NT; ++rt)
#pragma unroll
					for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
			}
			__syncthreads();   // X is in place

			// The chunk loop is unrolled by two with the parity of the chunk index a COMPILE-TIME constant (every step has an even chunk
			// count, so parity in the stream = parity in the step): which register set receives chunk k + 2 and which one is stored must not
			// be a run-time choice -- the compiler turns `if (k & 1) load(set B) else load(set A)` into loads to temporaries, s_waitcnt
			// vmcnt(0) and v_cndmask moves, i.e. it waits for the prefetch it has just issued.
			const int nchunk = s.nchunk;
			const float* const xa0 = X + li * FX_LD + lh * 4;
			const float* const wfrag0 = Wc0 + li * FW_LD + lh * 4;
			const float* const wfrag1 = Wc1 + li * FW_LD + lh * 4;
#define FUSED_CHUNK(PAR, c_)                                                                                                   \
			do {                                                                                                               \
				if (!FIND_ABL(g.ablate, 1)) { if (PAR) FUSED_W_LOAD(wb, pf_w, pf_ldw, pf_c); else FUSED_W_LOAD(wa, pf_w, pf_ldw, pf_c); }   \
				pf_advance();                                                                                                  \
				__builtin_amdgcn_sched_barrier(0); /* the prefetch is issued BEFORE the MFMAs (else it sinks next to its store) */ \
				const float* xa = xa0 + (s.src_kind == FS_SRC_PE ? ((c_) & 7) : (c_)) * 32;                                    \
				const float* wbp = PAR ? wfrag1 : wfrag0;                                                                      \
				float4 fa[2][NT], fb[2];                                                                                       \
				_Pragma("unroll") for (int rt = 0; rt < NT; ++rt) fa[0][rt] = *reinterpret_cast<const float4*>(xa + rt * 32 * FX_LD); \
				fb[0] = *reinterpret_cast<const float4*>(wbp);                                                                  \
				_Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                             \
					const int cur = gq & 1, nxt = cur ^ 1;                                                                     \
					if (gq < 3) { /* fragments of k-group q+1 are requested before the MFMAs of group q */                      \
						_Pragma("unroll") for (int rt = 0; rt < NT; ++rt)                                                      \
							fa[nxt][rt] = *reinterpret_cast<const float4*>(xa + rt * 32 * FX_LD + (gq + 1) * 8);               \
						fb[nxt] = *reinterpret_cast<const float4*>(wbp + (gq + 1) * 8);                                        \
					}                                                                                                          \
					__builtin_amdgcn_sched_barrier(0);                                                                         \
					const float4 b = fb[cur];                                                                                  \
					if (!FIND_ABL(g.ablate, 2)) {                                                                                     \
						_Pragma("unroll") for (int rt = 0; rt < NT; ++rt) {                                                    \
							const float4 a = fa[cur][rt];                                                                      \
							acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[rt], 0, 0, 0);                        \
							acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[rt], 0, 0, 0);                        \
							acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[rt], 0, 0, 0);                        \
							acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[rt], 0, 0, 0);                        \
						}                                                                                                      \
					} else { acc[0][0] += fa[cur][0].x + b.x; }                                                                \
					__builtin_amdgcn_sched_barrier(0);                                                                         \
				}                                                                                                              \
				/* chunk k + 1 (requested one iteration ago) moves into the stage chunk k - 1 has left; the ring is private to */ \
				/* the wave, whose LDS operations complete in order: no barrier */                                              \
				if (!FIND_ABL(g.ablate, 1)) { if (PAR) FUSED_W_STORE(wa, Wc0); else FUSED_W_STORE(wb, Wc1); }                          \
			} while (0)
			for (int c = 0; c < nchunk; c += 2) {
				if (s.src_kind == FS_SRC_PE && (c & 7) == 0 && !FIND_ABL(g.ablate, 8)) {
					// regenerate the X tile with the Fourier features of k-tile c / 8; a thread fills 16 columns of one row
					if (c > 0) __syncthreads();   // the previous k-tile has been consumed by every wave
					const int seg = tid & 15;
#pragma unroll
					for (int rt = 0; rt < NT; ++rt) {
					const int row = rt * 32 + (tid >> 4);
					const float* pp = g.pos + (int64_t)foot * g.pos_foot_stride + (int64_t)(v0 + min(row, valid - 1)) * 3;
					const float px = pp[0], py = pp[1], pz = pp[2];
					// chunk cc of the padded order (mlp_kernels.h: pe_value): sin of 32 features, cos of the same 32, ..., [x y z 0 ...], zeros
					const int cc = (c >> 3) * 8 + (seg >> 1), nsc = g.pe >> 4, j0 = (seg & 1) * 16;
					float* xr = X + row * FX_LD + (seg >> 1) * 32 + j0;
					if (cc < nsc) {
						const float* b0 = Bl + (cc >> 1) * 32 + j0;
						const bool is_cos = cc & 1;
#pragma unroll 4
						for (int j = 0; j < 16; ++j) {
							const float t = 2.0f * fmaf(pz, b0[2 * g.pe + j], fmaf(py, b0[g.pe + j], px * b0[j]));
							xr[j] = is_cos ? cospif(t) : sinpif(t);
						}
					} else {
#pragma unroll
						for (int j = 0; j < 16; j += 4) *reinterpret_cast<float4*>(xr + j) = make_float4(0.f, 0.f, 0.f, 0.f);
						if (cc == nsc && j0 == 0) { xr[0] = px; xr[1] = py; xr[2] = pz; }
					}
					}
					__syncthreads();
				}
				FUSED_CHUNK(0, c);
				FUSED_CHUNK(1, c + 1);
			}
#undef FUSED_CHUNK
			if (s.keep || FIND_ABL(g.ablate, 4)) continue;

			// ---- epilogue: bias + ReLU / mask, store to HBM, hand the tile to the next step through LDS
			__syncthreads();   // every wave has multiplied its last chunk: nobody reads X any more
			{
				// (separate straight-line passes under wave-uniform branches: the four conditions inside one unrolled loop cost > 100 registers)
				const int col = wave * 32 + li;
				if (s.relu) {
					const float bv = s.bias[(int64_t)foot * s.bias_foot_stride + col];
#pragma unroll
					for (int rt = 0; rt < NT; ++rt)
#pragma unroll
						for (int r = 0; r < 16; ++r) acc[rt][r] = fmaxf(acc[rt][r] + bv, 0.f);
				}
				// buffer loads / stores as in gemm4: one VGPR offset, the row of element r in the immediate / scalar offset, and the
				// descriptor's size = the tile's valid bytes, so rows past the end of a foot are dropped by the bounds check
				// (per-row 64-bit addresses and clamps cost the 64-row tile its registers: 608 bytes of scratch)
				const int voff = ((4 * lh) * W + col) * 4;
				if (s.mask) {
					const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(s.aux + row0 * W)), 0, valid * W * 4, 0x00020000);
#pragma unroll
					for (int rt = 0; rt < NT; ++rt) {
						float mv[16];
#pragma unroll
						for (int r = 0; r < 16; ++r)
							mv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(msrc, voff + ((r & 3) * W) * 4, ((rt * 32 + 8 * (r >> 2)) * W) * 4, 0));
#pragma unroll
						for (int r = 0; r < 16; ++r) acc[rt][r] = (mv[r] > 0.f) ? acc[rt][r] : 0.f;   // (a row past the end reads 0: its value is never stored)
					}
				}
				if (s.dst) {
					const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(s.dst + row0 * W)), 0, valid * W * 4, 0x00020000);
#pragma unroll
					for (int rt = 0; rt < NT; ++rt)
#pragma unroll
						for (int r = 0; r < 16; ++r)
							__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[rt][r]), drs, voff + ((r & 3) * W) * 4, ((rt * 32 + 8 * (r >> 2)) * W) * 4, 0);
				}
				if (s.to_lds) {
#pragma unroll
					for (int rt = 0; rt < NT; ++rt)
#pragma unroll
						for (int r = 0; r < 16; ++r) X[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * FX_LD + col] = acc[rt][r];
				}
			}
			// (the next step starts with a barrier before anyone reads X)
		}
		__syncthreads();   // the next tile overwrites X
	}
}

}  // namespace mlp
}  // namespace find
