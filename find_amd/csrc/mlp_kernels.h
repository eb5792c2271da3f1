// fp32-MFMA kernels of the FIND MLP path (gfx950).  Included once by mlp.hip.
//
// Data layout in HBM: every activation is a row-major (rows, 256) fp32 matrix whose rows are
// (foot, vertex) pairs, foot-major.  Workgroup tiles never straddle two feet, so the per-foot latent
// contribution of the first head layer is a workgroup-uniform bias ("algebraically a
// per-foot bias") and the shared-template trunk is evaluated once for all feet.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/issue/SIMD, 157.3 TF peak).  Operand maps
// (cdna_hip_programming.md §3):  A lane l = A[i=l&31][k=l>>5],  B lane l = B[k=l>>5][j=l&31],
// D lane l reg r = D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].
#pragma once
#include "common.h"

namespace find {
namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Waves that own all 256 accumulator registers inside an allocation of fewer than 512 (300 - 328 registers per lane in the kernels
// that showed it) get wrong register contents when waves of another kernel are allocated on the same SIMD (mlp.hip, "Co-residence
// fault": reproduced at will in round 2, with and without LDS in the victim).  Policy: a kernel either stays within 256 registers or
// claims the whole file with this -- marking v255 clobbered makes the allocation 256 architectural + 256 accumulator registers = all 512
// of the SIMD, and no other wave fits beside it.
#define FIND_CLAIM_WHOLE_REGISTER_FILE() asm volatile("" ::: "v255")

constexpr int W = 256;      // hidden width (reference default, model.py:207)
constexpr int KC = 32;      // K chunk staged through LDS per step
constexpr int LDSLD = 36;   // padded LDS row stride (floats): ds_read_b128 of 16 rows hits 16 distinct 16-B slots
constexpr int KP0 = 768;    // padded K of the first trunk layer's repacked weight (3 k-tiles of 256)

// ---------------------------------------------------------------------------------------------
// Positional encoding in the PADDED K order used by layer 0:
//   chunk c (32 columns) for c < pe/16: even c -> sin(2*pi*t_f), odd c -> cos(2*pi*t_f), f = (c/2)*32 + j
//   then [x, y, z, 0...].   t_f = pos . B[:, f]   (fourier_feature_transform.py:44-52)
// sinpif/cospif do exact range reduction, so the result is at least as accurate as the reference's
// sin(fl(2*pi*fl(t))).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float pe_value(int kp, int pe, float x, float y, float z, const float* Bl) {
	const int c = kp >> 5, j = kp & 31;
	const int nsc = pe >> 4;
	if (c < nsc) {
		const int f = (c >> 1) * 32 + j;
		float t = fmaf(z, Bl[2 * pe + f], fmaf(y, Bl[pe + f], x * Bl[f]));
		t = 2.0f * t;
		return (c & 1) ? cospif(t) : sinpif(t);
	}
	const int q = kp - nsc * 32;
	return q == 0 ? x : (q == 1 ? y : (q == 2 ? z : 0.0f));
}

// padded column -> column of the reference weight base.0.weight (width, in_dim + 2*pe), or -1 for padding
__host__ __device__ __forceinline__ int pe_col_to_orig(int kp, int pe, int in_dim) {
	const int c = kp >> 5, j = kp & 31;
	const int nsc = pe >> 4;
	if (c < nsc) {
		const int f = (c >> 1) * 32 + j;
		return in_dim + ((c & 1) ? pe : 0) + f;
	}
	const int q = kp - nsc * 32;
	return q < in_dim ? q : -1;
}

// ---------------------------------------------------------------------------------------------
// GEMM  Y[(foot,v), n] = epi( sum_seg sum_k A_seg[(foot,v), k] * Wseg[n, k] ),  n in [0,256)
// Used for every forward Linear (EPI_BIAS_RELU) and every backward dX (EPI_MASK, W pre-transposed).
// Segments let the backward sum the contributions of both heads and -- when the trunk is shared by
// all feet -- of every foot, inside the K loop (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
	const float* a0;  // A base for segments [0, nseg_per_base)
	const float* a1;  // A base for segments [nseg_per_base, 2*nseg_per_base) (nbase == 2)
	int nbase;
	int nseg_per_base;
	int64_t a_seg_stride;   // elements between consecutive segments of one base
	int64_t a_foot_stride;  // elements between feet (blockIdx.y); 0 = rows shared by all feet
	int lda;
	const float* pos;       // AMODE_PE: (pos_batch, V, 3)
	int64_t pos_foot_stride;
	const float* Bm;        // AMODE_PE: (3, pe)
	int pe;
	const float* w0;        // (256, ldw), K contiguous
	const float* w1;
	int ldw;
	int w_tr;               // host side (launch_gemm): w0 is untransposed and the launch goes to gemm7, which reads it transposed (linear_bwd_dx)
	int nchunk;             // 32-wide K chunks per segment
	const float* bias;      // EPI_BIAS_RELU: (.., 256)
	int64_t bias_foot_stride;
	const float* mask;      // EPI_MASK: same layout as y; output zeroed where mask <= 0
	int64_t mask_foot_stride;
	float* y;
	int64_t y_foot_stride;
	int ldy;
	int V;                  // rows per foot
	int h16;                // host side only (launch_gemm): a0, y and mask are fp16-STORED tensors (act16: gemm5_kernel<EPI, true>)
	const float* va_bias;   // bcast_fold: a0 is the shared fp32 product P and the operand is relu(P[v] + va_bias[foot]) (gemm5_kernel<.., VIRT>)
	int64_t va_bias_stride;
	const float* vm_bias;   // bcast_fold: mask is P and the mask is P[v] + vm_bias[foot]
	int64_t vm_bias_stride;
	float* fs_out;          // footsum_fold (with vm_bias): no y; the sum over feet goes to fs_out (+ fs_slot_stride: second partial), the per-foot
	int64_t fs_slot_stride; // column sums to cs_out [workgroup pair][foot][256] (gemm7_kernel<.., FSUM>)
	float* cs_out;
};

enum { AMODE_MAT = 0, AMODE_PE = 1 };
enum { EPI_NONE = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2 };

// ---------------------------------------------------------------------------------------------
// Weight gradient  dW[n, k] = sum_rows dZ[row, n] * X[row, k]   (split over rows; partial slabs)
// One 512-thread workgroup owns a full 256(n) x 256(k) tile so dZ and X are each read once.
// k-tile 0 also produces the column sums of dZ (bias gradient / per-foot sums for the latent grads).
// ---------------------------------------------------------------------------------------------
struct DwArgs {
	const float* dz;        // rows (foot, v), ld 256
	int64_t dz_foot_stride;
	const float* x;         // AMODE_MAT: rows (foot, v) or shared (x_foot_stride 0)
	int64_t x_foot_stride;
	int ldx;
	const float* pos;       // AMODE_PE
	int64_t pos_foot_stride;
	const float* Bm;
	int pe;
	int V;
	int spf;                // splits per foot
	int cps;                // 32-row chunks per split
	int v_begin;            // first row of every foot handled by this launch (0, or 16*floor(V/16) for the tail pass)
	int Kp;                 // padded K = 256 * gridDim.x
	float* pw;              // [n_feet*spf][256][Kp]
	float* pb;              // [n_feet*spf][256] or nullptr
	int all_blocks;         // profiling only ("ablate" bit 32): multiply the padding blocks of the Fourier layer's last k-tile as well
};

template <int AMODE>
__global__ __launch_bounds__(512) void dw_kernel(const DwArgs g) {
	__shared__ __attribute__((aligned(16))) float Zs[32 * 256];
	__shared__ __attribute__((aligned(16))) float Xs[32 * 256];
	__shared__ float Bl[(AMODE == AMODE_PE) ? 3 * 256 : 4];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const int wn = wave >> 2;  // n rows wn*128 .. +127 (4 blocks)
	const int wk = wave & 3;   // k cols wk*64 .. +63   (2 blocks)
	const int kt = blockIdx.x;
	const int split = blockIdx.y;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int cpf = (g.V - g.v_begin + 31) / 32;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, cpf);
	const int lr = tid >> 6;         // loader row within an 8-row group
	const int lc = (tid & 63) * 4;   // loader column
	// 32-column blocks of this wave that hold any real column.  The Fourier layer's padded K is (pe / 16 + 1) chunks of 32 -- 544 of the
	// 768 columns its three k-tiles span: the last tile has ONE live block (x, y, z and 29 zeros); six of its eight waves own nothing
	// but padding, and multiplying it was a third of this kernel's matrix work.
	int nact = 2;
	if constexpr (AMODE == AMODE_PE) {
		const int live = ((g.pe >> 4) + 1) * 32 - (kt * 256 + wk * 64);
		nact = __builtin_amdgcn_readfirstlane((live >= 64 || g.all_blocks) ? 2 : (live > 0 ? 1 : 0));
	}

	if constexpr (AMODE == AMODE_PE) {
		for (int i = tid; i < 3 * g.pe; i += 512) Bl[i] = g.Bm[i];
		__syncthreads();
	}

	f32x16 acc[4][2];
#pragma unroll
	for (int mi = 0; mi < 4; ++mi)
#pragma unroll
		for (int ni = 0; ni < 2; ++ni)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

	float4 rz[4], rx[4];
	float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
	const bool do_bias = (g.pb != nullptr) && (kt == 0);
	const float* dzp = g.dz + (int64_t)foot * g.dz_foot_stride + lc;
	const float* xp = (AMODE == AMODE_MAT) ? g.x + (int64_t)foot * g.x_foot_stride + kt * 256 + lc : nullptr;
	const float* pp = (AMODE == AMODE_PE) ? g.pos + (int64_t)foot * g.pos_foot_stride : nullptr;

	auto load_chunk = [&](int q) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int v = g.v_begin + q * 32 + lr + 8 * i;
			const bool ok = v < g.V;
			rz[i] = ok ? *reinterpret_cast<const float4*>(dzp + (int64_t)v * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
			if constexpr (AMODE == AMODE_MAT) {
				rx[i] = ok ? *reinterpret_cast<const float4*>(xp + (int64_t)v * g.ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
			} else {
				if (ok) {
					const float x = pp[(int64_t)v * 3 + 0], y = pp[(int64_t)v * 3 + 1], z = pp[(int64_t)v * 3 + 2];
					const int kp = kt * 256 + lc;
					rx[i].x = pe_value(kp + 0, g.pe, x, y, z, Bl);
					rx[i].y = pe_value(kp + 1, g.pe, x, y, z, Bl);
					rx[i].z = pe_value(kp + 2, g.pe, x, y, z, Bl);
					rx[i].w = pe_value(kp + 3, g.pe, x, y, z, Bl);
				} else {
					rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
				}
			}
		}
	};

	if (q0 < q1) load_chunk(q0);
	for (int q = q0; q < q1; ++q) {
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			*reinterpret_cast<float4*>(&Zs[(lr + 8 * i) * 256 + lc]) = rz[i];
			*reinterpret_cast<float4*>(&Xs[(lr + 8 * i) * 256 + lc]) = rx[i];
			if (do_bias) {
				bsum.x += rz[i].x; bsum.y += rz[i].y; bsum.z += rz[i].z; bsum.w += rz[i].w;
			}
		}
		__syncthreads();
		// The two waves of a SIMD (w and w + 4) take the two halves of the iteration in opposite orders: one regenerates the next chunk's
		// operands (global loads, sin / cos: VALU) while the other multiplies (matrix pipe).  In the same order both waited for their
		// loads together and then queued for the matrix pipe together (PMC: the pipe 22 % busy, the waves waiting 80 % of their time).
		const bool load_first = wave < 4;
		if (load_first && q + 1 < q1) load_chunk(q + 1);

		const int zo = (lane >> 5) * 256 + wn * 128 + (lane & 31);
		const int xo = (lane >> 5) * 256 + wk * 64 + (lane & 31);
		if (nact > 0) {   // (a wave whose two blocks are both padding skips the chunk; one live block still multiplies both: a third
			            //  code path for it made the compiler spill)
#pragma unroll
			for (int t = 0; t < 16; ++t) {
				float a[4], b[2];
#pragma unroll
				for (int mi = 0; mi < 4; ++mi) a[mi] = Zs[zo + t * 512 + mi * 32];
#pragma unroll
				for (int ni = 0; ni < 2; ++ni) b[ni] = Xs[xo + t * 512 + ni * 32];
#pragma unroll
				for (int mi = 0; mi < 4; ++mi)
#pragma unroll
					for (int ni = 0; ni < 2; ++ni)
						acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
			}
		}
		if (!load_first && q + 1 < q1) load_chunk(q + 1);
	}

	float* pw = g.pw + (int64_t)split * 256 * g.Kp + kt * 256;
#pragma unroll
	for (int mi = 0; mi < 4; ++mi)
#pragma unroll
		for (int ni = 0; ni < 2; ++ni)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				if (ni >= nact) continue;   // (padding columns: the slab reduce never maps them to an output)
				const int n = wn * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
				const int k = wk * 64 + ni * 32 + (lane & 31);
				pw[(int64_t)n * g.Kp + k] = acc[mi][ni][r];
			}

	if (do_bias) {
		__syncthreads();
		float* red = Zs;  // [8][256]
		*reinterpret_cast<float4*>(&red[lr * 256 + lc]) = bsum;
		__syncthreads();
		if (tid < 256) {
			float s = 0.f;
#pragma unroll
			for (int w = 0; w < 8; ++w) s += red[w * 256 + tid];
			g.pb[(int64_t)split * 256 + tid] = s;
		}
	}
}

// dW[n, map(k)] = sum_split pw[split][n][k]: blocks [0, nwblk), 64 float4 outputs each, the slab range cut 16 ways
// across the 1024 threads and combined through LDS in a fixed order (deterministic).  Bias / per-foot sums: block
// nwblk + f -> S[f][:], block nwblk + n_feet -> db.
struct ReduceWArgs {
	const float* pw;
	int nsplit;
	int Kp;
	float* out;
	int ld_out;
	int K_valid;   // identity map: columns [0, K_valid) are written
	int pe_map;    // 1: padded PE layout -> reference column order
	int pe;
	int in_dim;
	const float* pb;  // [n_feet*spf][256] or nullptr
	int n_feet, spf;
	float* db;        // (256) or nullptr
	float* S;         // (n_feet,256) or nullptr
	int nwblk;        // 256*Kp/4/64
};

__device__ __forceinline__ void reduce_w_body(const ReduceWArgs& g) {
	__shared__ __attribute__((aligned(16))) float red[16 * 256];
	const int tid = threadIdx.x;
	if ((int)blockIdx.x >= g.nwblk) {
		// bias: S[foot][n] = sum over that foot's splits;  db[n] = sum over every split
		if (g.pb == nullptr) return;
		const int f = (int)blockIdx.x - g.nwblk;
		const int n = tid & 255, q = tid >> 8;
		const bool all = f >= g.n_feet;
		if (!all && g.S == nullptr) return;
		if (all && g.db == nullptr) return;
		const int lo = all ? 0 : f * g.spf, hi = all ? g.n_feet * g.spf : (f + 1) * g.spf;
		float s0 = 0.f, s1 = 0.f;
		int k = lo + q;
		for (; k + 4 < hi; k += 8) { s0 += g.pb[(int64_t)k * 256 + n]; s1 += g.pb[(int64_t)(k + 4) * 256 + n]; }
		if (k < hi) s0 += g.pb[(int64_t)k * 256 + n];
		red[q * 256 + n] = s0 + s1;
		__syncthreads();
		if (q == 0) {
			const float t = (red[n] + red[256 + n]) + (red[512 + n] + red[768 + n]);
			if (all) g.db[n] = t;
			else g.S[(int64_t)f * 256 + n] = t;
		}
		return;
	}
	const int o = tid & 63, q = tid >> 6;  // output float4 within the block, slab slice
	const int64_t i4 = ((int64_t)blockIdx.x * 64 + o) * 4;
	const int n = (int)(i4 / g.Kp), k = (int)(i4 - (int64_t)n * g.Kp);
	float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
	const float* p = g.pw + i4;
	const int64_t stride = (int64_t)256 * g.Kp;
	int sidx = q;
	for (; sidx + 48 < g.nsplit; sidx += 64) {
		const float4 a = *reinterpret_cast<const float4*>(p + (sidx + 0) * stride);
		const float4 b = *reinterpret_cast<const float4*>(p + (sidx + 16) * stride);
		const float4 c = *reinterpret_cast<const float4*>(p + (sidx + 32) * stride);
		const float4 d = *reinterpret_cast<const float4*>(p + (sidx + 48) * stride);
		s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
		s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
	}
	for (; sidx < g.nsplit; sidx += 16) {
		const float4 a = *reinterpret_cast<const float4*>(p + sidx * stride);
		s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
	}
	*reinterpret_cast<float4*>(&red[(q * 64 + o) * 4]) = s;
	__syncthreads();
	if (tid < 256) {
		// thread -> scalar output (o2, j): conflict-free LDS columns
		const int o2 = tid >> 2, j = tid & 3;
		float t = 0.f;
#pragma unroll
		for (int qq = 0; qq < 16; ++qq) t += red[(qq * 64 + o2) * 4 + j];
		const int64_t e = ((int64_t)blockIdx.x * 64 + o2) * 4 + j;
		const int nn = (int)(e / g.Kp), kk = (int)(e - (int64_t)nn * g.Kp);
		int ko;
		if (g.pe_map) ko = pe_col_to_orig(kk, g.pe, g.in_dim);
		else ko = (kk < g.K_valid) ? kk : -1;
		if (ko >= 0) g.out[(int64_t)nn * g.ld_out + ko] = t;
	}
	(void)n; (void)k;
}

__global__ __launch_bounds__(1024) void reduce_w_kernel(const ReduceWArgs g) { reduce_w_body(g); }

// Diagnosis only ("reduce_exclusive" = 2): the same sums with NO LDS -- a thread owns its outputs and walks the slabs alone (same grid as
// reduce_w_kernel, the extra threads idle).  If the rare wrong weight-gradient elements that appear without the whole-LDS reservation
// (mlp.hip) need the reduce to USE LDS beside a ring kernel, they must vanish under this kernel; if they are a race on the slab buffers,
// they must not.  (The summation order differs from reduce_w_kernel's: compare runs of this kernel with each other.)
__global__ __launch_bounds__(1024) void reduce_w_nolds_kernel(const ReduceWArgs g) {
	const int tid = threadIdx.x;
	if ((int)blockIdx.x >= g.nwblk) {
		if (g.pb == nullptr || tid >= 256) return;
		const int f = (int)blockIdx.x - g.nwblk;
		const bool all = f >= g.n_feet;
		if ((!all && g.S == nullptr) || (all && g.db == nullptr)) return;
		const int lo = all ? 0 : f * g.spf, hi = all ? g.n_feet * g.spf : (f + 1) * g.spf;
		float s = 0.f;
		for (int k = lo; k < hi; ++k) s += g.pb[(int64_t)k * 256 + tid];
		if (all) g.db[tid] = s;
		else g.S[(int64_t)f * 256 + tid] = s;
		return;
	}
	if (tid >= 256) return;
	const int64_t e = (int64_t)blockIdx.x * 256 + tid;   // scalar output of the block's 64 float4
	const int nn = (int)(e / g.Kp), kk = (int)(e - (int64_t)nn * g.Kp);
	const int64_t stride = (int64_t)256 * g.Kp;
	float t = 0.f;
	for (int sidx = 0; sidx < g.nsplit; ++sidx) t += g.pw[e + sidx * stride];
	int ko;
	if (g.pe_map) ko = pe_col_to_orig(kk, g.pe, g.in_dim);
	else ko = (kk < g.K_valid) ? kk : -1;
	if (ko >= 0) g.out[(int64_t)nn * g.ld_out + ko] = t;
}

// the slab reduces of a grouped weight-gradient launch (mlp_dw2.h: dw2_group_kernel), blockIdx.y = job
constexpr int REDUCE_MAX_JOBS = 24;
struct ReduceWGroup { ReduceWArgs job[REDUCE_MAX_JOBS]; };
__global__ __launch_bounds__(1024) void reduce_w_group_kernel(const ReduceWGroup grp) {
	const int j = blockIdx.y;
	ReduceWArgs g;
	g.pw = grp.job[j].pw; g.nsplit = grp.job[j].nsplit; g.Kp = grp.job[j].Kp; g.out = grp.job[j].out; g.ld_out = grp.job[j].ld_out;
	g.K_valid = grp.job[j].K_valid; g.pe_map = grp.job[j].pe_map; g.pe = grp.job[j].pe; g.in_dim = grp.job[j].in_dim; g.pb = grp.job[j].pb;
	g.n_feet = grp.job[j].n_feet; g.spf = grp.job[j].spf; g.db = grp.job[j].db; g.S = grp.job[j].S; g.nwblk = grp.job[j].nwblk;
	reduce_w_body(g);
}

// Exact zeros into up to ZERO_MAX_JOBS buffers in one launch (grid (blocks, jobs)): the gradients of the parts a call skipped.
constexpr int ZERO_MAX_JOBS = 24;
struct ZeroArgs {
	float* p[ZERO_MAX_JOBS];
	int64_t n[ZERO_MAX_JOBS];
	int njobs;
};
__global__ __launch_bounds__(256) void zero_many_kernel(const ZeroArgs a) {
	float* __restrict__ p = a.p[blockIdx.y];
	const int64_t n = a.n[blockIdx.y];
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

// Shared-template backward of a head's first layer: because every foot multiplies the SAME trunk rows,
//   sum_b dZ0[b,v,:] @ W  ==  (sum_b dZ0[b,v,:]) @ W      and      dW0 = (sum_b dZ0[b])^T @ H.
// One pass over dZ0 (n_feet, V, 256) produces  zsum[v] = sum_b dZ0[b,v]  and partial per-foot column sums.
constexpr int FS_ROWS = 16;  // rows of v per block
constexpr int FS_FEET = 16;  // feet per round: that many independent 16-byte loads in flight per thread
// fp16-STORED tensors of the opt-in fp16 mode ("act16", mlp.hip): the heads' hidden activations and their gradients at the large
// shared-template shapes live in HBM as fp16 -- the matrix pipe rounds them to fp16 anyway, and those layers are bound by HBM.  Four
// consecutive elements of such a tensor (or of an fp32 one: `half` is uniform) as a float4, and back.
typedef _Float16 find_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_any(const void* base, int64_t elem, int half) {
	if (half) {
		const find_h4 h = *reinterpret_cast<const find_h4*>(reinterpret_cast<const _Float16*>(base) + elem);
		return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
	}
	return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
}
__device__ __forceinline__ void st4_any(void* base, int64_t elem, int half, const float4& v) {
	if (half) *reinterpret_cast<find_h4*>(reinterpret_cast<_Float16*>(base) + elem) = find_h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
	else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem) = v;
}

// v + its three counterparts 16, 32 and 48 lanes away (the four 16-lane rows of a wave), in every lane: gfx950's v_permlane16_swap /
// v_permlane32_swap on two copies of the value -- VALU only, and bit for bit the xor-16, xor-32 butterfly through __shfl_xor it replaces
// (128 ds_bpermute per thread and round of feet: the LDS pipe, not HBM, bounded footsum_kernel -- 2.4 TB/s at the C5 shape, round 6).
__device__ __forceinline__ float rows4_sum(float v) {
	const unsigned u = __float_as_uint(v);
	const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // rows (0, 1, 2, 3) -> (0, 0, 2, 2) and (1, 1, 3, 3)
	const float t = __uint_as_float(p[0]) + __uint_as_float(p[1]);
	const unsigned w = __float_as_uint(t);
	const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);     // lower half twice, upper half twice
	return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// grid (4 ceil(V/16)): block = 16 rows x 64 columns (row block blockIdx.x / 4, column quarter blockIdx.x % 4); thread (row r = tid>>4, column group cg = tid&15) walks the feet.
// (An LDS-free variant -- one wave per 16 columns, the 16 rows summed by shuffles, so that it could share CUs with the ring kernels
// that claim the whole LDS -- made the step slower, 2.30 against 2.26 ms: 64-byte row segments and the interference cost more than
// the wait for a CU.)
__global__ __launch_bounds__(256) void footsum_kernel(const float* __restrict__ dz, int n_feet, int V, float* __restrict__ zsum,
													   float* __restrict__ pS /* [gridDim.x][n_feet][256] */, int dz_half) {
	__shared__ __attribute__((aligned(16))) float red[4][FS_FEET][64];
	const int cg = threadIdx.x & 15, r = threadIdx.x >> 4, wave = threadIdx.x >> 6;
	// (the four column quarters of the same 16 rows are CONSECUTIVE workgroups: with the quarter in blockIdx.y a launch read one 128-byte
	// quarter of every 512-byte fp16 row, then the next quarter a whole grid later -- a quarter of every DRAM row per visit: 2.3 TB/s at C5)
	const int blk = blockIdx.x >> 2, quarter = blockIdx.x & 3;
	const int v = blk * FS_ROWS + r;
	const int c0 = quarter * 64 + cg * 4;
	const bool live = v < V;
	float4 zs = make_float4(0.f, 0.f, 0.f, 0.f);
	for (int b0 = 0; b0 < n_feet; b0 += FS_FEET) {
		const int nb = min(FS_FEET, n_feet - b0);
		float4 x[FS_FEET];
#pragma unroll
		for (int bb = 0; bb < FS_FEET; ++bb)
			x[bb] = (live && bb < nb) ? ld4_any(dz, ((int64_t)(b0 + bb) * V + v) * 256 + c0, dz_half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
		for (int bb = 0; bb < FS_FEET; ++bb) {
			zs.x += x[bb].x; zs.y += x[bb].y; zs.z += x[bb].z; zs.w += x[bb].w;
			// per-foot column sums: the 4 rows of this wave by shuffles (lanes 16 apart), the 4 waves through LDS
			float4 a = x[bb];
			a.x = rows4_sum(a.x); a.y = rows4_sum(a.y); a.z = rows4_sum(a.z); a.w = rows4_sum(a.w);
			if ((threadIdx.x & 63) < 16) *reinterpret_cast<float4*>(&red[wave][bb][cg * 4]) = a;
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nb * 64; i += 256) {
			const int bb = i >> 6, c = i & 63;
			pS[((int64_t)blk * n_feet + b0 + bb) * 256 + quarter * 64 + c] = (red[0][bb][c] + red[1][bb][c]) + (red[2][bb][c] + red[3][bb][c]);
		}
		__syncthreads();
	}
	if (live && zsum != nullptr) *reinterpret_cast<float4*>(zsum + (int64_t)v * 256 + c0) = zs;   // (no zsum: only the per-foot column sums are wanted)
}

// S[b][n] = sum_blk pS[blk][b][n]: grid (feet, 4 column quarters); 1024 threads = 64 columns x 16 slices of the block range
// (one block per foot left 16 workgroups to read 51 MB at the 50 002-vertex template: 318 us).
__global__ __launch_bounds__(1024) void footsum_reduce_kernel(const float* __restrict__ pS, int nblk, int n_feet, float* __restrict__ S) {
	__shared__ float red[16][64];
	const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
	const int f = blockIdx.x, n = blockIdx.y * 64 + c;
	const float* p = pS + (int64_t)f * 256 + n;
	const int64_t stride = (int64_t)n_feet * 256;
	// latency-bound (strided loads): 8 independent loads in flight per iteration
	float acc[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) acc[i] = 0.f;
	int k = q;
	for (; k + 112 < nblk; k += 128) {
#pragma unroll
		for (int i = 0; i < 8; ++i) acc[i] += p[(int64_t)(k + 16 * i) * stride];
	}
	for (; k < nblk; k += 16) acc[0] += p[(int64_t)k * stride];
#pragma unroll
	for (int i = 4; i >= 1; i >>= 1)
#pragma unroll
		for (int j = 0; j < i; ++j) acc[j] += acc[j + i];
	red[q][c] = acc[0];
	__syncthreads();
	if (q == 0) {
		float t[16];
#pragma unroll
		for (int i = 0; i < 16; ++i) t[i] = red[i][c];
#pragma unroll
		for (int i = 8; i >= 1; i >>= 1)
#pragma unroll
			for (int j = 0; j < i; ++j) t[j] += t[j + i];
		S[(int64_t)f * 256 + n] = t[0];
	}
}

// a += b (float4 granules): the two partial foot sums of gemm7_kernel<.., FSUM> -> the foot sum
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n4) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n4) return;
	float4 x = reinterpret_cast<float4*>(a)[i];
	const float4 y = reinterpret_cast<const float4*>(b)[i];
	x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
	reinterpret_cast<float4*>(a)[i] = x;
}

// db[n] = sum_b S[b][n]
__global__ void colsum_small_kernel(const float* __restrict__ S, int n_feet, float* __restrict__ db) {
	const int n = threadIdx.x;
	float s = 0.f;
	for (int b = 0; b < n_feet; ++b) s += S[(int64_t)b * 256 + n];
	db[n] = s;
}

// ---------------------------------------------------------------------------------------------
// Weight repacking (one launch, grid.y = job): copy a column block / transpose / PE permutation.
// ---------------------------------------------------------------------------------------------
struct RepackJob {
	const float* src;
	float* dst;
	int rows, cols;   // extent of the source block
	int ld_src, coff; // source row stride and first column
	int ld_dst;
	int mode;         // 0 copy, 1 transpose (dst[c][r]), 2 PE permute into KP0 padded columns
	int pe, in_dim;
};
constexpr int MAX_REPACK = 28;
struct RepackArgs {
	RepackJob job[MAX_REPACK];
	int njobs;
};

__global__ void repack_kernel(const RepackArgs a) {
	const RepackJob& j = a.job[blockIdx.y];
	if (j.mode == 2) {
		const int64_t total = (int64_t)j.rows * KP0;
		for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
			const int r = (int)(i / KP0), kp = (int)(i - (int64_t)r * KP0);
			const int ko = pe_col_to_orig(kp, j.pe, j.in_dim);
			j.dst[(int64_t)r * j.ld_dst + kp] = (ko >= 0) ? j.src[(int64_t)r * j.ld_src + ko] : 0.f;
		}
		return;
	}
	const int64_t total = (int64_t)j.rows * j.cols;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
		const int r = (int)(i / j.cols), c = (int)(i - (int64_t)r * j.cols);
		const float v = j.src[(int64_t)r * j.ld_src + j.coff + c];
		if (j.mode == 0) j.dst[(int64_t)r * j.ld_dst + c] = v;
		else j.dst[(int64_t)c * j.ld_dst + r] = v;
	}
}

// ---------------------------------------------------------------------------------------------
// Per-foot latent bias of a head's first layer (model.py:428-437 folded into a bias):
//   fb[foot][n] = b[n] + sum_j Wfull[n][256 + j] * lat[foot][j]
// ---------------------------------------------------------------------------------------------
// grid (64, n_feet) x 256 threads: wave -> output n of one foot, lanes over the latent dimension (coalesced row of Wfull).
__global__ __launch_bounds__(256) void latent_bias_kernel(const float* Wfull, int ldw, const float* b, const float* lat, int L, float* fb) {
	const int lane = threadIdx.x & 63;
	const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int foot = blockIdx.y;
	const float* wr = Wfull + (int64_t)n * ldw + W;
	const float* lv = lat + (int64_t)foot * L;
	float s = 0.f;
	for (int j = lane; j < L; j += 64) s = fmaf(wr[j], lv[j], s);
	s = wave_sum(s);
	if (lane == 0) fb[(int64_t)foot * W + n] = b[n] + s;
}

// latent gradients from S[foot][n] = sum_v dZ0[(foot,v)][n]:
//   dlat[foot][j] = sum_n S[foot][n] * Wfull[n][256+j];   dWfull[n][256+j] = sum_foot S[foot][n] * lat[foot][j]
// 256 threads.  Blocks [0, n_feet): 4 groups of 64 lanes split n, lanes over j (coalesced), LDS-combined.
// Blocks [n_feet, n_feet + 256): one output row n of dWfull's latent columns.
// Block n_feet + 256 (launched only when db != nullptr): db[n] = sum_foot S[foot][n].
__global__ __launch_bounds__(256) void latent_grad_kernel(const float* Wfull, int ldw, const float* lat, int L, const float* S, int n_feet,
														   float* dlat, float* dWfull, float* db) {
	__shared__ float red[4][64];
	const int b = blockIdx.x;
	const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
	if (b == n_feet + W) {
		float s = 0.f;
		for (int f = 0; f < n_feet; ++f) s += S[(int64_t)f * W + threadIdx.x];
		db[threadIdx.x] = s;
		return;
	}
	if (b < n_feet) {
		if (dlat == nullptr) return;
		for (int j0 = 0; j0 < L; j0 += 64) {
			const int j = j0 + lane;
			float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
			if (j < L) {
				const float* wp = Wfull + (int64_t)(q * 64) * ldw + W + j;
				const float* sp = S + (int64_t)b * W + q * 64;
#pragma unroll
				for (int n = 0; n < 64; n += 4) {
					s0 = fmaf(sp[n + 0], wp[(int64_t)(n + 0) * ldw], s0);
					s1 = fmaf(sp[n + 1], wp[(int64_t)(n + 1) * ldw], s1);
					s2 = fmaf(sp[n + 2], wp[(int64_t)(n + 2) * ldw], s2);
					s3 = fmaf(sp[n + 3], wp[(int64_t)(n + 3) * ldw], s3);
				}
			}
			red[q][lane] = (s0 + s1) + (s2 + s3);
			__syncthreads();
			if (q == 0 && j < L) dlat[(int64_t)b * L + j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
			__syncthreads();
		}
	} else {
		const int n = b - n_feet;  // 0..255
		for (int j = threadIdx.x; j < L; j += blockDim.x) {
			float s = 0.f;
			for (int f = 0; f < n_feet; ++f) s = fmaf(S[(int64_t)f * W + n], lat[(int64_t)f * L + j], s);
			dWfull[(int64_t)n * ldw + W + j] = s;
		}
	}
}

// out[foot][v][:] = relu(P[v][:] + bias[foot][:]) : first head layer with a shared template (P = H W^T computed once on V rows).
// grid (ceil(V*64/256), n_feet-groups): a thread keeps its float4 of P in registers and writes it for FEET_PER feet (P is read
// once per group instead of once per foot; the writes are the traffic: n_feet * V * 1 KB).
constexpr int BCAST_FEET = 4;
__global__ __launch_bounds__(256) void bias_relu_bcast_kernel(const float* __restrict__ P, const float* __restrict__ bias, int64_t bias_foot_stride,
															   int n_feet, int64_t V, float* __restrict__ out, int out_half) {
	const int64_t per_foot = V * (W / 4);
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= per_foot) return;
	const int c4 = (int)(r & (W / 4 - 1));
	const float4 p = reinterpret_cast<const float4*>(P)[r];
	const int f0 = blockIdx.y * BCAST_FEET;
#pragma unroll
	for (int k = 0; k < BCAST_FEET; ++k) {
		const int foot = f0 + k;
		if (foot >= n_feet) break;
		const float4 b = *reinterpret_cast<const float4*>(bias + foot * bias_foot_stride + c4 * 4);
		float4 o;
		o.x = fmaxf(p.x + b.x, 0.f); o.y = fmaxf(p.y + b.y, 0.f); o.z = fmaxf(p.z + b.z, 0.f); o.w = fmaxf(p.w + b.w, 0.f);
		st4_any(out, (foot * per_foot + r) * 4, out_half, o);
	}
}

// ---------------------------------------------------------------------------------------------
// Final 256->3 layers of both heads + output activations (model.py:439-451).  One wave per row.
//   head 0: disp = 0.1*tanh(z)        head 1: col = 0.5*(1+tanh(z)) [+ avg_col]
// ---------------------------------------------------------------------------------------------
struct HeadOutArgs {
	const float* x[2];    // (rows, 256) last hidden activation of each head
	const float* w[2];    // (3, 256)
	const float* b[2];    // (3)
	float* z[2];          // (rows, 3) pre-activation, saved for backward (may be null)
	float* out[2];        // (rows, 3)
	const float* avg_col; // or null
	int64_t rows;
	int head0;            // head of blockIdx.y == 0 (the two heads can be launched separately, grid.y = 1)
	int x_half;           // x is stored as fp16 (act16)
};

// Sum over the 16 lanes of a DPP row, in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror -- VALU-only, and bit for
// bit the xor-1, 2, 4, 8 butterfly it replaces (after the quad steps the four lanes of a quad hold one value, so the mirror pairings add
// the same partner sums).  The butterfly through __shfl_xor was 24 ds_bpermute per 8 rows: the LDS pipe, not HBM, bounded the kernel
// once its input was fp16-stored.
__device__ __forceinline__ float row16_sum(float t) {
	t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xF, 0xF, false));
	t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xF, 0xF, false));
	t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x141, 0xF, 0xF, false));
	t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x140, 0xF, 0xF, false));
	return t;
}

// 16 lanes per row (4 rows per wave, 2 row groups in flight): lane part p holds columns 4p + 64i, i = 0..3.
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const HeadOutArgs g) {
	const int head = blockIdx.y + g.head0;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const float* x = head ? g.x[1] : g.x[0];
	const float* w = head ? g.w[1] : g.w[0];
	const float* b = head ? g.b[1] : g.b[0];
	float* z = head ? g.z[1] : g.z[0];
	float* out = head ? g.out[1] : g.out[0];
	if (out == nullptr) return;
	const int part = lane & 15, sub = lane >> 4;
	float4 wv[3][4];
#pragma unroll
	for (int c = 0; c < 3; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) wv[c][i] = *reinterpret_cast<const float4*>(w + c * W + part * 4 + 64 * i);
	const int c3 = part < 3 ? part : 0;
	const float bc = b[c3];
	const float av = (head && g.avg_col) ? g.avg_col[c3] : 0.f;
	constexpr int U = 2;
	for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * (4 * U); r0 < g.rows; r0 += (int64_t)gridDim.x * 4 * (4 * U)) {
		float4 xv[U][4];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t row = r0 + u * 4 + sub;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				xv[u][i] = (row < g.rows) ? ld4_any(x, row * W + part * 4 + 64 * i, g.x_half) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t row = r0 + u * 4 + sub;
			float sc[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				float t = 0.f;
#pragma unroll
				for (int i = 0; i < 4; ++i) t += xv[u][i].x * wv[c][i].x + xv[u][i].y * wv[c][i].y + xv[u][i].z * wv[c][i].z + xv[u][i].w * wv[c][i].w;
				sc[c] = row16_sum(t);
			}
			if (part < 3 && row < g.rows) {
				const float zz = (part == 0 ? sc[0] : (part == 1 ? sc[1] : sc[2])) + bc;
				const float t = tanhf(zz);
				if (z) z[row * 3 + part] = zz;
				out[row * 3 + part] = head ? (av + 0.5f * (1.0f + t)) : 0.1f * t;
			}
		}
	}
}

// The same for fp16-STORED activations (act16, the opt-in fp16 mode at the large shared-template shapes) on the fp16 matrix pipe: a row of
// x is 512 B, and the fp32 kernel above spends more instructions converting, multiplying and DPP-summing a row than the memory system
// needs to deliver it (2.8 TB/s at the C5 shape, round 6).  Here a wave takes 16-row blocks: x is the MFMA's row operand AS IT LIES IN
// MEMORY (v_mfma_f32_16x16x32_f16: lane (row l % 16, k quarter l / 16) loads 16 bytes = 8 consecutive k), W the column operand, split
// into an fp16 pair hi + lo of 16 w (exact to 2^-22 |w|; the factor keeps lo out of the fp16 subnormals) held in registers for the whole
// run -- 16 MFMAs and 8 loads per block and lane, four blocks in flight.  fp32 accumulation; the three live columns of the 16 x 16
// result (lanes l % 16 < 3, four rows each) go through the same bias / tanh epilogue.
typedef _Float16 find_h8 __attribute__((ext_vector_type(8)));
typedef float find_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void head_out_fwd_h16_kernel(const HeadOutArgs g) {
	const int head = blockIdx.y + g.head0;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const _Float16* x = reinterpret_cast<const _Float16*>(head ? g.x[1] : g.x[0]);
	const float* w = head ? g.w[1] : g.w[0];
	const float* b = head ? g.b[1] : g.b[0];
	float* z = head ? g.z[1] : g.z[0];
	float* out = head ? g.out[1] : g.out[0];
	if (out == nullptr) return;
	const int m = lane & 15, kq = lane >> 4;
	const int ch = m < 3 ? m : 0;
	find_h8 wh[8], wl[8];
#pragma unroll
	for (int s = 0; s < 8; ++s)
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const float v = m < 3 ? 16.0f * w[ch * W + 32 * s + 8 * kq + j] : 0.f;
			const _Float16 hi = (_Float16)v;
			wh[s][j] = hi;
			wl[s][j] = (_Float16)(v - (float)hi);
		}
	const float bc = b[ch];
	const float av = (head && g.avg_col) ? g.avg_col[ch] : 0.f;
	constexpr int U = 4;
	const int64_t nblk = (g.rows + 15) >> 4;
	for (int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * U; b0 < nblk; b0 += (int64_t)gridDim.x * 4 * U) {
		find_h8 xa[U][8];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t row = min((b0 + u) * 16 + m, g.rows - 1);   // (rows past the end re-read the last row: never stored)
			const find_h8* p = reinterpret_cast<const find_h8*>(x + row * W + 8 * kq);
#pragma unroll
			for (int s = 0; s < 8; ++s) xa[u][s] = p[4 * s];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			find_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int s = 0; s < 8; ++s) {
				acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[u][s], wl[s], acc, 0, 0, 0);   // (small terms first)
				acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[u][s], wh[s], acc, 0, 0, 0);
			}
			if (m < 3) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int64_t row = (b0 + u) * 16 + 4 * kq + i;
					if (row < g.rows) {
						const float zz = acc[i] * 0.0625f + bc;
						const float t = tanhf(zz);
						if (z) z[row * 3 + m] = zz;
						out[row * 3 + m] = head ? (av + 0.5f * (1.0f + t)) : 0.1f * t;
					}
				}
			}
		}
	}
}

// Backward of the final layers: dz = g * act'(z);  dY[row][k] = (sum_c dz_c W[c][k]) * (Y[row][k] > 0);
// partial dW[c][k] = sum_rows dz_c * Y[row][k], partial db[c] = sum_rows dz_c   (per workgroup slabs).
struct HeadOutBwdArgs {
	const float* y[2];     // (rows,256) last hidden activation
	const float* w[2];     // (3,256)
	const float* z[2];     // (rows,3)
	const float* gout[2];  // (rows,3) upstream gradient, null -> head skipped
	float* dy[2];          // (rows,256) masked gradient wrt last hidden pre-activation
	float* pw[2];          // [gridDim.x][3][256]
	float* pb[2];          // [gridDim.x][4]
	int64_t rows;
	int half;              // y and dy are stored as fp16 (act16)
};

template <int U>   // rows in flight per wave: 4, or 8 for fp16-stored tensors (half the bytes per row: twice the rows for the same bytes in flight)
__device__ __forceinline__ void head_out_bwd_body(const HeadOutBwdArgs& g) {
	const int head = blockIdx.y;
	const float* gout = head ? g.gout[1] : g.gout[0];
	if (gout == nullptr) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const float* y = head ? g.y[1] : g.y[0];
	const float* w = head ? g.w[1] : g.w[0];
	const float* z = head ? g.z[1] : g.z[0];
	float* dy = head ? g.dy[1] : g.dy[0];
	float* pw = head ? g.pw[1] : g.pw[0];
	float* pb = head ? g.pb[1] : g.pb[0];
	const float4 w0 = *reinterpret_cast<const float4*>(w + 0 * W + lane * 4);
	const float4 w1 = *reinterpret_cast<const float4*>(w + 1 * W + lane * 4);
	const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * W + lane * 4);
	const float scale = head ? 0.5f : 0.1f;
	float4 aw0 = make_float4(0, 0, 0, 0), aw1 = aw0, aw2 = aw0;
	float ab0 = 0.f, ab1 = 0.f, ab2 = 0.f;
	for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * U; r0 < g.rows; r0 += (int64_t)gridDim.x * 4 * U) {
		float4 yv[U];
		float d[U][3];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t row = r0 + u;
			yv[u] = (row < g.rows) ? ld4_any(y, row * W + lane * 4, g.half) : make_float4(0, 0, 0, 0);
		}
		// the U*3 output gradients of these rows are computed once (lanes 0..U*3-1: one tanh each) and broadcast
		float dmine = 0.f;
		if (lane < U * 3) {
			const int64_t e = r0 * 3 + lane;  // rows are contiguous: element (u, c) = r0*3 + u*3 + c
			if (e < g.rows * 3) {
				const float t = tanhf(z[e]);
				dmine = gout[e] * scale * (1.f - t * t);
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u)
#pragma unroll
			for (int c = 0; c < 3; ++c) d[u][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dmine), u * 3 + c));   // (bits, not the value: readlane converts a float argument)
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t row = r0 + u;
			if (row >= g.rows) break;
			const float d0 = d[u][0], d1 = d[u][1], d2 = d[u][2];
			float4 o;
			o.x = (yv[u].x > 0.f) ? d0 * w0.x + d1 * w1.x + d2 * w2.x : 0.f;
			o.y = (yv[u].y > 0.f) ? d0 * w0.y + d1 * w1.y + d2 * w2.y : 0.f;
			o.z = (yv[u].z > 0.f) ? d0 * w0.z + d1 * w1.z + d2 * w2.z : 0.f;
			o.w = (yv[u].w > 0.f) ? d0 * w0.w + d1 * w1.w + d2 * w2.w : 0.f;
			st4_any(dy, row * W + lane * 4, g.half, o);
			aw0.x += d0 * yv[u].x; aw0.y += d0 * yv[u].y; aw0.z += d0 * yv[u].z; aw0.w += d0 * yv[u].w;
			aw1.x += d1 * yv[u].x; aw1.y += d1 * yv[u].y; aw1.z += d1 * yv[u].z; aw1.w += d1 * yv[u].w;
			aw2.x += d2 * yv[u].x; aw2.y += d2 * yv[u].y; aw2.z += d2 * yv[u].z; aw2.w += d2 * yv[u].w;
			ab0 += d0; ab1 += d1; ab2 += d2;
		}
	}
	__shared__ __attribute__((aligned(16))) float red[4][3][256];
	__shared__ float redb[4][4];
	*reinterpret_cast<float4*>(&red[wave][0][lane * 4]) = aw0;
	*reinterpret_cast<float4*>(&red[wave][1][lane * 4]) = aw1;
	*reinterpret_cast<float4*>(&red[wave][2][lane * 4]) = aw2;
	if (lane == 0) { redb[wave][0] = ab0; redb[wave][1] = ab1; redb[wave][2] = ab2; }
	__syncthreads();
	const int k = threadIdx.x;
	for (int c = 0; c < 3; ++c)
		pw[((int64_t)blockIdx.x * 3 + c) * 256 + k] = red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k];
	if (k < 3) pb[(int64_t)blockIdx.x * 4 + k] = redb[0][k] + redb[1][k] + redb[2][k] + redb[3][k];
}
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const HeadOutBwdArgs g) {
	if (g.half) head_out_bwd_body<8>(g);
	else head_out_bwd_body<4>(g);
}

// grid (12, 2 heads) x 1024 threads: block handles 64 of the 768 dW outputs, the partial range cut 16 ways, LDS-combined.
struct HeadOutReduceArgs {
	const float* pw[2];
	const float* pb[2];
	float* dw[2];
	float* db[2];
	int nblk;
};
__global__ __launch_bounds__(1024) void head_out_reduce_kernel(const HeadOutReduceArgs g) {
	const int head = blockIdx.y;
	float* dw = head ? g.dw[1] : g.dw[0];
	if (dw == nullptr) return;
	const float* pw = head ? g.pw[1] : g.pw[0];
	const float* pb = head ? g.pb[1] : g.pb[0];
	float* db = head ? g.db[1] : g.db[0];
	__shared__ float red[16][64];
	__shared__ float redb[16][4];
	const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
	const int o = blockIdx.x * 64 + j;  // 0..767
	float s0 = 0.f, s1 = 0.f;
	int i = q;
	for (; i + 16 < g.nblk; i += 32) { s0 += pw[(int64_t)i * 768 + o]; s1 += pw[(int64_t)(i + 16) * 768 + o]; }
	if (i < g.nblk) s0 += pw[(int64_t)i * 768 + o];
	red[q][j] = s0 + s1;
	if (blockIdx.x == 0 && j < 3) {
		float t = 0.f;
		for (int k = q; k < g.nblk; k += 16) t += pb[(int64_t)k * 4 + j];
		redb[q][j] = t;
	}
	__syncthreads();
	if (q == 0) {
		float t = 0.f;
#pragma unroll
		for (int k = 0; k < 16; ++k) t += red[k][j];
		dw[o] = t;
		if (blockIdx.x == 0 && j < 3) {
			float u = 0.f;
#pragma unroll
			for (int k = 0; k < 16; ++k) u += redb[k][j];
			db[j] = u;
		}
	}
}

}  // namespace mlp
}  // namespace find
