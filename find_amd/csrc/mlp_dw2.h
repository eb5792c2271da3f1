// dw2_kernel: weight gradient  dW[n,k] = sum_rows dZ[row,n] * X[row,k]  (256 x 256 output, fp32 MFMA) on the LDS-DMA
// pipeline.  One 4-wave workgroup owns a full 256x256 output tile over a contiguous range of 16-row chunks of ONE foot
// (so dZ and X are each read from HBM exactly once and per-foot bias sums fall out for free); the partial tile goes to a
// slab reduced by reduce_w_kernel (deterministic, no atomics).
// (Tried for the launches with few rows -- the shared trunk's V rows, the texture pass: 64x64 output blocks over 16-64 row splits
// with both MFMA operands loaded straight from global memory in operand layout, no LDS; 20 us against 25 us in isolation, but no
// faster inside the step, where these launches overlap the large ones: dropped.)
// ROUND 2: this kernel was the victim of the co-residence fault (mlp.hip): 312 registers per lane, and a wave above 256 registers gets
// wrong register contents when waves of another kernel share its SIMD.  dw2_kernel now claims the whole register file (nothing fits
// beside it: 0 wrong tensors in 600 stress passes) and is selectable with "dw_lds_free" = 0; the default is dw4_kernel (mlp_dw4.h: no
// LDS, <= 256 registers, 148 against 139-142 us per 110 240-row weight gradient including the slab reduce, the same step time);
// dw2_repro_kernel is the kernel as round 1 had it, the reproducer.
//
//   * chunk = 16 rows of dZ (16 KB) + 16 rows of X (16 KB), each row one 1-KB global_load_lds_dwordx4; 3-stage ring;
//   * MFMA operands come from LDS with ONE ds_read_b128 per operand per k-pair: lane l reads columns 4(l&31)..+3 of row
//     2t+(l>>5); component j feeds accumulator j, so accumulator (ja,jb) holds the outputs n = 4i+ja, k = 4j+jb
//     (i, j = MFMA row/col).  A 128x128 wave tile therefore needs 2 LDS reads per 16 MFMAs (the 32-row GEMM needs 5), and
//     the epilogue recombines jb = 0..3 into one coalesced 16-byte store;
//   * schedule as gemm3: early barrier before the last MFMA step of a chunk, DMA issue interleaved, hoisted arguments.
// Rows [16*floor(V/16), V) of every foot (at most 15) are one extra, un-pipelined chunk of the foot's last split.
#pragma once
#include "mlp_gemm3.h"

namespace find {
namespace mlp {

struct Dw2Args {
	const float* dz;         // rows (foot, v), ld 256
	int64_t dz_foot_stride;
	const float* x;          // rows (foot, v) or shared (x_foot_stride 0), ld 256
	int64_t x_foot_stride;
	int chunks_per_foot;     // floor(V / 16)
	int tail_rows;           // V % 16: leftover rows of every foot, folded in by the foot's last split
	int spf;                 // splits per foot (>= 1)
	int cps;                 // chunks per split
	float* pw;               // [n_feet*spf][256][256]
	float* pb;               // [n_feet*spf][256] or nullptr
	unsigned long long* dbg; // diagnosis only (tools/probe_lds_fault.py): every published ring stage is compared with its source in HBM
};

constexpr int DW2_STAGE = 2 * 16 * 1024;                // dZ rows then X rows
constexpr int DW2_LDS = 3 * DW2_STAGE + 4 * 256 * 4;    // + bias reduction scratch

#ifdef FIND_DIAG
// Diagnosis of the LDS co-residence fault (mlp.hip): thread t compares bytes [128 t, 128 t + 128) of a published stage (16 dZ rows then 16 X
// rows of 1 KB) with the same bytes read straight from HBM.  Log layout (uint64): [0] mismatching 16-byte pieces, [1] stages checked by
// thread 0, [2] stages checked by workgroups whose LDS base is not 0, then up to 64 records of 8 words:
// {split | chunk << 32, stage | tid << 8 | piece << 24, HW_REG_LDS_ALLOC, HW_REG_HW_ID, got.x bits | expected.x bits << 32,
//  1 if the piece equals what the stage held three chunks ago (stale) else 0, XCC id, 0}.
__device__ __noinline__ void dw2_verify_stage(unsigned long long* log, const char* stage, const float* zsrc, const float* xsrc, const float* zold,
											   const float* xold, int split, int chunk, int stage_idx) {
	const int tid = threadIdx.x;
	const int row = tid >> 3, part = tid & 7;   // 32 rows x 8 pieces of 128 B
	const float* src = row < 16 ? zsrc + row * 256 + part * 32 : xsrc + (row - 16) * 256 + part * 32;
	const float* old = row < 16 ? (zold ? zold + row * 256 + part * 32 : nullptr) : (xold ? xold + (row - 16) * 256 + part * 32 : nullptr);
	const unsigned alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);
	if (tid == 0) {
		atomicAdd(log + 1, 1ull);
		if (alloc & 0xFF) atomicAdd(log + 2, 1ull);
	}
#pragma unroll 1
	for (int i = 0; i < 8; ++i) {
		const float4 got = *reinterpret_cast<const float4*>(stage + row * 1024 + part * 128 + i * 16);
		const float4 want = *reinterpret_cast<const float4*>(src + i * 4);
		if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) {
			const unsigned long long n = atomicAdd(log, 1ull);
			if (n < 64) {
				unsigned long long* r = log + 8 + n * 8;
				int stale = 0;
				if (old) {
					const float4 o = *reinterpret_cast<const float4*>(old + i * 4);
					stale = (got.x == o.x && got.y == o.y && got.z == o.z && got.w == o.w) ? 1 : 0;
				}
				r[0] = (unsigned)split | ((unsigned long long)chunk << 32);
				r[1] = (unsigned)stage_idx | ((unsigned)tid << 8) | ((unsigned)i << 24);
				r[2] = alloc;
				r[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
				r[4] = __float_as_uint(got.x) | ((unsigned long long)__float_as_uint(want.x) << 32);
				r[5] = stale;
				r[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
				r[7] = 0;
			}
		}
	}
}
#endif

__device__ __forceinline__ void dw2_body(const Dw2Args& g, const int split) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const unsigned lds_base = (unsigned)(uintptr_t)smem;
	float* red = reinterpret_cast<float*>(smem + 3 * DW2_STAGE);

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wn = wave >> 1, wk = wave & 1;
	const int fh = lane >> 5;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, g.chunks_per_foot);
	const int total = max(q1 - q0, 0);
	const int tail = (sidx == g.spf - 1) ? g.tail_rows : 0;
	float* const pw = g.pw + (int64_t)split * 65536;
	float* const pb = g.pb ? g.pb + (int64_t)split * 256 : nullptr;
	const float* const zfoot = g.dz + (int64_t)foot * g.dz_foot_stride;
	const float* const xfoot = g.x + (int64_t)foot * g.x_foot_stride;

	f32x16 acc[4][4];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < 4; ++b)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
	float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

	// DMA: wave w moves rows 4w..4w+3 of both operands; lane i the 16 bytes at column 4i
	const unsigned d_dst = (wave * 4) * 1024;
	// fragment addresses: row 2t + fh, 16 B at column 4*(lane&31) of this wave's 128-column block
	const int za = fh * 1024 + (wn * 128 + 4 * (lane & 31)) * 4;
	const int xa = 16384 + fh * 1024 + (wk * 128 + 4 * (lane & 31)) * 4;
	const int ba = (wave * 4) * 1024 + lane * 16;  // bias partial: rows 4w..4w+3, column 4*lane
	float4 a0, b0, a1, b1;
	auto load_frag = [&](const char* sb, int t, float4& a, float4& b) {
		a = *reinterpret_cast<const float4*>(sb + za + t * 2048);
		b = *reinterpret_cast<const float4*>(sb + xa + t * 2048);
	};
	auto mfma_ja = [&](const float4& a, const float4& b, int ja) {
		const float av = ja == 0 ? a.x : (ja == 1 ? a.y : (ja == 2 ? a.z : a.w));
		acc[ja][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.x, acc[ja][0], 0, 0, 0);
		acc[ja][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.y, acc[ja][1], 0, 0, 0);
		acc[ja][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.z, acc[ja][2], 0, 0, 0);
		acc[ja][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.w, acc[ja][3], 0, 0, 0);
	};
	auto mfma_step = [&](const float4& a, const float4& b) { mfma_ja(a, b, 0); mfma_ja(a, b, 1); mfma_ja(a, b, 2); mfma_ja(a, b, 3); };
	auto bias_rows = [&](const char* sb) {  // column sums of dZ: 4 rows x 4 columns per thread per chunk
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float4 z = *reinterpret_cast<const float4*>(sb + ba + j * 1024);
			bsum.x += z.x; bsum.y += z.y; bsum.z += z.z; bsum.w += z.w;
		}
	};

	if (total > 0) {
		const float* zb = zfoot + (int64_t)q0 * 16 * 256;
		const float* xb = xfoot + (int64_t)q0 * 16 * 256;
		unsigned doff[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) doff[j] = (unsigned)(((wave * 4 + j) * 256 + lane * 4) * 4);
		int issued = 0, consumed = 0, ps = 0;
		const float* iz = nullptr;
		const float* ix = nullptr;
		unsigned idst = 0;
		auto issue_prepare = [&]() {
			iz = uniform_ptr(zb + (int64_t)issued * 4096);
			ix = uniform_ptr(xb + (int64_t)issued * 4096);
			idst = __builtin_amdgcn_readfirstlane(lds_base + ps * DW2_STAGE + d_dst);
		};
		auto issue_z = [&]() { dma4(iz, idst, doff[0], doff[1], doff[2], doff[3]); };
		auto issue_x = [&]() { dma4(ix, idst + 16384, doff[0], doff[1], doff[2], doff[3]); };
		auto issue_advance = [&]() { ++issued; ps = (ps == 2) ? 0 : ps + 1; };

		for (int k = 0; k < 3 && issued < total; ++k) { issue_prepare(); issue_z(); issue_x(); issue_advance(); }
		if (issued >= 3) FIND_WAIT_VMCNT(16);
		else if (issued == 2) FIND_WAIT_VMCNT(8);
		else FIND_WAIT_VMCNT(0);
		__builtin_amdgcn_s_barrier();

		load_frag(smem, 0, a0, b0);
		int cs = 0;
		for (int c = 0; c < total; ++c) {
			const char* sb = smem + cs * DW2_STAGE;
			load_frag(sb, 1, a1, b1); mfma_step(a0, b0);
			load_frag(sb, 2, a0, b0); mfma_step(a1, b1);
			load_frag(sb, 3, a1, b1); mfma_step(a0, b0);
			if (pb) bias_rows(sb);
			load_frag(sb, 4, a0, b0); mfma_step(a1, b1);
			load_frag(sb, 5, a1, b1); mfma_step(a0, b0);
			load_frag(sb, 6, a0, b0); mfma_step(a1, b1);
			load_frag(sb, 7, a1, b1);
			const bool more = consumed + 1 < total;
			const bool do_issue = more && issued < total;
			if (do_issue) issue_prepare();
			const int ns = (cs == 2) ? 0 : cs + 1;
			const char* nsb = smem + ns * DW2_STAGE;
			const int ahead = issued - (consumed + 2);
			mfma_step(a0, b0);
			// (diagnosis) the stage this wave has just consumed must still hold chunk c: stage index + 8 in the log
#ifdef FIND_DIAG
			if (g.dbg) dw2_verify_stage(g.dbg, sb, zb + (int64_t)c * 4096, xb + (int64_t)c * 4096, nullptr, nullptr, split, c, cs + 8);
#endif
			if (more) {
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				if (ahead <= 0) FIND_WAIT_VMCNT(0);
				else FIND_WAIT_VMCNT(8);
				__builtin_amdgcn_s_barrier();
#ifdef FIND_DIAG
				if (g.dbg) dw2_verify_stage(g.dbg, nsb, zb + (int64_t)(c + 1) * 4096, xb + (int64_t)(c + 1) * 4096, c >= 2 ? zb + (int64_t)(c - 2) * 4096 : nullptr,
											c >= 2 ? xb + (int64_t)(c - 2) * 4096 : nullptr, split, c + 1, ns);
#endif
			}
			__builtin_amdgcn_sched_barrier(0);
			mfma_ja(a1, b1, 0);
			__builtin_amdgcn_sched_barrier(0);
			load_frag(nsb, 0, a0, b0);  // stale stage at the very end; never used
			if (do_issue) issue_z();
			__builtin_amdgcn_sched_barrier(0);
			mfma_ja(a1, b1, 1);
			__builtin_amdgcn_sched_barrier(0);
			if (do_issue) issue_x();
			__builtin_amdgcn_sched_barrier(0);
			mfma_ja(a1, b1, 2);
			mfma_ja(a1, b1, 3);
			if (do_issue) issue_advance();
			cs = ns;
			++consumed;
		}
	}

	// ---- leftover rows [16*chunks_per_foot, V) of the foot (last split only): one un-pipelined chunk through stage 0.
	// Row addresses are clamped to the last valid row (finite data); the dZ rows past the end are then zeroed in LDS.
	if (tail > 0) {
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();  // every wave is done reading the ring
		unsigned toff[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) toff[j] = (unsigned)((min(wave * 4 + j, tail - 1) * 256 + lane * 4) * 4);
		const float* tz = uniform_ptr(zfoot + (int64_t)g.chunks_per_foot * 4096);
		const float* tx = uniform_ptr(xfoot + (int64_t)g.chunks_per_foot * 4096);
		const unsigned tdst = __builtin_amdgcn_readfirstlane(lds_base + d_dst);
		dma4(tz, tdst, toff[0], toff[1], toff[2], toff[3]);
		dma4(tx, tdst + 16384, toff[0], toff[1], toff[2], toff[3]);
		FIND_WAIT_VMCNT(0);
#pragma unroll
		for (int j = 0; j < 4; ++j)
			if (wave * 4 + j >= tail) *reinterpret_cast<float4*>(smem + d_dst + j * 1024 + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		if (pb) bias_rows(smem);
		const int steps = (tail + 1) >> 1;  // k-pairs that contain a valid row
		for (int t = 0; t < steps; ++t) {
			load_frag(smem, t, a0, b0);
			mfma_step(a0, b0);
		}
	}

	// ---- epilogue: accumulator (ja, jb) element (i, j) is output n = 4i + ja, k = 4j + jb (within the wave's 128x128 block)
	{
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(pw)), 0, 65536 * 4, 0x00020000);
		const int voff = ((wn * 128 + 16 * fh) * 256 + wk * 128 + 4 * (lane & 31)) * 4;
#pragma unroll
		for (int ja = 0; ja < 4; ++ja)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int nrow = 4 * ((r & 3) + 8 * (r >> 2)) + ja;
				typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
				u32x4 v;
				// (value copies: __builtin_bit_cast on a vector-element lvalue reads element 0)
				const float f0 = acc[ja][0][r], f1 = acc[ja][1][r], f2 = acc[ja][2][r], f3 = acc[ja][3][r];
				v.x = __float_as_uint(f0); v.y = __float_as_uint(f1); v.z = __float_as_uint(f2); v.w = __float_as_uint(f3);
				store_b128(v, rsrc, voff, nrow * 1024);
			}
	}
	if (pb) {
		__syncthreads();
		*reinterpret_cast<float4*>(&red[wave * 256 + lane * 4]) = bsum;
		__syncthreads();
		pb[tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
	}
}

__global__ __launch_bounds__(256, 1) void dw2_kernel(const Dw2Args g) {
	FIND_CLAIM_WHOLE_REGISTER_FILE();   // 312 registers by itself: the fault's victim (see the macro); with all 512, 0 wrong tensors in 450 stress passes
	dw2_body(g, blockIdx.x);
}
#ifdef FIND_DIAG
// The kernel as round 1 had it (312 registers, foreign waves fit beside it): the reproducer of the fault ("dw_lds_free" = 3), nothing else.
__global__ __launch_bounds__(256, 1) void dw2_repro_kernel(const Dw2Args g) { dw2_body(g, blockIdx.x); }
#endif

// Several weight gradients of the same geometry in ONE launch (blockIdx.y = job): the small calls -- batch 1, the texture samples --
// have eleven 256 x 256 weight gradients of 54 workgroups each; launched one by one they neither fill the chip nor overlap well.
constexpr int DW2_MAX_JOBS = 24;
struct Dw2Group { Dw2Args job[DW2_MAX_JOBS]; };
__global__ __launch_bounds__(256, 1) void dw2_group_kernel(const Dw2Group grp) {
	const int j = blockIdx.y;
	Dw2Args g;   // (fields copied one by one: a reference into the kernel-argument array makes the compiler copy the array to scratch)
	g.dz = grp.job[j].dz; g.dz_foot_stride = grp.job[j].dz_foot_stride; g.x = grp.job[j].x; g.x_foot_stride = grp.job[j].x_foot_stride;
	g.chunks_per_foot = grp.job[j].chunks_per_foot; g.tail_rows = grp.job[j].tail_rows; g.spf = grp.job[j].spf; g.cps = grp.job[j].cps;
	g.pw = grp.job[j].pw; g.pb = grp.job[j].pb; g.dbg = nullptr;
	FIND_CLAIM_WHOLE_REGISTER_FILE();
	dw2_body(g, blockIdx.x);
}

}  // namespace mlp
}  // namespace find
