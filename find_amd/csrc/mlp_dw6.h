// dw6_kernel: weight gradient  dW[n,k] = sum_rows dZ[row,n] * X[row,k]  in fp32-faithful arithmetic on the bf16 matrix pipe
// ("bf16x3", mlp_gemm6.h says why the six products a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) of the exact three-way bf16
// splits carry less error than one fp32 multiplication).  Both operands are activations here, so both are split on the way in.
// dw3's skeleton (the contraction runs over ROWS: the operands are transposed on their way into LDS), resized for six planes:
//   * one 4-wave workgroup owns the full 256 x 256 output (256 accumulator registers per lane) over a run of 16-row chunks of ONE
//     foot; partial tiles go to dw2's slabs and reduce_w_kernel, per-foot bias sums likewise (fp32 sums of the dZ values as loaded);
//   * staging: wave w stages operand (w >> 1), rows 8 (w & 1) .. +7 of the chunk: lane c loads the float4 at columns 4c .. 4c+3 of
//     each of its eight rows (a row is read by 64 lanes as one contiguous KB), splits every (row j, row j+1) pair of a column into
//     its three packed bf16 pairs (split_pair: 9 instructions per pair) and writes, per column and plane, the 8 rows as ONE 16-byte
//     LDS word.  LDS layout per plane and operand: [column'][16 rows] bf16 = 32 B per column, column' = column ^ ((column >> 2) & 7):
//     the 64 lanes of a fragment read still cover one contiguous KB, the 8 lanes of a write phase hit 8 different bank groups;
//   * an MFMA k-step is the whole 16-row chunk: lane (i, h) reads rows 8h .. 8h+7 of column 32 t + i of every plane with one
//     ds_read_b128; 16 blocks x 6 products = 96 MFMAs per wave and chunk (3072 matrix-pipe cycles) against ~150 VALU instructions
//     of splitting, 12 LDS writes and 24 LDS reads;
//   * two chunk buffers (2 x 48 KB): the chunk after the one being multiplied is loaded, split and stored under the MFMAs; one
//     barrier per chunk.
// Rows past the end of a foot are zero-filled at load time (no tail path): chunks_per_foot = ceil(V / 16).
// Where the time goes (round 5, same box, find_linear_wgrad at 16 x 6890 rows incl. the 20-us slab reduce): 101.5 us as is; without the
// split arithmetic 85.5; without the LDS plane writes 93.6; without the loads 89.7; without the barrier 97.4; without the fragment reads of
// the k loop 102.4; with all of that gone -- MFMAs, prologue and slab epilogue only -- 83.5.  A loop of nothing but these MFMAs runs at
// 16 - 17.6 us per 1000 (tools/mfma_bf16_mix.hip: the clock under bf16 MFMA load is 1.8 - 2.0 GHz, not the 2.4 the 2.5 PFLOP/s peak is
// quoted at), i.e. 44 us for the 2592 of a workgroup here: the kernel's loop is at ~0.7 of what the pipe sustains, 0.40 of the quoted peak.
#pragma once
#include "mlp_dw3.h"
#include "mlp_gemm6.h"

namespace find {
namespace mlp {

constexpr int DW6_PLANE = 256 * 32;             // one plane of one operand of one chunk: 256 columns x 16 rows bf16 = 8 KB
constexpr int DW6_OPER = 3 * DW6_PLANE;         // 24 KB
constexpr int DW6_BUF = 2 * DW6_OPER;           // dZ then X: 48 KB
constexpr int DW6_LDS = 2 * DW6_BUF + 4 * 256 * 4;  // double buffer + bias reduction scratch = 102 400 B

// VX ("bcast_fold", mlp.hip): the X operand is not stored -- the waves that stage it read the shared V x 256 product (g.x, foot stride 0) and
// form relu(product + xbias[foot]) in front of the split (one packed add and two max per row pair; branch-free: the waves that stage dZ
// add zero and clamp at -inf).
template <bool VX = false>
__device__ __forceinline__ void dw6_body(const Dw3Args& g, const int split) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* red = reinterpret_cast<float*>(smem + 2 * DW6_BUF);

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wn = wave >> 1, wk = wave & 1;
	const int li = lane & 31, fh = lane >> 5;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, g.chunks_per_foot);
	float* const pw = g.pw + (int64_t)split * 65536;
	float* const pb = g.pb ? g.pb + (int64_t)split * 256 : nullptr;
	// staging role of this wave: operand sop (0 = dZ, 1 = X), row group srg (rows 8 srg .. +7 of a chunk)
	const int sop = wave >> 1, srg = wave & 1;
	const float* const sfoot = sop == 0 ? g.dz + (int64_t)foot * g.dz_foot_stride : g.x + (int64_t)foot * g.x_foot_stride;

	f32x16 acc[4][4];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < 4; ++b)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
	float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
	const float bias_w = (g.pb != nullptr && (wave >> 1) == 0) ? 1.f : 0.f;   // only the waves that stage dZ sum bias gradients

	// Staging registers: two chunks in flight -- rows j of the wave's row group, columns 4 lane .. +3; a load is consumed two chunks later
	// (HBM latency under load is 1-2 us, a chunk takes ~1.4).  (Loads through inline asm with hand-counted waits gave wrong results here:
	// mlp_gemm7.h.)
	// Buffer loads: the descriptor's size is the foot's valid bytes, so rows past its end come back as zeros from the bounds check.
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	typedef int i4 __attribute__((ext_vector_type(4)));
	u4 st[2][8];
	const int c4 = lane * 4;
	float xb[4] = {0.f, 0.f, 0.f, 0.f};
	float xlo = -INFINITY;
	if constexpr (VX) {
		if (sop == 1) {
			const float4 b4 = *reinterpret_cast<const float4*>(g.xbias + (int64_t)foot * g.xbias_stride + c4);
			xb[0] = b4.x; xb[1] = b4.y; xb[2] = b4.z; xb[3] = b4.w;
			xlo = 0.f;
		}
	}
	auto virt = [&](f32x2 v, int e) -> f32x2 {
		if constexpr (VX) { v = v + f32x2{xb[e], xb[e]}; v = f32x2{fmaxf(v[0], xlo), fmaxf(v[1], xlo)}; }
		return v;
	};
	const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(sfoot)), 0, g.V * 1024, 0x00020000);
	auto load_chunk = [&](int q, u4 (&set)[8]) {
		const int r0 = q * 16 + 8 * srg;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			set[j] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, c4 * 4, (r0 + j) * 1024, 0);
		}
	};
#define FIND_DW6_WAIT(set, n) ((void)0)
	auto comp = [](const u4& v, int e) -> float { return __uint_as_float(e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w))); };
	// column 4 lane + e lands at column' = (4 lane + e) ^ (lane & 7)
	// the 16-byte row half of a column is swapped with the other one where bit 4 of the column is set (tools/lds_b128_probe2.hip: with the
	// halves in place the fragment reads cost 9.6 cycles and the writes 16, swapped 7.6 - 8.0 and 13.2; conflict-free reads are 7.2)
	const int wbase_h = sop * DW6_OPER + (srg ^ ((lane >> 2) & 1)) * 16;   // column 4 lane + e: bit 4 = bit 2 of the lane
	// one column (e) of the staged rows: split, write the three planes; the bias sums count a chunk once (keep = 0 for a repeated store)
	auto store_part = [&](char* buf, const u4 (&st)[8], int e, float keep) {
		u32x4 p1, p2, p3;
#pragma unroll
		for (int jj = 0; jj < 4; ++jj) {
			const Split2 s = split_pair(virt(f32x2{comp(st[2 * jj], e), comp(st[2 * jj + 1], e)}, e));
			p1[jj] = s.p1; p2[jj] = s.p2; p3[jj] = s.p3;
		}
		char* dst = buf + wbase_h + (((c4 + e) ^ (lane & 7)) * 32);
		*reinterpret_cast<u32x4*>(dst) = p1;
		*reinterpret_cast<u32x4*>(dst + DW6_PLANE) = p2;
		*reinterpret_cast<u32x4*>(dst + 2 * DW6_PLANE) = p3;
		{   // (branch-free: the waves that stage X add with keep = 0)
			float t = 0.f;
#pragma unroll
			for (int j = 0; j < 8; ++j) t += comp(st[j], e);
			if (e == 0) bsum.x += keep * t; else if (e == 1) bsum.y += keep * t; else if (e == 2) bsum.z += keep * t; else bsum.w += keep * t;
		}
	};
	// fragment addresses: dZ column wn*128 + 32 ti + li, X column wk*128 + 32 tj + li; (column >> 2) & 7 == (li >> 2) & 7 for all of them
	const int ci = (li ^ ((li >> 2) & 7)) * 32 + (fh ^ ((li >> 4) & 1)) * 16;   // (column 32 t + li: bit 4 = bit 4 of li)
	const int za = (wn * 128) * 32 + ci;
	const int xa = DW6_OPER + (wk * 128) * 32 + ci;
	auto frag = [&](const char* buf, int base, int t, int p) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(buf + base + p * DW6_PLANE + t * (32 * 32)); };

	if (q0 < q1) {
		load_chunk(q0, st[0]);
		load_chunk(min(q0 + 1, q1 - 1), st[1]);
		FIND_DW6_WAIT(st[0], 8);
#pragma unroll
		for (int e = 0; e < 4; ++e) store_part(smem, st[0], e, bias_w);
		load_chunk(min(q0 + 2, q1 - 1), st[0]);
		lds_barrier();
		int cb = 0;
		// One basic block per chunk -- no branch inside: a repeated store goes to the buffer nobody reads any more, a repeated load re-reads
		// the run's last chunk -- so that the instruction order can be prescribed: while the matrix pipe runs the 24 products of column
		// block tj, the wave splits and stores column e = tj of the NEXT chunk's rows (`set`: in registers for the last two chunks), then
		// refills `set` with the chunk three ahead.
		// The split of one column (e) of the wave's eight staged rows -- four row pairs, three pieces each -- cut into twelve stages of four
		// instructions (piece k of pair jj: round what is left to bf16, take it off): one stage goes behind every PAIR of MFMAs, fenced
		// by sched_barrier, so that the matrix pipe never waits for a run of VALU work.  (Round 4 left the placement to
		// sched_group_barrier hints; the compiler clustered 18-MFMA and 50-VALU runs and the pipe sat idle during the latter: 43 % busy.)
		struct ColSplit { f32x2 r[4]; u32x4 p[3]; float t; };
		auto cs_stage = [&](ColSplit& c, int jj, int k) {
			const unsigned qv = __builtin_bit_cast(unsigned, __builtin_convertvector(c.r[jj], bf16x2));
			c.p[k][jj] = qv;
			if (k < 2) c.r[jj] = c.r[jj] - f32x2{__uint_as_float(qv << 16), __uint_as_float(qv & 0xffff0000u)};
		};
		auto chunk_body = [&](int q, u4 (&set)[8]) {
			const char* buf = smem + cb * DW6_BUF;
			char* other = smem + (cb ^ 1) * DW6_BUF;   // its readers finished before the last barrier
			const float keep = (q + 1 < q1) ? bias_w : 0.f;
			bf16x8 a1[4], a2[4], a3[4], b1[2], b2[2], b3[2];
#pragma unroll
			for (int i = 0; i < 4; ++i) { a1[i] = frag(buf, za, i, 0); a2[i] = frag(buf, za, i, 1); a3[i] = frag(buf, za, i, 2); }
			b1[0] = frag(buf, xa, 0, 0); b2[0] = frag(buf, xa, 0, 1); b3[0] = frag(buf, xa, 0, 2);
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int tj = 0; tj < 4; ++tj) {
				const int cu = tj & 1, nx = cu ^ 1;
				ColSplit c;
#pragma unroll
				for (int jj = 0; jj < 4; ++jj) c.r[jj] = virt(f32x2{comp(set[2 * jj], tj), comp(set[2 * jj + 1], tj)}, tj);
				c.t = 0.f;
#pragma unroll
				for (int sl = 0; sl < 12; ++sl) {
					// MFMAs 2 sl, 2 sl + 1 of the 24: product group g (smallest terms first), row block ti
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int m = 2 * sl + h, gq = m >> 2, ti = m & 3;
						const bf16x8& av = (gq == 0) ? a3[ti] : ((gq == 2 || gq == 3) ? a2[ti] : a1[ti]);
						const bf16x8& bv = (gq == 0 || gq == 3 || gq == 5) ? b1[cu] : ((gq == 2 || gq == 4) ? b2[cu] : b3[cu]);
						acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[ti][tj], 0, 0, 0);
					}
					if (sl == 1 && tj + 1 < 4) { b1[nx] = frag(buf, xa, tj + 1, 0); b2[nx] = frag(buf, xa, tj + 1, 1); b3[nx] = frag(buf, xa, tj + 1, 2); }
					cs_stage(c, sl & 3, sl >> 2);
					if (sl < 8) c.t += comp(set[sl], tj);   // (the bias sums: the dZ values as loaded, rows in order)
					__builtin_amdgcn_sched_barrier(0);
				}
				{
					char* dst = other + wbase_h + (((c4 + tj) ^ (lane & 7)) * 32);
					*reinterpret_cast<u32x4*>(dst) = c.p[0];
					*reinterpret_cast<u32x4*>(dst + DW6_PLANE) = c.p[1];
					*reinterpret_cast<u32x4*>(dst + 2 * DW6_PLANE) = c.p[2];
					if (tj == 0) bsum.x += keep * c.t; else if (tj == 1) bsum.y += keep * c.t; else if (tj == 2) bsum.z += keep * c.t; else bsum.w += keep * c.t;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			load_chunk(min(q + 3, q1 - 1), set);   // three chunks ahead: consumed two iterations from now
			lds_barrier();   // (not __syncthreads: that would wait for the loads just issued)
			cb ^= 1;
		};
		for (int q = q0; q < q1; q += 2) {
			chunk_body(q, st[1]);
			if (q + 1 < q1) chunk_body(q + 1, st[0]);
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
		*reinterpret_cast<float4*>(&red[wave * 256 + lane * 4]) = bsum;   // (waves 2, 3 staged X: zeros)
		__syncthreads();
		pb[tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
	}
}

__global__ __launch_bounds__(256, 1) void dw6_kernel(const Dw3Args g) {
	FIND_CLAIM_WHOLE_REGISTER_FILE();   // 256 fp32 accumulators in AGPRs + ~150 VGPRs: more than 256 registers (see the macro)
	dw6_body(g, blockIdx.x);
}
__global__ __launch_bounds__(256, 1) void dw6v_kernel(const Dw3Args g) {   // (bcast_fold: X formed from the shared product)
	FIND_CLAIM_WHOLE_REGISTER_FILE();
	dw6_body<true>(g, blockIdx.x);
}

// Several weight gradients of the same geometry in ONE launch (blockIdx.y = job), as dw4_group_kernel: the 256 x 256 layers of a small call
// (the texture samples' seven, the shared trunk's four) in bf16x3 arithmetic -- 6/16 of dw4_group's matrix-pipe time, which is what counts
// for kernels that run BESIDE the step's other matrix-pipe work (round 6; fp32 dw4_group: `dw_lds_free` path of the other precisions).
constexpr int DW6_MAX_JOBS = 24;
struct Dw6Group { Dw3Args job[DW6_MAX_JOBS]; };
__global__ __launch_bounds__(256, 1) void dw6_group_kernel(const Dw6Group grp) {
	FIND_CLAIM_WHOLE_REGISTER_FILE();
	const int j = blockIdx.y;
	Dw3Args g;   // (fields copied one by one: a reference into the kernel-argument array makes the compiler copy the array to scratch)
	g.dz = grp.job[j].dz; g.dz_foot_stride = grp.job[j].dz_foot_stride; g.x = grp.job[j].x; g.x_foot_stride = grp.job[j].x_foot_stride;
	g.V = grp.job[j].V; g.chunks_per_foot = grp.job[j].chunks_per_foot; g.spf = grp.job[j].spf; g.cps = grp.job[j].cps;
	g.pw = grp.job[j].pw; g.pb = grp.job[j].pb;
	dw6_body(g, blockIdx.x);
}

}  // namespace mlp
}  // namespace find
