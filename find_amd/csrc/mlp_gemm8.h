// gemm8_kernel: gemm7 (bf16x3 K = 256 Linear layer: weights resident in accumulation registers, activations split once per workgroup and
// staged through LDS -- mlp_gemm7.h) with HALF its LDS fragment traffic.
// What bounded gemm7 (round 6, tools/ablate_x3.py, 82 us at the headline shape): per 32-row unit its eight waves each read the whole
// activation tile from LDS (8 x 48 KB: 48 ds_read_b128 per wave) for 96 sixteen-cycle MFMAs per wave -- LDS pipe and matrix pipe both
// ~3000 cycles per unit: co-bounds (without the fragment reads 69 us, without split + plane writes 60, matrix pipe alone 50).  A fragment
// read fed ONE 16-column MFMA because a wave's registers hold W for 16 columns x all of K (96 registers).  Here the same 96 registers hold
// 32 columns x HALF of K:
//   * wave w = (column group cg = w & 3: 32 columns, k half kh = w >> 2: 128 of the 256 k).  v_mfma_f32_32x32x16_bf16, W the row operand
//     (32 columns x 16 k: 4 registers per plane and k-step, 8 k-steps x 3 planes = 96, pinned to AGPRs as in gemm7), the activation
//     fragment 32 rows x 16 k = ONE ds_read_b128 per lane: 24 fragment reads per wave and unit instead of 48, each feeding twice the
//     multiply-accumulates; 48 thirty-two-cycle MFMAs per wave and unit -- the same matrix-pipe time from half the instructions;
//   * the two k halves of a 32 x 32 block meet through LDS: the waves w and w ^ 4 each KEEP 16 of the block's 32 columns and GIVE the
//     other 16 away (8 accumulator registers: two ds_write_b128 behind the unit's last MFMA, two ds_read_b128 behind the next unit's
//     barrier -- 4 KB per wave and unit against the 24 KB of fragment reads saved).  Which 16 a wave keeps is arranged through the ORDER
//     of its W rows (MFMA row m holds column m ^ 16 kh), so that "kept" is always accumulator elements 0..7 and "given away" 8..15 and
//     the partner's element i + 8 of a lane is this wave's element i of the same lane: no runtime register indexing, no shuffles;
//   * everything else is gemm7's: whole-row loads two units ahead, the split of unit u + 1 cut into stages between the MFMAs of unit u,
//     one LDS-only barrier per unit, blocks stored under the next unit's MFMAs, 16-byte stores of four consecutive columns.
// Summation order: k-step by k-step inside a half (smallest terms first), then kept half + partner's half -- fp32 accumulation as before;
// the bf16x3 error statement of mlp_gemm6.h is unchanged (tests/test_gpu_mlp_bf16x3.py runs on whichever kernel the context selects).
#pragma once
#include "mlp_gemm7.h"

namespace find {
namespace mlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G8_XBUF = 8 * 2 * 64 * 16;        // one exchange buffer: 8 waves x two 16-byte words per lane = 16 KB
constexpr int GEMM8_LDS = 2 * G7_BUF + 2 * G8_XBUF;   // 137 216 B
constexpr int GEMM8_NW = 8;

// VAR (experiments): bit 0 = no s_setprio, bit 1 = two accumulator chains, bit 2 = no sched_barrier
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(GEMM8_NW * 64, 1) void gemm8_kernel(const Gemm2Args g) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	char* const xbuf = smem + 2 * G7_BUF;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int li = lane & 31, fh = lane >> 5;
	const int cg = wave & 3, kh = wave >> 2;
	const int b = blockIdx.x;
	const int npairs = gridDim.x / 2;
	const int pair = (b / 16) * 8 + (b & 7);      // the two column halves of the same rows are 8 blocks apart: same XCD
	const int col0 = ((b >> 3) & 1) * 128 + cg * 32;
	const int V = g.V, lda = g.lda, ldy = g.ldy, upf = g.tiles_per_foot;
	const int u0 = (int)((int64_t)pair * g.ntiles / npairs);
	const int u1 = (int)((int64_t)(pair + 1) * g.ntiles / npairs);
	if (u0 >= u1) return;

	// ---- prologue: W rows col0 + (li ^ 16 kh), k = 128 kh + 16 s + 8 fh .. + 8, as MFMA row operands
	bf16x8 B1[8], B2[8], B3[8];
	{
		const float4* wrow = reinterpret_cast<const float4*>(g.w0 + (int64_t)(col0 + (li ^ (16 * kh))) * g.ldw + 128 * kh + fh * 8);
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			split3(wrow[s * 4], wrow[s * 4 + 1], B1[s], B2[s], B3[s]);
			asm volatile("" : "+a"(B1[s]));
			asm volatile("" : "+a"(B2[s]));
			asm volatile("" : "+a"(B3[s]));
		}
	}

	// ---- staging (gemm7's): thread (wave w, lane l) loads the float4 at columns 4 l .. 4 l + 3 of rows 8 r + w, r = 0 .. 3, of a unit
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	typedef unsigned u2 __attribute__((ext_vector_type(2)));
	u4 st[2][4];
	auto make_srd = [&](const float* base, int nbytes) -> __amdgpu_buffer_rsrc_t {
		return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(base)), 0, nbytes, 0x00020000);
	};
	auto unit_rsrc = [&](int uu) -> __amdgpu_buffer_rsrc_t {
		// (a unit past the end of the range: size 0, every load comes back as zeros; so do the rows past the end of a foot)
		const int foot = uu / upf;
		const int v0 = (uu - foot * upf) * 32;
		const int valid = uu < u1 ? min(32, V - v0) : 0;
		return make_srd(g.a0 + (int64_t)foot * g.a_foot_stride + (int64_t)v0 * lda, valid * lda * 4);
	};
	const int lvoff = lane * 16;
	auto vm_load = [&](u4& dst, const __amdgpu_buffer_rsrc_t& srd, int voff, int soff) { dst = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0); };
	auto load_row = [&](const __amdgpu_buffer_rsrc_t& rs, u4 (&slot)[4], int r) { vm_load(slot[r], rs, lvoff, (8 * r + wave) * lda * 4); };
	struct RowSplit {
		f32x2 r[2];
		unsigned p[2][3];
		__device__ __forceinline__ void begin(const u4& v) {
			r[0] = f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
			r[1] = f32x2{__uint_as_float(v.z), __uint_as_float(v.w)};
		}
		__device__ __forceinline__ void stage(int h, int k) {
			p[h][k] = __builtin_bit_cast(unsigned, __builtin_convertvector(r[h], bf16x2));
			if (k < 2) r[h] = r[h] - f32x2{__uint_as_float(p[h][k] << 16), __uint_as_float(p[h][k] & 0xffff0000u)};
		}
	};
	const int wbase = wave * G7_ROW + lane * 8;
	auto write_plane = [&](char* buf, const RowSplit& q, int r, int k) {
		*reinterpret_cast<u2*>(buf + wbase + r * (8 * G7_ROW) + k * G7_PLANE) = u2{q.p[0][k], q.p[1][k]};
	};
	auto store_row = [&](char* buf, const u4 (&slot)[4], int r) {   // (the prologue's unit: all at once)
		RowSplit q;
		q.begin(slot[r]);
#pragma unroll
		for (int k = 0; k < 3; ++k) { q.stage(0, k); q.stage(1, k); }
#pragma unroll
		for (int k = 0; k < 3; ++k) write_plane(buf, q, r, k);
	};
	// MFMA column operand of k-step s, plane p: row li of the unit, 8 bf16 at k = 128 kh + 16 s + 8 fh
	const int abase = li * G7_ROW + kh * 256 + fh * 16;
	auto frag = [&](const char* buf, int p, int s) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(buf + abase + p * G7_PLANE + s * 32); };

	// this lane's kept columns: col0 + 16 kh + 8 j + 4 fh .. + 3, j = 0, 1 (accumulator elements 4 j .. 4 j + 3)
	const int kcol = col0 + 16 * kh + 4 * fh;
	u4 bnext[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
	auto load_bias = [&](int uu) {
		if constexpr (EPI == EPI_BIAS_RELU) {
			const int foot = min(uu, u1 - 1) / upf;
			const __amdgpu_buffer_rsrc_t bs = make_srd(g.bias + (int64_t)foot * g.bias_foot_stride, 256 * 4);
			vm_load(bnext[0], bs, kcol * 4, 0);
			vm_load(bnext[1], bs, (kcol + 8) * 4, 0);
		}
	};
	{
		load_bias(u0);
		const __amdgpu_buffer_rsrc_t r0 = unit_rsrc(u0);
#pragma unroll
		for (int r = 0; r < 4; ++r) load_row(r0, st[0], r);
		const __amdgpu_buffer_rsrc_t r1 = unit_rsrc(u0 + 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) load_row(r1, st[1], r);
#pragma unroll
		for (int r = 0; r < 4; ++r) store_row(smem, st[0], r);
		const __amdgpu_buffer_rsrc_t r2 = unit_rsrc(u0 + 2);
#pragma unroll
		for (int r = 0; r < 4; ++r) load_row(r2, st[0], r);
	}

	int cb = 0;
	f32x16 acc;
#pragma unroll
	for (int i = 0; i < 16; ++i) acc[i] = 0.f;
	f32x4 pend[2];   // the kept half of the previous unit's block: stored (plus the partner's half) under the current unit's MFMAs
	f32x4 xin[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};    // the partner's half of it
	auto init_acc = [&]() {
#pragma unroll
		for (int i = 0; i < 16; ++i) acc[i] = 0.f;
		if constexpr (EPI == EPI_BIAS_RELU) {
			acc[0] = __uint_as_float(bnext[0].x); acc[1] = __uint_as_float(bnext[0].y); acc[2] = __uint_as_float(bnext[0].z); acc[3] = __uint_as_float(bnext[0].w);
			acc[4] = __uint_as_float(bnext[1].x); acc[5] = __uint_as_float(bnext[1].y); acc[6] = __uint_as_float(bnext[1].z); acc[7] = __uint_as_float(bnext[1].w);
		}
	};
	auto keep = [&]() {
		pend[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
		pend[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
	};
	// exchange: this wave's given-away half goes to its own slot, the partner's comes from the partner's slot
	const int xw = wave * (2 * 64 * 16) + lane * 16, xr = (wave ^ 4) * (2 * 64 * 16) + lane * 16;
	auto give = [&](char* xb) {
		*reinterpret_cast<f32x4*>(xb + xw) = f32x4{acc[8], acc[9], acc[10], acc[11]};
		*reinterpret_cast<f32x4*>(xb + xw + 64 * 16) = f32x4{acc[12], acc[13], acc[14], acc[15]};
	};
	auto take = [&](const char* xb) {
		xin[0] = *reinterpret_cast<const f32x4*>(xb + xr);
		xin[1] = *reinterpret_cast<const f32x4*>(xb + xr + 64 * 16);
	};
	struct OutTile { __amdgpu_buffer_rsrc_t y, m; };
	auto out_tile = [&](int uu) -> OutTile {
		const int foot = uu / upf;
		const int v0 = (uu - foot * upf) * 32;
		const int nbytes = min(32, V - v0) * ldy * 4;
		OutTile t;
		t.y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(g.y + (int64_t)foot * g.y_foot_stride + (int64_t)v0 * ldy)), 0, nbytes, 0x00020000);
		t.m = t.y;
		if constexpr (EPI == EPI_MASK) t.m = make_srd(g.mask + (int64_t)foot * g.mask_foot_stride + (int64_t)v0 * ldy, nbytes);
		return t;
	};
	const int ovoff = (li * ldy + kcol) * 4;
	u4 mv[2];
	auto store_block = [&](const OutTile& t, int j) {
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			v[e] = pend[j][e] + xin[j][e];
			if constexpr (EPI == EPI_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
			if constexpr (EPI == EPI_MASK) v[e] = (__uint_as_float(mv[j][e]) > 0.f) ? v[e] : 0.f;
		}
		store_b128(u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, t.y, ovoff, j * 8 * 4);   // (no SGPR offset: common.h)
	};
	f32x16 acc2;
	auto mm = [&](const bf16x8& w, const bf16x8& x) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc, 0, 0, 0); };
	auto mm2 = [&](const bf16x8& w, const bf16x8& x) {
		if constexpr (VAR & 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc2, 0, 0, 0);
		else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc, 0, 0, 0);
	};
#define G8_SB() do { if constexpr (!(VAR & 4)) __builtin_amdgcn_sched_barrier(0); } while (0)

	// Unit u: multiply it (buffer cb).  Under its 48 MFMAs: split and store unit u + 1 from `slot` (one PAIR of values per k-step), refill
	// `slot` with unit u + 3, add the partner's half to the previous unit's block and store it (`first`: there is none).
	auto unit_body = [&](int u, u4 (&slot)[4], bool first) {
		lds_barrier();   // unit u is complete in buffer cb, the given-away halves of unit u - 1 in exchange buffer cb ^ 1; nobody reads plane buffer cb ^ 1 any more
		const char* buf = smem + cb * G7_BUF;
		char* other = smem + (cb ^ 1) * G7_BUF;
		const __amdgpu_buffer_rsrc_t rs2 = unit_rsrc(u + 3);
		const OutTile prev = out_tile(first ? u : u - 1);
		keep();
		if (!first) take(xbuf + (cb ^ 1) * G8_XBUF);
		init_acc();
		if constexpr (VAR & 2) {
#pragma unroll
			for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
		}
		load_bias(u + 1);
		if constexpr (EPI == EPI_MASK) {
			vm_load(mv[0], prev.m, ovoff, 0);
			vm_load(mv[1], prev.m, ovoff + 8 * 4, 0);
		}
		bf16x8 a1[2], a2[2], a3[2];
		a1[0] = frag(buf, 0, 0); a2[0] = frag(buf, 1, 0); a3[0] = frag(buf, 2, 0);
		RowSplit q;
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const int cu = s & 1, nx = cu ^ 1, row = s >> 1, h = s & 1;
			// the further a wave is into its unit, the lower its priority: the two waves of a SIMD then share the matrix pipe (gemm7)
			if (h == 0 && !(VAR & 1)) {
				if (row == 0) __builtin_amdgcn_s_setprio(3);
				else if (row == 1) __builtin_amdgcn_s_setprio(2);
				else if (row == 2) __builtin_amdgcn_s_setprio(1);
				else __builtin_amdgcn_s_setprio(0);
			}
			if (s + 1 < 8) { a1[nx] = frag(buf, 0, s + 1); a2[nx] = frag(buf, 1, s + 1); a3[nx] = frag(buf, 2, s + 1); }
			if (h == 0) q.begin(slot[row]);
			// smallest terms first
			mm(B1[s], a3[cu]);
			q.stage(h, 0);
			G8_SB();
			mm2(B3[s], a1[cu]);
			G8_SB();
			mm(B2[s], a2[cu]);
			q.stage(h, 1);
			G8_SB();
			mm2(B1[s], a2[cu]);
			if (h == 1 && !first && s >= 4) store_block(prev, (s - 4) >> 1);
			G8_SB();
			mm(B2[s], a1[cu]);
			q.stage(h, 2);
			G8_SB();
			mm2(B1[s], a1[cu]);
			if (h == 1) {
				write_plane(other, q, row, 0); write_plane(other, q, row, 1); write_plane(other, q, row, 2);
				load_row(rs2, slot, row);   // unit u + 3: on its way for two units
			}
			G8_SB();
		}
		if constexpr (VAR & 2) {
#pragma unroll
			for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
		}
		give(xbuf + cb * G8_XBUF);
		cb ^= 1;
	};
	unit_body(u0, st[1], true);
	for (int u = u0 + 1; u < u1; u += 2) {
		unit_body(u, st[0], false);
		if (u + 1 < u1) unit_body(u + 1, st[1], false);
	}
	{   // the last unit's block
		lds_barrier();
		const OutTile last = out_tile(u1 - 1);
		keep();
		take(xbuf + (cb ^ 1) * G8_XBUF);
		if constexpr (EPI == EPI_MASK) {
			vm_load(mv[0], last.m, ovoff, 0);
			vm_load(mv[1], last.m, ovoff + 8 * 4, 0);
		}
		store_block(last, 0); store_block(last, 1);
	}
}

#undef G8_SB
}  // namespace mlp
}  // namespace find
