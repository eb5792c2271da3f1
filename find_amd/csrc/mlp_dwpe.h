// dwpe_kernel: weight gradient of the Fourier (first trunk) layer,  dW0[n, k] = sum_rows dZ[row, n] * PE(pos[row])[k], without LDS and
// without ever materialising the features (round 3; replaces dw_kernel<AMODE_PE>, whose 64 KB of LDS staging left the matrix pipe 22 % busy).
// The data movement is dw4_kernel's (mlp_dw4.h): lane l of a k-pair (two rows) loads 16 bytes of dZ at columns 128 wn + 4 (l & 31).. of
// row 2t + (l >> 5); component ja feeds MFMA row-block ja, so accumulator (ja, jb) element (i, j) is output n = 128 wn + 4 i + ja.
// What differs is the B operand: column j of block jb is padded feature column  k = 256 kt + 64 wk + 32 jb + j  (pe_value's layout: 32-column
// chunks alternate sin / cos of the same 32 frequencies), so a lane owns ONE frequency for the whole run -- its three B entries sit in
// registers -- and one sin / cos evaluation per row gives both of its operands (block 0 = sin chunk, block 1 = cos chunk).  The chunk after
// the last sin / cos pair holds x, y, z and 29 zero columns: those three columns are plain FMAs in the two waves that also sum the bias
// gradient (a third k-tile of workgroups for them put the launch over one round of the chip: 288 workgroups at one per CU).
// Same split geometry (spf, cps in 32-row chunks, v_begin) as dw_kernel; slabs [split][256][Kp] with Kp = 2 pe + 32, reduce_w_kernel unchanged.
#pragma once
#include "mlp_kernels.h"

namespace find {
namespace mlp {

constexpr int DWPE_PD = 6;   // prefetch distance in k-pairs (ring of 8 register slots)

// sin(pi t), cos(pi t) without a branch: t = k / 2 + r with k = rint(2t), |r| <= 1/4 (both steps exact), the two minimax polynomials of the
// ROCm device library's sinpif / cospif on r, then the quadrant k mod 4 as a swap and two sign flips.  (The library's own sincospif is
// correct but compiles to several basic blocks, which tears the software pipeline below apart: the loads ended up next to their uses.)
__device__ __forceinline__ void sincospi_poly(const float t, float& s, float& c) {
	const float k = __builtin_rintf(t + t);
	const float r = __builtin_fmaf(k, -0.5f, t);
	const float r2 = r * r;
	float ps = __builtin_fmaf(r2, __uint_as_float(0x3e75aa41u), __uint_as_float(0xbf1f24beu));
	ps = __builtin_fmaf(r2, ps, __uint_as_float(0x40234736u));
	ps = __builtin_fmaf(r2, ps, __uint_as_float(0xc0a55e0eu));
	const float sp = __builtin_fmaf(__uint_as_float(0x40490fdbu), r, (r * r2) * ps);
	float pc = __builtin_fmaf(r2, __uint_as_float(0x3d4be544u), __uint_as_float(0x3e642e9du));
	pc = __builtin_fmaf(r2, pc, __uint_as_float(0xbfaad1dau));
	pc = __builtin_fmaf(r2, pc, __uint_as_float(0x4081e0d3u));
	pc = __builtin_fmaf(r2, pc, __uint_as_float(0xc09de9e6u));
	const float cp = __builtin_fmaf(r2, pc, 1.0f);
	const int q = (int)k;
	const bool odd = (q & 1) != 0;
	const float ss = odd ? cp : sp, cc = odd ? sp : cp;
	s = __uint_as_float(__float_as_uint(ss) ^ ((unsigned)(q & 2) << 30));          // quadrants 2, 3: sin < 0
	c = __uint_as_float(__float_as_uint(cc) ^ ((unsigned)((q + 1) & 2) << 30));    // quadrants 1, 2: cos < 0
}

template <bool XYZ>   // XYZ: the wave (k-tile 0, wk 0) that also owns the bias gradient and the x, y, z columns
__device__ __forceinline__ void dwpe_body(const DwArgs& g, const int kt, const int split, const int wn, const int wk, const int c0) {
	const int lane = threadIdx.x & 63;
	const int li = lane & 31, fh = lane >> 5;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int cpf = (g.V - g.v_begin + 31) / 32;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, cpf);
	const int r0 = g.v_begin + q0 * 32;
	const int nrows = max(min(g.v_begin + q1 * 32, g.V) - r0, 0);
	const int nkp = (nrows + 1) >> 1;
	const int nfull = (nrows >> 1) & ~7;   // k-pairs of the pipelined loop: whole groups of eight with both rows valid

	const int f = (c0 >> 1) * 32 + li;
	const float bx = g.Bm[f], by = g.Bm[g.pe + f], bz = g.Bm[2 * g.pe + f];
	auto feat = [&](const float x, const float y, const float z, float& b0, float& b1) {
		float t2 = fmaf(z, bz, fmaf(y, by, x * bx));   // pe_value's order of operations: the features of the forward pass
		t2 = 2.0f * t2;
		sincospi_poly(t2, b0, b1);
	};

	f32x16 acc[4][2];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
	typedef float v2f __attribute__((ext_vector_type(2)));
	v2f bs01 = {0.f, 0.f}, bs23 = {0.f, 0.f};           // column sums of the dZ values this lane loads
	float4 sx = make_float4(0.f, 0.f, 0.f, 0.f), sy = sx, sz = sx;   // ... and their products with x, y, z
	auto mfma8 = [&](const float4& a, const float b0, const float b1) {
#pragma unroll
		for (int ja = 0; ja < 4; ++ja) {
			const float av = ja == 0 ? a.x : (ja == 1 ? a.y : (ja == 2 ? a.z : a.w));
			acc[ja][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[ja][0], 0, 0, 0);
			acc[ja][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[ja][1], 0, 0, 0);
		}
	};
	auto sums = [&](const float4& a, const float x, const float y, const float z) {
		if constexpr (XYZ) {
			bs01 += v2f{a.x, a.y}; bs23 += v2f{a.z, a.w};
			sx.x = fmaf(a.x, x, sx.x); sx.y = fmaf(a.y, x, sx.y); sx.z = fmaf(a.z, x, sx.z); sx.w = fmaf(a.w, x, sx.w);
			sy.x = fmaf(a.x, y, sy.x); sy.y = fmaf(a.y, y, sy.y); sy.z = fmaf(a.z, y, sy.z); sy.w = fmaf(a.w, y, sy.w);
			sz.x = fmaf(a.x, z, sz.x); sz.y = fmaf(a.y, z, sz.y); sz.z = fmaf(a.z, z, sz.z); sz.w = fmaf(a.w, z, sz.w);
		}
	};

	const float* const zp = g.dz + (int64_t)foot * g.dz_foot_stride + ((int64_t)r0 + fh) * 256 + wn * 128 + 4 * li;
	const float* const pp = g.pos + (int64_t)foot * g.pos_foot_stride + ((int64_t)r0 + fh) * 3;

	if (nfull > 0) {
		float4 ra[8];
		float px[8], py[8], pz[8];
		const int last = nfull - 1;
		// k-pair t: rows 2t, 2t + 1 = 512 floats of dZ / 6 of pos further.  Loads past the end re-read the last k-pair (never used).
#pragma unroll
		for (int t = 0; t < DWPE_PD; ++t) {
			ra[t] = *reinterpret_cast<const float4*>(zp + (int64_t)t * 512);
			px[t] = pp[t * 6]; py[t] = pp[t * 6 + 1]; pz[t] = pp[t * 6 + 2];
			__builtin_amdgcn_sched_barrier(0);   // (in this order: issued out of order, the loop's first wait has to drain every load in flight -- on every iteration)
		}
		float b0, b1;
		feat(px[0], py[0], pz[0], b0, b1);
		for (int t0 = 0; t0 < nfull; t0 += 8) {
#pragma unroll
			for (int s = 0; s < 8; ++s) {
				{
					const int tt = min(t0 + s + DWPE_PD, last);
					ra[(s + DWPE_PD) & 7] = *reinterpret_cast<const float4*>(zp + (int64_t)tt * 512);
					const float* q = pp + tt * 6;
					px[(s + DWPE_PD) & 7] = q[0]; py[(s + DWPE_PD) & 7] = q[1]; pz[(s + DWPE_PD) & 7] = q[2];
				}
				__builtin_amdgcn_sched_barrier(0);
				// one scheduling region: this k-pair's eight MFMAs with the next k-pair's sin / cos between them
				const float4 a = ra[s];
				sums(a, px[s], py[s], pz[s]);
				mfma8(a, b0, b1);
				float n0, n1;
				feat(px[(s + 1) & 7], py[(s + 1) & 7], pz[(s + 1) & 7], n0, n1);
				b0 = n0; b1 = n1;
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
	// ---- the rest of the run (fewer than eight whole k-pairs and an odd last row): un-pipelined; a row past the end contributes zeros
	for (int t = nfull; t < nkp; ++t) {
		const bool ok = 2 * t + fh < nrows;
		const int off = ok ? 2 * t : 2 * t - 1;   // rows from this lane's first row (zp / pp point at row fh); the odd tail's missing row: the one before it, zeroed
		float4 a = *reinterpret_cast<const float4*>(zp + (int64_t)off * 256);
		const float* q = pp + off * 3;
		if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
		float b0, b1;
		feat(q[0], q[1], q[2], b0, b1);
		sums(a, q[0], q[1], q[2]);
		mfma8(a, b0, b1);
	}

	// ---- epilogue: accumulator (ja, jb) element (i, j): n = 128 wn + 4 i + ja, i = (r & 3) + 8 (r >> 2) + 4 fh; k = 256 kt + 64 wk + 32 jb + j
	float* const slab = g.pw + (int64_t)split * 256 * g.Kp;
	float* const pw = slab + kt * 256 + wk * 64 + li;
#pragma unroll
	for (int ja = 0; ja < 4; ++ja)
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const int n = wn * 128 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * fh) + ja;
			pw[(int64_t)n * g.Kp] = acc[ja][0][r];
			pw[(int64_t)n * g.Kp + 32] = acc[ja][1][r];
		}
	if constexpr (XYZ) {
		// the two row parities of a k-pair sit in the two halves of the wave; lane li of the lower half owns rows n = 128 wn + 4 li .. + 3
		auto fold = [&](float4& v) {
			v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
		};
		float4 bsum = make_float4(bs01.x, bs01.y, bs23.x, bs23.y);
		fold(bsum); fold(sx); fold(sy); fold(sz);
		if (fh == 0) {
			float* q = slab + (int64_t)(wn * 128 + 4 * li) * g.Kp + (g.pe >> 4) * 32;   // x, y, z: the first three columns of the last chunk
			q[0] = sx.x; q[1] = sy.x; q[2] = sz.x; q += g.Kp;
			q[0] = sx.y; q[1] = sy.y; q[2] = sz.y; q += g.Kp;
			q[0] = sx.z; q[1] = sy.z; q[2] = sz.z; q += g.Kp;
			q[0] = sx.w; q[1] = sy.w; q[2] = sz.w;
			if (g.pb != nullptr) *reinterpret_cast<float4*>(g.pb + (int64_t)split * 256 + wn * 128 + 4 * li) = bsum;
		}
	}
}

// grid (ceil(pe / 128), n_feet * spf): k-tile kt holds the sin / cos chunk pairs 4 kt .. 4 kt + 3, one per wk.  pe >= 32.
__global__ __launch_bounds__(512) void dwpe_kernel(const DwArgs g) {
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int wn = wave >> 2, wk = wave & 3;
	const int kt = blockIdx.x, split = blockIdx.y;
	const int c0 = kt * 8 + wk * 2;       // this wave's sin chunk; c0 + 1 is the cos chunk of the same frequencies
	if (c0 >= (g.pe >> 4)) return;        // past the last pair (no barrier in this kernel)
	if (c0 == 0) dwpe_body<true>(g, kt, split, wn, wk, c0);
	else dwpe_body<false>(g, kt, split, wn, wk, c0);
}

}  // namespace mlp
}  // namespace find
