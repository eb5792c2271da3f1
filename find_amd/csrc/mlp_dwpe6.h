// dwpe6_kernel: dwpe_kernel (mlp_dwpe.h: the Fourier layer's weight gradient dW0[n, k] = sum_rows dZ[row, n] * PE(pos[row])[k] without LDS,
// the feature operand GENERATED -- a lane owns one frequency for the whole run) in bf16x3 arithmetic (mlp_gemm6.h): both operands split
// exactly into three bf16 pieces in registers, six products per block on the bf16 matrix pipe, fp32 accumulation.
//   * the same workgroup / wave geometry, slab layout and epilogue as dwpe_kernel (wave (wn, wk): 128 output rows n x the sin and the cos chunk
//     of 32 frequencies; accumulator (ja, jb) element (i, j) is n = 128 wn + 4 i + ja, k = 256 kt + 64 wk + 32 jb + j), so reduce_w_kernel and
//     the launch code are unchanged;
//   * a k-step is 16 ROWS (v_mfma_f32_32x32x16_bf16): lane (li, fh) holds rows 8 fh .. 8 fh + 7 of the step in BOTH operands -- of dZ as eight
//     16-byte loads (columns 128 wn + 4 li .. + 3 of each row: component ja is the lane's value for row block ja, dwpe's trick), of the
//     features as eight sin / cos evaluations of ITS frequency; pairs of consecutive rows are split together (split_pair: 9 instructions
//     per pair); 48 MFMAs per step against ~440 VALU instructions: the kernel is VALU-bound at about 2.3 x the fp32 kernel's rate, and
//     takes 6/16 of its matrix-pipe time -- what counts beside the step's other matrix-pipe work;
//   * rows past the end of a run come back as zeros from the buffer loads' bounds check (no tail path): a zero dZ row contributes nothing
//     whatever its features are;
//   * single-buffered by ORDER: a step computes its feature planes first and refills the position registers for the next step at once,
//     then splits and multiplies the dZ blocks one row block at a time and refills the dZ registers behind the last split -- the next
//     step's ~300 instructions of sin / cos cover that latency; two waves per SIMD (<= 256 registers) cover each other's phases.
#pragma once
#include "mlp_dwpe.h"
#include "mlp_gemm6.h"

namespace find {
namespace mlp {

__device__ __forceinline__ void dwpe6_body(const DwArgs& g, const int kt, const int split, const int wn, const int wk, const int c0) {
	const int lane = threadIdx.x & 63;
	const int li = lane & 31, fh = lane >> 5;
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int cpf = (g.V - g.v_begin + 31) / 32;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, cpf);
	const int r0 = g.v_begin + q0 * 32;
	const int nrows = max(min(g.v_begin + q1 * 32, g.V) - r0, 0);
	const int nsteps = (nrows + 15) >> 4;

	const int f = (c0 >> 1) * 32 + li;
	const float bx = g.Bm[f], by = g.Bm[g.pe + f], bz = g.Bm[2 * g.pe + f];

	f32x16 acc[4][2];
#pragma unroll
	for (int a = 0; a < 4; ++a)
#pragma unroll
		for (int b = 0; b < 2; ++b)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

	// rows of the run through buffer descriptors: what lies past its end reads as zero
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<float*>(uniform_ptr(g.dz + (int64_t)foot * g.dz_foot_stride + (int64_t)r0 * 256)), 0, nrows * 1024, 0x00020000);
	const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<float*>(uniform_ptr(g.pos + (int64_t)foot * g.pos_foot_stride + (int64_t)r0 * 3)), 0, nrows * 12, 0x00020000);
	const int zvoff = fh * (8 * 1024) + (wn * 128 + 4 * li) * 4;
	const int pvoff = fh * (8 * 12);
	u4 ra[8];
	float ph[8];   // this step's feature phases 2 (x bx + y by + z bz) of the lane's eight rows (the positions themselves are gone by then)
	auto load_dz = [&](int s) {
#pragma unroll
		for (int j = 0; j < 8; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(zr, zvoff, (s * 16 + j) * 1024, 0);
	};
	struct Pos { float x[8], y[8], z[8]; };
	auto load_pos = [&](int s, Pos& p) {
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			p.x[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pr, pvoff, (s * 16 + j) * 12, 0));
			p.y[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pr, pvoff, (s * 16 + j) * 12 + 4, 0));
			p.z[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pr, pvoff, (s * 16 + j) * 12 + 8, 0));
		}
	};
	auto phases = [&](const Pos& p) {
#pragma unroll
		for (int j = 0; j < 8; ++j) ph[j] = 2.0f * fmaf(p.z[j], bz, fmaf(p.y[j], by, p.x[j] * bx));   // pe_value's order of operations: the features of the forward pass
	};
	auto comp = [](const u4& v, int e) -> float { return __uint_as_float(e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w))); };

	if (nsteps > 0) {
		Pos p0;
		load_pos(0, p0);
		load_dz(0);
		phases(p0);
	}
	for (int s = 0; s < nsteps; ++s) {
		Pos pn;
		load_pos(min(s + 1, nsteps - 1), pn);   // the next step's positions: in flight under this step's sin / cos (the last step re-reads its own)
		// ---- the feature planes of this step: rows 8 fh + j, this lane's frequency: sin -> block 0, cos -> block 1 (a row pair at a time:
		// evaluated, split, gone -- the register budget is 256 with 128 of them accumulators)
		bf16x8 b1[2], b2[2], b3[2];
		{
			u32x4 s1, s2, s3, c1, c2, c3;
#pragma unroll
			for (int jj = 0; jj < 4; ++jj) {
				float sv[2], cv[2];
				sincospi_poly(ph[2 * jj], sv[0], cv[0]);
				sincospi_poly(ph[2 * jj + 1], sv[1], cv[1]);
				const Split2 qs = split_pair(f32x2{sv[0], sv[1]}), qc = split_pair(f32x2{cv[0], cv[1]});
				s1[jj] = qs.p1; s2[jj] = qs.p2; s3[jj] = qs.p3; c1[jj] = qc.p1; c2[jj] = qc.p2; c3[jj] = qc.p3;
			}
			b1[0] = __builtin_bit_cast(bf16x8, s1); b2[0] = __builtin_bit_cast(bf16x8, s2); b3[0] = __builtin_bit_cast(bf16x8, s3);
			b1[1] = __builtin_bit_cast(bf16x8, c1); b2[1] = __builtin_bit_cast(bf16x8, c2); b3[1] = __builtin_bit_cast(bf16x8, c3);
		}
		phases(pn);   // (three registers per row become one)
		__builtin_amdgcn_sched_barrier(0);
		// ---- the dZ blocks, one row block (ja) at a time: split, then twelve MFMAs (smallest terms first)
#pragma unroll
		for (int ja = 0; ja < 4; ++ja) {
			u32x4 p1, p2, p3;
#pragma unroll
			for (int jj = 0; jj < 4; ++jj) { const Split2 q = split_pair(f32x2{comp(ra[2 * jj], ja), comp(ra[2 * jj + 1], ja)}); p1[jj] = q.p1; p2[jj] = q.p2; p3[jj] = q.p3; }
			const bf16x8 a1 = __builtin_bit_cast(bf16x8, p1), a2 = __builtin_bit_cast(bf16x8, p2), a3 = __builtin_bit_cast(bf16x8, p3);
			if (ja == 3) load_dz(min(s + 1, nsteps - 1));   // (the dZ registers are free from here on; the last step re-reads its own rows, unused)
#pragma unroll
			for (int jb = 0; jb < 2; ++jb) {
				acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[jb], acc[ja][jb], 0, 0, 0);
				acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[jb], acc[ja][jb], 0, 0, 0);
				acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[jb], acc[ja][jb], 0, 0, 0);
				acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[jb], acc[ja][jb], 0, 0, 0);
				acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[jb], acc[ja][jb], 0, 0, 0);
				acc[ja][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[jb], acc[ja][jb], 0, 0, 0);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
	}

	// ---- epilogue (dwpe_kernel's): accumulator (ja, jb) element (i, j): n = 128 wn + 4 i + ja, i = (r & 3) + 8 (r >> 2) + 4 fh; k = 256 kt + 64 wk + 32 jb + j
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
}

// The x, y, z columns of dW0 and the bias gradient: column n of dZ against (x, y, z, 1) over the rows of a run -- 4 of the 515 columns,
// plain fp32 FMAs; dwpe_kernel folds them into two of its waves, which dwpe6's register budget has no room for.  grid (n_feet * spf) x 1024
// threads (four row groups x 256 output rows n: a row of dZ is one contiguous KB per load); writes the same slab columns / bias rows as dwpe_kernel.
__global__ __launch_bounds__(1024) void dwxyz_kernel(const DwArgs g) {
	__shared__ float red[4][4][256];
	const int split = blockIdx.x, n = threadIdx.x & 255, grp = threadIdx.x >> 8;   // four row groups of 256 columns: rows grp, grp + 4, ...
	const int foot = split / g.spf;
	const int sidx = split - foot * g.spf;
	const int cpf = (g.V - g.v_begin + 31) / 32;
	const int q0 = sidx * g.cps;
	const int q1 = min(q0 + g.cps, cpf);
	const int r0 = g.v_begin + q0 * 32;
	const int nrows = max(min(g.v_begin + q1 * 32, g.V) - r0, 0);
	const float* zp = g.dz + (int64_t)foot * g.dz_foot_stride + (int64_t)r0 * 256 + n;
	const float* pp = g.pos + (int64_t)foot * g.pos_foot_stride + (int64_t)r0 * 3;
	float sx = 0.f, sy = 0.f, sz = 0.f, sb = 0.f;
	int r = grp;
	for (; r + 12 < nrows; r += 16) {   // four rows of this group per turn: their loads are independent
		float a[4], x[4], y[4], z[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int rr = r + 4 * u;
			a[u] = zp[(int64_t)rr * 256]; x[u] = pp[rr * 3]; y[u] = pp[rr * 3 + 1]; z[u] = pp[rr * 3 + 2];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) { sx = fmaf(a[u], x[u], sx); sy = fmaf(a[u], y[u], sy); sz = fmaf(a[u], z[u], sz); sb += a[u]; }
	}
	for (; r < nrows; r += 4) {
		const float a = zp[(int64_t)r * 256];
		sx = fmaf(a, pp[r * 3], sx); sy = fmaf(a, pp[r * 3 + 1], sy); sz = fmaf(a, pp[r * 3 + 2], sz); sb += a;
	}
	red[grp][0][n] = sx; red[grp][1][n] = sy; red[grp][2][n] = sz; red[grp][3][n] = sb;
	__syncthreads();
	if (grp == 0) {
		float* q = g.pw + (int64_t)split * 256 * g.Kp + (int64_t)n * g.Kp + (g.pe >> 4) * 32;   // x, y, z: the first three columns of the last chunk
#pragma unroll
		for (int c = 0; c < 3; ++c) q[c] = (red[0][c][n] + red[1][c][n]) + (red[2][c][n] + red[3][c][n]);
		if (g.pb != nullptr) g.pb[(int64_t)split * 256 + n] = (red[0][3][n] + red[1][3][n]) + (red[2][3][n] + red[3][3][n]);
	}
}

// grid (ceil(pe / 128), n_feet * spf) as dwpe_kernel.  pe >= 32.
__global__ __launch_bounds__(512) void dwpe6_kernel(const DwArgs g) {
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int wn = wave >> 2, wk = wave & 3;
	const int kt = blockIdx.x, split = blockIdx.y;
	const int c0 = kt * 8 + wk * 2;       // this wave's sin chunk; c0 + 1 is the cos chunk of the same frequencies
	if (c0 >= (g.pe >> 4)) return;        // past the last pair (no barrier in this kernel)
	dwpe6_body(g, kt, split, wn, wk, c0);
}

}  // namespace mlp
}  // namespace find
