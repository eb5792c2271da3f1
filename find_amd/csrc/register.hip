// Similarity registration  X = ((v + disp) * S) @ R + t,  R = Rx(e0) Ry(e1) Rz(e2)  (row-vector convention).
// Replaces euler_angles_to_matrix + Transform3d.scale().rotate().translate().transform_points in
// NeuralDisplacementField.get_meshes (reference src/model/model.py:481-491).  reg = [t(3), euler(3), S(3)].
#include "common.h"

namespace find {
namespace reg {

constexpr int VPB = 1024;  // vertices per block (256 threads x 4)

struct Rot {
	float m[9];
};

__device__ __forceinline__ void euler_xyz(const float* e, float* R) {
	float s0, c0, s1, c1, s2, c2;
	sincosf(e[0], &s0, &c0);
	sincosf(e[1], &s1, &c1);
	sincosf(e[2], &s2, &c2);
	// Rx @ Ry @ Rz
	R[0] = c1 * c2;                  R[1] = -c1 * s2;                 R[2] = s1;
	R[3] = c0 * s2 + s0 * s1 * c2;   R[4] = c0 * c2 - s0 * s1 * s2;   R[5] = -s0 * c1;
	R[6] = s0 * s2 - c0 * s1 * c2;   R[7] = s0 * c2 + c0 * s1 * s2;   R[8] = c0 * c1;
}

__global__ __launch_bounds__(256) void register_fwd_kernel(const float* __restrict__ verts, int64_t verts_foot_stride,
															const float* __restrict__ disp, const float* __restrict__ reg,
															int V, float* __restrict__ out) {
	__shared__ float R[9];
	const int foot = blockIdx.y;
	const float* rg = reg + (int64_t)foot * 9;
	if (threadIdx.x == 0) euler_xyz(rg + 3, R);
	__syncthreads();
	const float tx = rg[0], ty = rg[1], tz = rg[2], sx = rg[6], sy = rg[7], sz = rg[8];
	const float* vp = verts + (int64_t)foot * verts_foot_stride;
	const float* dp = disp + (int64_t)foot * V * 3;
	float* op = out + (int64_t)foot * V * 3;
	for (int i = 0; i < 4; ++i) {
		const int v = blockIdx.x * VPB + i * 256 + threadIdx.x;
		if (v >= V) break;
		const float qx = (vp[v * 3 + 0] + dp[v * 3 + 0]) * sx;
		const float qy = (vp[v * 3 + 1] + dp[v * 3 + 1]) * sy;
		const float qz = (vp[v * 3 + 2] + dp[v * 3 + 2]) * sz;
		op[v * 3 + 0] = qx * R[0] + qy * R[3] + qz * R[6] + tx;
		op[v * 3 + 1] = qx * R[1] + qy * R[4] + qz * R[7] + ty;
		op[v * 3 + 2] = qx * R[2] + qy * R[5] + qz * R[8] + tz;
	}
}

// partial[foot][blk][16]: 0-2 sum dX (dt); 3-5 sum (p+d)*dq (dS); 6-14 G_ij = sum q_i dX_j
__global__ __launch_bounds__(256) void register_bwd_kernel(const float* __restrict__ verts, int64_t verts_foot_stride,
															const float* __restrict__ disp, const float* __restrict__ reg,
															const float* __restrict__ dout, int V, float* __restrict__ d_disp,
															float* __restrict__ partial) {
	__shared__ float R[9];
	__shared__ float red[4][16];
	const int foot = blockIdx.y;
	const float* rg = reg + (int64_t)foot * 9;
	if (threadIdx.x == 0) euler_xyz(rg + 3, R);
	__syncthreads();
	const float sx = rg[6], sy = rg[7], sz = rg[8];
	const float* vp = verts + (int64_t)foot * verts_foot_stride;
	const float* dp = disp + (int64_t)foot * V * 3;
	const float* gp = dout + (int64_t)foot * V * 3;
	float* ddp = d_disp + (int64_t)foot * V * 3;
	float acc[15];
#pragma unroll
	for (int k = 0; k < 15; ++k) acc[k] = 0.f;
	for (int i = 0; i < 4; ++i) {
		const int v = blockIdx.x * VPB + i * 256 + threadIdx.x;
		if (v >= V) break;
		const float ux = vp[v * 3 + 0] + dp[v * 3 + 0], uy = vp[v * 3 + 1] + dp[v * 3 + 1], uz = vp[v * 3 + 2] + dp[v * 3 + 2];
		const float gx = gp[v * 3 + 0], gy = gp[v * 3 + 1], gz = gp[v * 3 + 2];
		const float dqx = gx * R[0] + gy * R[1] + gz * R[2];
		const float dqy = gx * R[3] + gy * R[4] + gz * R[5];
		const float dqz = gx * R[6] + gy * R[7] + gz * R[8];
		ddp[v * 3 + 0] = dqx * sx;
		ddp[v * 3 + 1] = dqy * sy;
		ddp[v * 3 + 2] = dqz * sz;
		acc[0] += gx; acc[1] += gy; acc[2] += gz;
		acc[3] += ux * dqx; acc[4] += uy * dqy; acc[5] += uz * dqz;
		const float qx = ux * sx, qy = uy * sy, qz = uz * sz;
		acc[6] += qx * gx; acc[7] += qx * gy; acc[8] += qx * gz;
		acc[9] += qy * gx; acc[10] += qy * gy; acc[11] += qy * gz;
		acc[12] += qz * gx; acc[13] += qz * gy; acc[14] += qz * gz;
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < 15; ++k) {
		const float s = wave_sum(acc[k]);
		if (lane == 0) red[wave][k] = s;
	}
	__syncthreads();
	if (threadIdx.x < 15) {
		const int k = threadIdx.x;
		partial[((int64_t)foot * gridDim.x + blockIdx.x) * 16 + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
	}
}

__global__ void register_finalize_kernel(const float* __restrict__ reg, const float* __restrict__ partial, int nblk,
										 float* __restrict__ d_reg) {
	const int foot = blockIdx.x;
	__shared__ float tot[16];
	if (threadIdx.x < 15) {
		float s = 0.f;
		for (int b = 0; b < nblk; ++b) s += partial[((int64_t)foot * nblk + b) * 16 + threadIdx.x];
		tot[threadIdx.x] = s;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const float* e = reg + (int64_t)foot * 9 + 3;
		float s0, c0, s1, c1, s2, c2;
		sincosf(e[0], &s0, &c0);
		sincosf(e[1], &s1, &c1);
		sincosf(e[2], &s2, &c2);
		const float* G = tot + 6;  // dL/dR_ij
		// dR/de0 = Rx' Ry Rz ; rows 1,2 of R rotate: d(row1) = -row2?  Written out explicitly:
		const float dR0[9] = {0.f, 0.f, 0.f,
							  -s0 * s2 + c0 * s1 * c2, -s0 * c2 - c0 * s1 * s2, -c0 * c1,
							  c0 * s2 + s0 * s1 * c2, c0 * c2 - s0 * s1 * s2, -s0 * c1};
		const float dR1[9] = {-s1 * c2, s1 * s2, c1,
							  s0 * c1 * c2, -s0 * c1 * s2, s0 * s1,
							  -c0 * c1 * c2, c0 * c1 * s2, -c0 * s1};
		const float dR2[9] = {-c1 * s2, -c1 * c2, 0.f,
							  c0 * c2 - s0 * s1 * s2, -c0 * s2 - s0 * s1 * c2, 0.f,
							  s0 * c2 + c0 * s1 * s2, -s0 * s2 + c0 * s1 * c2, 0.f};
		float de0 = 0.f, de1 = 0.f, de2 = 0.f;
		for (int k = 0; k < 9; ++k) {
			de0 += G[k] * dR0[k];
			de1 += G[k] * dR1[k];
			de2 += G[k] * dR2[k];
		}
		float* o = d_reg + (int64_t)foot * 9;
		o[0] = tot[0]; o[1] = tot[1]; o[2] = tot[2];
		o[3] = de0; o[4] = de1; o[5] = de2;
		o[6] = tot[3]; o[7] = tot[4]; o[8] = tot[5];
	}
}

}  // namespace reg
}  // namespace find

using namespace find;

extern "C" int find_register_fwd(const float* verts, int64_t verts_batch, const float* disp, const float* reg,
								 int64_t n_feet, int64_t n_pts, float* out, void* stream) {
	FIND_REQUIRE(verts && disp && reg && out, "find_register_fwd: NULL argument");
	FIND_REQUIRE(n_feet >= 1 && n_pts >= 1 && n_pts < (1ll << 29), "find_register_fwd: bad sizes");
	FIND_REQUIRE(verts_batch == 1 || verts_batch == n_feet, "find_register_fwd: verts_batch must be 1 or n_feet");
	hipStream_t s = reinterpret_cast<hipStream_t>(stream);
	dim3 grid((unsigned)cdiv(n_pts, reg::VPB), (unsigned)n_feet);
	hipLaunchKernelGGL(reg::register_fwd_kernel, grid, dim3(256), 0, s, verts, verts_batch == 1 ? 0 : n_pts * 3, disp, reg, (int)n_pts, out);
	FIND_LAUNCH_CHECK("register_fwd_kernel");
	return FIND_OK;
}

extern "C" int64_t find_register_bwd_ws_bytes(int64_t n_feet, int64_t n_pts) {
	return align_up(n_feet * cdiv(n_pts, reg::VPB) * 16 * (int64_t)sizeof(float), 256);
}

extern "C" int find_register_bwd(const float* verts, int64_t verts_batch, const float* disp, const float* reg,
								 const float* d_out, int64_t n_feet, int64_t n_pts, float* d_disp, float* d_reg,
								 void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(verts && disp && reg && d_out && d_disp && d_reg && ws, "find_register_bwd: NULL argument");
	FIND_REQUIRE(n_feet >= 1 && n_pts >= 1 && n_pts < (1ll << 29), "find_register_bwd: bad sizes");
	FIND_REQUIRE(verts_batch == 1 || verts_batch == n_feet, "find_register_bwd: verts_batch must be 1 or n_feet");
	if (ws_bytes < find_register_bwd_ws_bytes(n_feet, n_pts)) {
		set_error("find_register_bwd: workspace too small");
		return FIND_EWORKSPACE;
	}
	hipStream_t s = reinterpret_cast<hipStream_t>(stream);
	const int nblk = (int)cdiv(n_pts, reg::VPB);
	dim3 grid((unsigned)nblk, (unsigned)n_feet);
	hipLaunchKernelGGL(reg::register_bwd_kernel, grid, dim3(256), 0, s, verts, verts_batch == 1 ? 0 : n_pts * 3, disp, reg, d_out,
					   (int)n_pts, d_disp, reinterpret_cast<float*>(ws));
	hipLaunchKernelGGL(reg::register_finalize_kernel, dim3((unsigned)n_feet), dim3(64), 0, s, reg, reinterpret_cast<const float*>(ws), nblk, d_reg);
	FIND_LAUNCH_CHECK("register_bwd");
	return FIND_OK;
}
