// Geometry kernels of the FIND loss path (gfx950): face areas, surface-sample gather, brute-force K=1 nearest
// neighbour (Chamfer), mesh edge loss and cotangent-Laplacian smoothing.  HBM-bound / VALU-bound integer+float work:
// coalesced streams, LDS-staged target tiles, wave shuffles for reductions.  No MFMA (not GEMM-shaped).
//
// Replaces (reference call sites): pytorch3d.ops.sample_points_from_meshes (src/model/losses.py:39-41,63,67;
// src/eval/eval_3d.py:149-150), pytorch3d.ops.knn_points inside chamfer_distance (losses.py:77,85,88;
// eval_3d.py:151,159), pytorch3d.loss.mesh_edge_loss / mesh_laplacian_smoothing (losses.py:95-97).
#include "common.h"

namespace find {
namespace geom {

__device__ __forceinline__ float3 ld3(const float* p) { return make_float3(p[0], p[1], p[2]); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
	return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float norm3(float3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

// ------------------------------------------------------------------------------------------- face areas
__global__ void face_areas_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
								  int n_verts, int n_faces, float* __restrict__ areas) {
	const int m = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_faces) return;
	const int32_t* fp = faces + (int64_t)m * faces_mesh_stride + (int64_t)f * 3;
	const float* vp = verts + (int64_t)m * n_verts * 3;
	if (fp[0] < 0) {  // -1 padding of ragged batches: never sampled
		areas[(int64_t)m * n_faces + f] = 0.f;
		return;
	}
	const float3 a = ld3(vp + 3 * fp[0]), b = ld3(vp + 3 * fp[1]), c = ld3(vp + 3 * fp[2]);
	areas[(int64_t)m * n_faces + f] = 0.5f * norm3(cross3(sub3(b, a), sub3(c, a)));
}

// ------------------------------------------------------------------------------------------- surface sampling
__device__ __forceinline__ void bary_weights(const float* uv, float* w) {
	const float us = sqrtf(uv[0]);
	w[0] = 1.0f - us;
	w[1] = us * (1.0f - uv[1]);
	w[2] = us * uv[1];
}

__global__ void sample_fwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
								  const int32_t* __restrict__ face_idx, const float* __restrict__ uv, int n_verts, int n_samples,
								  float* __restrict__ out, const float* __restrict__ attr, float* __restrict__ attr_out) {
	const int m = blockIdx.y;
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_samples) return;
	const int64_t o = (int64_t)m * n_samples + s;
	const int32_t* fp = faces + (int64_t)m * faces_mesh_stride + (int64_t)face_idx[o] * 3;
	float w[3];
	bary_weights(uv + o * 2, w);
	const float* vp = verts + (int64_t)m * n_verts * 3;
	float3 p = make_float3(0.f, 0.f, 0.f), q = p;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int vi = fp[c];
		const float3 v = ld3(vp + 3 * vi);
		p.x += w[c] * v.x; p.y += w[c] * v.y; p.z += w[c] * v.z;
		if (attr) {
			const float3 a = ld3(attr + ((int64_t)m * n_verts + vi) * 3);
			q.x += w[c] * a.x; q.y += w[c] * a.y; q.z += w[c] * a.z;
		}
	}
	out[o * 3 + 0] = p.x; out[o * 3 + 1] = p.y; out[o * 3 + 2] = p.z;
	if (attr) { attr_out[o * 3 + 0] = q.x; attr_out[o * 3 + 1] = q.y; attr_out[o * 3 + 2] = q.z; }
}

__global__ void sample_bwd_kernel(const int32_t* __restrict__ faces, int64_t faces_mesh_stride, const int32_t* __restrict__ face_idx,
								  const float* __restrict__ uv, const float* __restrict__ d_out, int n_verts, int n_samples,
								  float* __restrict__ d_verts) {
	const int m = blockIdx.y;
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_samples) return;
	const int64_t o = (int64_t)m * n_samples + s;
	const int32_t* fp = faces + (int64_t)m * faces_mesh_stride + (int64_t)face_idx[o] * 3;
	float w[3];
	bary_weights(uv + o * 2, w);
	const float gx = d_out[o * 3 + 0], gy = d_out[o * 3 + 1], gz = d_out[o * 3 + 2];
	float* dv = d_verts + (int64_t)m * n_verts * 3;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const int vi = fp[c];
		atomicAdd(dv + 3 * vi + 0, w[c] * gx);
		atomicAdd(dv + 3 * vi + 1, w[c] * gy);
		atomicAdd(dv + 3 * vi + 2, w[c] * gz);
	}
}

// Sampler with the face choice on the device: pytorch3d.ops.sample_points_from_meshes draws faces ~ multinomial(area) with
// replacement; torch.multinomial normalises the weights, builds their running sum and searches it with a uniform draw -- five
// launches and a 13 776-element scan per mesh.  Here: face_areas_kernel (the vertex gathers spread over the whole chip), one block per
// mesh turns the areas into their running sum in place (rounds of 16 384 faces, sixteen consecutive ones per thread, block scan of the
// 1024 per-thread totals), then one thread per sample searches it with its own uniform draw r in [0,1): the first face whose running
// sum exceeds r * total.  Faces of zero area (the -1 padding of ragged batches included) can never be that first face.
// (Measured and dropped: areas computed inside the scan block -- one launch less, but a mesh's 13 776 x 12 scattered loads then go
// through ONE CU's L1: 44 - 67 us against 5 + 12.)
constexpr int CDF_PER = 16;
__global__ __launch_bounds__(1024) void area_scan_kernel(int n_faces, float* __restrict__ cdf /* in: areas, out: their running sum */) {
	__shared__ float wsum[16];
	float* out = cdf + (int64_t)blockIdx.x * n_faces;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const bool vec = (n_faces & 3) == 0;   // rows of the (n_meshes, n_faces) buffer stay 16-byte aligned
	float carry = 0.f;   // running sum of the faces before this round (the same value in every thread)
	// a fixed order -- consecutive faces inside a thread, shuffles inside the wave, LDS across the 16 waves --, so the result is deterministic
	for (int f0 = 0; f0 < n_faces; f0 += 1024 * CDF_PER) {
		const int f = f0 + CDF_PER * (int)threadIdx.x;
		float a[CDF_PER];
		if (vec && f + CDF_PER <= n_faces) {
#pragma unroll
			for (int k = 0; k < CDF_PER; k += 4) {
				const float4 t = *reinterpret_cast<const float4*>(out + f + k);
				a[k] = t.x; a[k + 1] = t.y; a[k + 2] = t.z; a[k + 3] = t.w;
			}
		} else {
#pragma unroll
			for (int k = 0; k < CDF_PER; ++k) a[k] = f + k < n_faces ? out[f + k] : 0.f;
		}
#pragma unroll
		for (int k = 1; k < CDF_PER; ++k) a[k] += a[k - 1];
		float inc = a[CDF_PER - 1];
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const float t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		__syncthreads();   // (the previous round's wsum has been read)
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		float base = carry, total = carry;
		for (int w = 0; w < 16; ++w) {
			if (w < wave) base += wsum[w];
			total += wsum[w];
		}
		const float off = base + (inc - a[CDF_PER - 1]);
		if (vec && f + CDF_PER <= n_faces) {
#pragma unroll
			for (int k = 0; k < CDF_PER; k += 4)
				*reinterpret_cast<float4*>(out + f + k) = make_float4(off + a[k], off + a[k + 1], off + a[k + 2], off + a[k + 3]);
		} else {
#pragma unroll
			for (int k = 0; k < CDF_PER; ++k)
				if (f + k < n_faces) out[f + k] = off + a[k];
		}
		carry = total;
	}
}

__global__ void sample_surface_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
									  const float* __restrict__ cdf, const float* __restrict__ rnd /* (n, s, 3): face draw, u, v */, int n_verts,
									  int n_faces, int n_samples, int32_t* __restrict__ face_idx, float* __restrict__ uv, float* __restrict__ out,
									  const float* __restrict__ attr, float* __restrict__ attr_out) {
	const int m = blockIdx.y;
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_samples) return;
	const int64_t o = (int64_t)m * n_samples + s;
	const float* c = cdf + (int64_t)m * n_faces;
	const float total = c[n_faces - 1];
	// r < total always: a draw that rounds up to the total would find no face
	const float r = fminf(rnd[o * 3] * total, __uint_as_float(__float_as_uint(total) - 1u));
	int lo = 0, hi = n_faces - 1;   // first f with c[f] > r; c[n_faces - 1] = total > r
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (c[mid] > r) hi = mid; else lo = mid + 1;
	}
	const int f = total > 0.f ? lo : 0;
	face_idx[o] = f;
	const float u2[2] = {rnd[o * 3 + 1], rnd[o * 3 + 2]};
	uv[o * 2] = u2[0]; uv[o * 2 + 1] = u2[1];
	const int32_t* fp = faces + (int64_t)m * faces_mesh_stride + (int64_t)f * 3;
	float w[3];
	bary_weights(u2, w);
	const float* vp = verts + (int64_t)m * n_verts * 3;
	float3 p = make_float3(0.f, 0.f, 0.f), q = p;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const int vi = max(fp[k], 0);
		const float3 v = ld3(vp + 3 * vi);
		p.x += w[k] * v.x; p.y += w[k] * v.y; p.z += w[k] * v.z;
		if (attr) {
			const float3 t = ld3(attr + ((int64_t)m * n_verts + vi) * 3);
			q.x += w[k] * t.x; q.y += w[k] * t.y; q.z += w[k] * t.z;
		}
	}
	out[o * 3 + 0] = p.x; out[o * 3 + 1] = p.y; out[o * 3 + 2] = p.z;
	if (attr) { attr_out[o * 3 + 0] = q.x; attr_out[o * 3 + 1] = q.y; attr_out[o * 3 + 2] = q.z; }
}

// Masked mean squared error of the texture loss (reference losses.py:43-57): points whose target colour is saturated in every channel
// (no channel < 1) are left out of the sum, the mean is over ALL n * 3 elements.  One workgroup, deterministic.
__global__ __launch_bounds__(1024) void masked_mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n_pts,
															   float* __restrict__ loss) {
	__shared__ float red[16];
	float s = 0.f;
	// A thread takes FOUR consecutive points at a time -- 12 floats = three 16-byte loads per tensor, fully coalesced -- two such groups in
	// flight.  (Round 5 read a point as three dwords 12 bytes apart: 96 load instructions per thread, each touching a dozen cache lines, all
	// through ONE compute unit's address pipeline: 15 us for the texture pass's 16 000 points, on the step's critical path between the forward
	// and the backward.)  The points past the last whole group are the first threads', one each.
	auto point = [&](float t0, float t1, float t2, float q0, float q1, float q2) {
		if (t0 < 1.f || t1 < 1.f || t2 < 1.f) {
			const float a = q0 - t0, b = q1 - t1, c = q2 - t2;
			s += a * a + b * b + c * c;
		}
	};
	const int64_t n_grp = n_pts >> 2;
	const bool vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15) == 0;
	if (vec) {
		constexpr int U = 2;
		for (int64_t g0 = threadIdx.x; g0 < n_grp; g0 += 1024 * U) {
			float4 t[U][3], q[U][3];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int64_t g = min(g0 + (int64_t)u * 1024, n_grp - 1);
#pragma unroll
				for (int k = 0; k < 3; ++k) { t[u][k] = reinterpret_cast<const float4*>(target)[g * 3 + k]; q[u][k] = reinterpret_cast<const float4*>(pred)[g * 3 + k]; }
			}
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (g0 + (int64_t)u * 1024 >= n_grp) continue;
				point(t[u][0].x, t[u][0].y, t[u][0].z, q[u][0].x, q[u][0].y, q[u][0].z);
				point(t[u][0].w, t[u][1].x, t[u][1].y, q[u][0].w, q[u][1].x, q[u][1].y);
				point(t[u][1].z, t[u][1].w, t[u][2].x, q[u][1].z, q[u][1].w, q[u][2].x);
				point(t[u][2].y, t[u][2].z, t[u][2].w, q[u][2].y, q[u][2].z, q[u][2].w);
			}
		}
	}
	for (int64_t i = (vec ? n_grp * 4 : 0) + threadIdx.x; i < n_pts; i += 1024)
		point(target[i * 3], target[i * 3 + 1], target[i * 3 + 2], pred[i * 3], pred[i * 3 + 1], pred[i * 3 + 2]);
	s = wave_sum(s);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		float t = 0.f;
		for (int w = 0; w < 16; ++w) t += red[w];
		*loss = t / (float)(n_pts * 3);
	}
}

__global__ void masked_mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n_pts, const float* __restrict__ g_loss,
									  float* __restrict__ d_pred) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pts) return;
	const float t0 = target[i * 3], t1 = target[i * 3 + 1], t2 = target[i * 3 + 2];
	const float g = (t0 < 1.f || t1 < 1.f || t2 < 1.f) ? 2.0f * (*g_loss) / (float)(n_pts * 3) : 0.f;
	d_pred[i * 3] = g * (pred[i * 3] - t0); d_pred[i * 3 + 1] = g * (pred[i * 3 + 1] - t1); d_pred[i * 3 + 2] = g * (pred[i * 3 + 2] - t2);
}

// ------------------------------------------------------------------------------------------- image losses
// mean( (a * am - b * bm)^2 ) over n_pix x C values: FIND's pixel loss compares the rendered images INSIDE their silhouettes (a, b (.., C)
// images, am, bm (..) masks: model.py:1101-1105) and its silhouette loss the two masks (C = 1, no am / bm: model.py:1107-1108, losses.py:122-128).
// As torch expressions these were five elementwise passes + a reduction forward and as many backward over 12.6 M-value tensors at 512^2
// (~0.1 ms each, 1.7 ms of the C4 step between the render and its backward); here one pass each way.  Deterministic: fixed block ->
// element mapping, per-block partial sums, one wave adds the partials in index order.
constexpr int IMSE_BLOCKS = 1024;
// A thread takes FOUR consecutive pixels when it can (n_pix a multiple of 4, C = 3 or 1: 16-byte loads of a, b and of the masks); the
// scalar version moved 134 MB in 137 us at 512^2.
template <int C, bool VEC>
__global__ __launch_bounds__(1024) void image_mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ am, const float* __restrict__ b,
															  const float* __restrict__ bm, int64_t n_pix, int Cdyn, float* __restrict__ partial) {
	__shared__ float red[16];
	float s = 0.f;
	if constexpr (VEC) {
		const int64_t nq = n_pix >> 2;
		for (int64_t q = (int64_t)blockIdx.x * 1024 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 1024) {
			float va[4 * C], vb[4 * C], ma[4], mb[4];
#pragma unroll
			for (int k = 0; k < C; ++k) {
				const float4 x = reinterpret_cast<const float4*>(a)[q * C + k], y = reinterpret_cast<const float4*>(b)[q * C + k];
				va[4 * k] = x.x; va[4 * k + 1] = x.y; va[4 * k + 2] = x.z; va[4 * k + 3] = x.w;
				vb[4 * k] = y.x; vb[4 * k + 1] = y.y; vb[4 * k + 2] = y.z; vb[4 * k + 3] = y.w;
			}
			const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
			const float4 x = am ? reinterpret_cast<const float4*>(am)[q] : one, y = bm ? reinterpret_cast<const float4*>(bm)[q] : one;
			ma[0] = x.x; ma[1] = x.y; ma[2] = x.z; ma[3] = x.w; mb[0] = y.x; mb[1] = y.y; mb[2] = y.z; mb[3] = y.w;
#pragma unroll
			for (int i = 0; i < 4 * C; ++i) {
				const float d = va[i] * ma[i / C] - vb[i] * mb[i / C];
				s += d * d;
			}
		}
	} else {
		for (int64_t p = (int64_t)blockIdx.x * 1024 + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * 1024) {
			const float ma = am ? am[p] : 1.f, mb = bm ? bm[p] : 1.f;
			for (int c = 0; c < Cdyn; ++c) {
				const float d = a[p * Cdyn + c] * ma - b[p * Cdyn + c] * mb;
				s += d * d;
			}
		}
	}
	s = wave_sum(s);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		float t = 0.f;
		for (int w = 0; w < 16; ++w) t += red[w];
		partial[blockIdx.x] = t;
	}
}

__global__ __launch_bounds__(64) void image_mse_finalize_kernel(const float* __restrict__ partial, int nblk, float scale, float* __restrict__ loss) {
	float t = 0.f;
	for (int k = threadIdx.x; k < nblk; k += 64) t += partial[k];
	t = wave_sum(t);
	if (threadIdx.x == 0) *loss = t * scale;
}

// d_a = g * 2 / (n_pix C) * (a am - b bm) * am;   d_am = g * 2 / (n_pix C) * sum_c (a am - b bm) * a
template <int C, bool VEC>
__global__ __launch_bounds__(256) void image_mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ am, const float* __restrict__ b,
															 const float* __restrict__ bm, int64_t n_pix, int Cdyn, const float* __restrict__ g_loss,
															 float* __restrict__ d_a, float* __restrict__ d_am) {
	const float s = 2.0f * (*g_loss) / ((float)n_pix * (float)(VEC ? C : Cdyn));
	if constexpr (VEC) {
		const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
		if (q >= (n_pix >> 2)) return;
		float va[4 * C], vb[4 * C], ma[4], mb[4], da[4 * C], dm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int k = 0; k < C; ++k) {
			const float4 x = reinterpret_cast<const float4*>(a)[q * C + k], y = reinterpret_cast<const float4*>(b)[q * C + k];
			va[4 * k] = x.x; va[4 * k + 1] = x.y; va[4 * k + 2] = x.z; va[4 * k + 3] = x.w;
			vb[4 * k] = y.x; vb[4 * k + 1] = y.y; vb[4 * k + 2] = y.z; vb[4 * k + 3] = y.w;
		}
		const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
		const float4 x = am ? reinterpret_cast<const float4*>(am)[q] : one, y = bm ? reinterpret_cast<const float4*>(bm)[q] : one;
		ma[0] = x.x; ma[1] = x.y; ma[2] = x.z; ma[3] = x.w; mb[0] = y.x; mb[1] = y.y; mb[2] = y.z; mb[3] = y.w;
#pragma unroll
		for (int i = 0; i < 4 * C; ++i) {
			const float d = s * (va[i] * ma[i / C] - vb[i] * mb[i / C]);
			da[i] = d * ma[i / C];
			dm[i / C] += d * va[i];
		}
		if (d_a) {
#pragma unroll
			for (int k = 0; k < C; ++k) reinterpret_cast<float4*>(d_a)[q * C + k] = make_float4(da[4 * k], da[4 * k + 1], da[4 * k + 2], da[4 * k + 3]);
		}
		if (d_am) reinterpret_cast<float4*>(d_am)[q] = make_float4(dm[0], dm[1], dm[2], dm[3]);
	} else {
		const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
		if (p >= n_pix) return;
		const float ma = am ? am[p] : 1.f, mb = bm ? bm[p] : 1.f;
		float dm = 0.f;
		for (int c = 0; c < Cdyn; ++c) {
			const float av = a[p * Cdyn + c];
			const float d = s * (av * ma - b[p * Cdyn + c] * mb);
			if (d_a) d_a[p * Cdyn + c] = d * ma;
			dm += d * av;
		}
		if (d_am) d_am[p] = dm;
	}
}

// ------------------------------------------------------------------------------------------- nearest neighbour
// A lane owns 2*NP query points, held as NP packed pairs (v_pk_add/mul/fma_f32: two queries per VALU instruction); the four waves
// of a block own the SAME 128*NP queries and each scans one quarter of every target tile (targets stream through LDS in tiles of
// NN_TILE, x,y,z,pad float4; a wave-uniform ds_read broadcasts one target to all lanes), so that FIND's small clouds (5-10 k points
// x 1-16 feet) still put several waves on every SIMD.  Targets are taken four at a time: the minimum of the four distances is
// compared with the running best (strict '<': the EARLIEST group of four wins a tie) and only the group's first index is kept --
// 4.9 VALU instructions per query-target pair instead of 9 for a compare-and-select per pair.  After the scan each query recomputes
// the four distances of its winning group with the same three roundings (mul, fma, fma) and takes the first that equals the best:
// lowest index on ties, as knn_points.  The four partial results -- and, when the target range of a cloud is split over `splits`
// blocks to fill the chip at batch 1, the partial results of the blocks -- are merged as 64-bit keys (distance bits << 32 | index),
// whose unsigned order IS (distance, index) for the non-negative distances here.
constexpr int NN_TILE = 1024;
typedef float v2f __attribute__((ext_vector_type(2)));

struct NnDir {
	const float* x;          // queries (n, p1_max, 3)
	const int32_t* x_len;    // (n) or NULL
	const float* y;          // targets (n, p2_max, 3)
	const int32_t* y_len;
	int p1_max, p2_max;
	float* dist;             // (n, p1_max) or NULL   \ direct outputs (splits == 1 only)
	int32_t* idx;            // (n, p1_max) or NULL   /
	unsigned long long* key; // (n, p1_max) or NULL: merged keys; plain store when splits == 1, atomicMin (buffer preset to ~0) otherwise
};
struct NnArgs {
	NnDir d[2];
	int n_dirs, splits;
	unsigned* counter;   // or NULL: zeroed by the first block (the block counter of the reduction that follows this launch)
};

__device__ __forceinline__ float nn_dist(float qx, float qy, float qz, float tx, float ty, float tz) {
	const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
	return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ unsigned long long nn_key(float d, int i) {
	return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
}

template <int NP>
__global__ __launch_bounds__(256) void nn_kernel(const NnArgs a) {
	constexpr int NQ = 2 * NP;
	__shared__ float4 ty[NN_TILE];
	__shared__ unsigned long long pkey[3][64 * NQ];
	const int dir = blockIdx.z / a.splits, split = blockIdx.z - dir * a.splits;
	if (a.counter && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) *a.counter = 0u;
	// (copy the fields: a reference into the kernel-argument array would move it to scratch)
	const float* __restrict__ x = dir ? a.d[1].x : a.d[0].x;
	const float* __restrict__ y = dir ? a.d[1].y : a.d[0].y;
	const int32_t* x_len = dir ? a.d[1].x_len : a.d[0].x_len;
	const int32_t* y_len = dir ? a.d[1].y_len : a.d[0].y_len;
	const int p1_max = dir ? a.d[1].p1_max : a.d[0].p1_max, p2_max = dir ? a.d[1].p2_max : a.d[0].p2_max;
	if ((int)blockIdx.x * 64 * NQ >= p1_max) return;
	const int n = blockIdx.y;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int p1 = x_len ? x_len[n] : p1_max;
	const int p2 = y_len ? y_len[n] : p2_max;
	const float* xp = x + (int64_t)n * p1_max * 3;
	const float* yp = y + (int64_t)n * p2_max * 3;
	// this block's share of the targets: whole groups of four
	const int share = (int)((((int64_t)p2 + a.splits - 1) / a.splits + 3) & ~3ll);
	const int lo = min(p2, split * share), hi = min(p2, lo + share);
	v2f qx[NP], qy[NP], qz[NP];
	float best[NQ];
	int gj[NQ], qi[NQ];
#pragma unroll
	for (int k = 0; k < NQ; ++k) {
		qi[k] = (blockIdx.x * NQ + k) * 64 + lane;
		const int i = min(qi[k], p1_max - 1);
		qx[k >> 1][k & 1] = xp[i * 3 + 0]; qy[k >> 1][k & 1] = xp[i * 3 + 1]; qz[k >> 1][k & 1] = xp[i * 3 + 2];
		best[k] = INFINITY; gj[k] = -1;
	}
	for (int j0 = lo; j0 < hi; j0 += NN_TILE) {
		const int cnt = min(NN_TILE, hi - j0);
		const int cnt4 = (cnt + 3) & ~3;
		__syncthreads();
		for (int j = threadIdx.x; j < cnt4; j += 256) {
			const float* s = yp + (int64_t)(j0 + min(j, cnt - 1)) * 3;
			// the last group is padded with targets at an infinite distance: never '<' the running best
			ty[j] = j < cnt ? make_float4(s[0], s[1], s[2], 0.f) : make_float4(1e30f, 1e30f, 1e30f, 0.f);
		}
		__syncthreads();
		const int ja = __builtin_amdgcn_readfirstlane(wave * (NN_TILE / 4));
		const int jb = __builtin_amdgcn_readfirstlane(min(cnt4, ja + NN_TILE / 4));
#pragma unroll 2
		for (int j = ja; j < jb; j += 4) {
			const float4 t0 = ty[j], t1 = ty[j + 1], t2 = ty[j + 2], t3 = ty[j + 3];
#pragma unroll
			for (int k = 0; k < NP; ++k) {
#define FIND_NN_D2(t) ({ const v2f dx = qx[k] - t.x, dy = qy[k] - t.y, dz = qz[k] - t.z;                                 \
						 __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)); })
				const v2f d0 = FIND_NN_D2(t0), d1 = FIND_NN_D2(t1), d2 = FIND_NN_D2(t2), d3 = FIND_NN_D2(t3);
#undef FIND_NN_D2
				const float m0 = fminf(fminf(d0.x, d1.x), fminf(d2.x, d3.x));
				const float m1 = fminf(fminf(d0.y, d1.y), fminf(d2.y, d3.y));
				if (m0 < best[2 * k]) { best[2 * k] = m0; gj[2 * k] = j0 + j; }
				if (m1 < best[2 * k + 1]) { best[2 * k + 1] = m1; gj[2 * k + 1] = j0 + j; }
			}
		}
	}
	// the winning group -> the first of its targets at the best distance
	unsigned long long key[NQ];
#pragma unroll
	for (int k = 0; k < NQ; ++k) {
		key[k] = ~0ull;
		if (gj[k] >= 0) {
			const float fx = qx[k >> 1][k & 1], fy = qy[k >> 1][k & 1], fz = qz[k >> 1][k & 1];
			int jj = gj[k];
#pragma unroll
			for (int c = 3; c >= 0; --c) {
				const int j = gj[k] + c;
				if (j < hi) {
					const float* s = yp + (int64_t)j * 3;
					if (nn_dist(fx, fy, fz, s[0], s[1], s[2]) == best[k]) jj = j;
				}
			}
			key[k] = nn_key(best[k], jj);
		}
	}
	if (wave) {
#pragma unroll
		for (int k = 0; k < NQ; ++k) pkey[wave - 1][k * 64 + lane] = key[k];
	}
	__syncthreads();
	if (wave == 0) {
		float* dist = dir ? a.d[1].dist : a.d[0].dist;
		int32_t* idx = dir ? a.d[1].idx : a.d[0].idx;
		unsigned long long* kout = dir ? a.d[1].key : a.d[0].key;
#pragma unroll
		for (int k = 0; k < NQ; ++k) {
			unsigned long long b = key[k];
#pragma unroll
			for (int w = 0; w < 3; ++w) b = min(b, pkey[w][k * 64 + lane]);
			if (qi[k] >= p1_max) continue;
			const int64_t o = (int64_t)n * p1_max + qi[k];
			const bool valid = qi[k] < p1 && b != ~0ull;
			if (kout) {
				if (a.splits == 1) kout[o] = qi[k] < p1 ? b : ~0ull;
				else if (valid) atomicMin(kout + o, b);
			}
			if (dist) dist[o] = valid ? __uint_as_float((unsigned)(b >> 32)) : 0.f;
			if (idx) idx[o] = valid ? (int)(unsigned)b : -1;
		}
	}
}

// Chamfer loss from the merged keys of both directions (pytorch3d.loss.chamfer_distance defaults: point mean, batch mean):
//   loss = ( sum_n sum_i d_xy[n,i] / max(len_x[n],1)  +  sum_n sum_j d_yx[n,j] / max(len_y[n],1) ) / N
// Block (n, dir) sums one cloud in a fixed thread -> element mapping and reduction tree; the block that finishes last (a counter in
// the workspace, zeroed by the nearest-neighbour launch before) adds the 2 N partial means in index order: deterministic.
__global__ __launch_bounds__(1024) void chamfer_reduce_kernel(const unsigned long long* __restrict__ kx, const int32_t* __restrict__ x_len, int p1_max,
															  const unsigned long long* __restrict__ ky, const int32_t* __restrict__ y_len, int p2_max,
															  int n_clouds, float* __restrict__ partial, unsigned* __restrict__ counter,
															  float* __restrict__ loss) {
	__shared__ float red[16];
	__shared__ bool last;
	const int n = blockIdx.x, dir = blockIdx.y;
	const unsigned long long* k = dir ? ky : kx;
	const int32_t* len = dir ? y_len : x_len;
	const int pmax = dir ? p2_max : p1_max;
	const int p = len ? len[n] : pmax;
	float s = 0.f;
	for (int i = threadIdx.x; i < p; i += 1024) {
		const unsigned long long v = k[(int64_t)n * pmax + i];
		s += v == ~0ull ? 0.f : __uint_as_float((unsigned)(v >> 32));
	}
	s = wave_sum(s);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		float t = 0.f;
		for (int w = 0; w < 16; ++w) t += red[w];
		partial[dir * n_clouds + n] = t / (float)max(p, 1);
		__threadfence();
		last = atomicAdd(counter, 1u) == 2u * n_clouds - 1u;
	}
	__syncthreads();
	if (last && threadIdx.x == 0) {
		__threadfence();
		float total = 0.f;
		for (int i = 0; i < 2 * n_clouds; ++i) total += __builtin_nontemporal_load(partial + i);
		*loss = total / (float)max(n_clouds, 1);
	}
}

// Gradient of that loss: for every valid point i of either direction with nearest neighbour j,
//   g = 2 * (*g_loss) / (N * max(len,1)) * (a_i - b_j);   d_a[i] += g;  d_b[j] -= g      (d_x / d_y zero-initialised by the caller)
__global__ void chamfer_bwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ x_len, int p1_max, const float* __restrict__ y,
								   const int32_t* __restrict__ y_len, int p2_max, const unsigned long long* __restrict__ kx,
								   const unsigned long long* __restrict__ ky, const float* __restrict__ g_loss, int n_clouds,
								   float* __restrict__ d_x, float* __restrict__ d_y) {
	const int n = blockIdx.y, dir = blockIdx.z;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int pa_max = dir ? p2_max : p1_max, pb_max = dir ? p1_max : p2_max;
	const int32_t* la = dir ? y_len : x_len;
	const int pa = la ? la[n] : pa_max;
	if (i >= pa) return;
	const unsigned long long v = (dir ? ky : kx)[(int64_t)n * pa_max + i];
	if (v == ~0ull) return;
	const int j = (int)(unsigned)v;
	const float* ap = (dir ? y : x) + ((int64_t)n * pa_max + i) * 3;
	const float* bp = (dir ? x : y) + ((int64_t)n * pb_max + j) * 3;
	const float g = 2.0f * (*g_loss) / ((float)n_clouds * (float)max(pa, 1));
	const float gx = g * (ap[0] - bp[0]), gy = g * (ap[1] - bp[1]), gz = g * (ap[2] - bp[2]);
	float* da = dir ? d_y : d_x;
	float* db = dir ? d_x : d_y;
	if (da) {
		float* o = da + ((int64_t)n * pa_max + i) * 3;
		unsafeAtomicAdd(o + 0, gx); unsafeAtomicAdd(o + 1, gy); unsafeAtomicAdd(o + 2, gz);
	}
	if (db) {
		float* o = db + ((int64_t)n * pb_max + j) * 3;
		unsafeAtomicAdd(o + 0, -gx); unsafeAtomicAdd(o + 1, -gy); unsafeAtomicAdd(o + 2, -gz);
	}
}

__global__ void nn_bwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ x_len, const float* __restrict__ y,
							  const int32_t* __restrict__ idx, const float* __restrict__ w, int p1_max, int p2_max,
							  float* __restrict__ d_x, float* __restrict__ d_y) {
	const int n = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int p1 = x_len ? x_len[n] : p1_max;
	if (i >= p1) return;
	const int64_t o = (int64_t)n * p1_max + i;
	const int j = idx[o];
	if (j < 0) return;
	const float g = 2.0f * w[o];
	const float* xp = x + o * 3;
	const float* yp = y + ((int64_t)n * p2_max + j) * 3;
	const float gx = g * (xp[0] - yp[0]), gy = g * (xp[1] - yp[1]), gz = g * (xp[2] - yp[2]);
	if (d_x) {
		// each x row is owned by one thread, but the other Chamfer direction may already have written here
		d_x[o * 3 + 0] += gx; d_x[o * 3 + 1] += gy; d_x[o * 3 + 2] += gz;
	}
	if (d_y) {
		float* dy = d_y + ((int64_t)n * p2_max + j) * 3;
		atomicAdd(dy + 0, -gx); atomicAdd(dy + 1, -gy); atomicAdd(dy + 2, -gz);
	}
}

// ------------------------------------------------------------------------------------------- smoothness
// Static topology (one template): vertex -> incident (face, corner) CSR and vertex -> neighbour CSR are built once on
// the host; every kernel below is a deterministic gather (no float atomics).
//
// cot weights per face (ops/laplacian_matrices.py): A=|v1-v2|, B=|v0-v2|, C=|v0-v1|, area = sqrt(max(heron, 1e-12)),
// cot = [(B2+C2-A2), (A2+C2-B2), (A2+B2-C2)] / area / 4;  L[v1,v2]+=cot_a, L[v2,v0]+=cot_b, L[v0,v1]+=cot_c, symmetrised.
__global__ void cot_weights_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int n_verts, int n_faces,
								   float* __restrict__ fw) {
	const int m = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_faces) return;
	const int32_t* fp = faces + (int64_t)f * 3;
	const float* vp = verts + (int64_t)m * n_verts * 3;
	const float3 v0 = ld3(vp + 3 * fp[0]), v1 = ld3(vp + 3 * fp[1]), v2 = ld3(vp + 3 * fp[2]);
	const float A = norm3(sub3(v1, v2)), B = norm3(sub3(v0, v2)), C = norm3(sub3(v0, v1));
	const float s = 0.5f * (A + B + C);
	const float area = sqrtf(fmaxf(s * (s - A) * (s - B) * (s - C), 1e-12f));
	const float A2 = A * A, B2 = B * B, C2 = C * C;
	float* o = fw + ((int64_t)m * n_faces + f) * 3;
	o[0] = (B2 + C2 - A2) / area / 4.0f;
	o[1] = (A2 + C2 - B2) / area / 4.0f;
	o[2] = (A2 + B2 - C2) / area / 4.0f;
}

// r_i = sum over incident corners of [ w1 * q_j1 + w2 * q_j2 ]  (= (L q)_i),  rowsum_i = sum (w1 + w2).
// Corner c of face (a0,a1,a2) sees neighbour a[(c+1)%3] with weight cot[(c+2)%3] and a[(c+2)%3] with weight cot[(c+1)%3].
// FOUR LANES PER VERTEX: lane t of a quad walks the corner entries lo + t, lo + t + 4, ... and the quad adds up with two xor-shuffles.
// Every entry is a chain of dependent loads (item -> face corners -> vertex rows), so a vertex of valence 6 was six such chains one
// after the other in one lane (40 us for 16 x 6890 vertices, most of the chip idle); per quad lane it is at most two.
// Every lane of the wave has to call this (shuffles): a quad whose vertex is past the end passes live = false.
__device__ __forceinline__ float quad_sum(float v) {
	v += __shfl_xor(v, 1, 64);
	v += __shfl_xor(v, 2, 64);
	return v;
}

// Q(j) -> float3: the vector of vertex j the operator is applied to
template <typename Q>
__device__ __forceinline__ void apply_L(Q qv, const int32_t* __restrict__ faces, const float* __restrict__ fw,
										const int32_t* __restrict__ vf_off, const int32_t* __restrict__ vf_items, int i, int t, bool live,
										float3* r, float* rowsum) {
	float3 acc = make_float3(0.f, 0.f, 0.f);
	float rs = 0.f;
	if (live) {
		const int hi = vf_off[i + 1];
		for (int e = vf_off[i] + t; e < hi; e += 4) {
			const int item = vf_items[e];
			const int f = item / 3, c = item - f * 3;
			const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
			const int j1 = faces[f * 3 + c1], j2 = faces[f * 3 + c2];
			const float w1 = fw[f * 3 + c2], w2 = fw[f * 3 + c1];
			const float3 a = qv(j1), b = qv(j2);
			acc.x += w1 * a.x + w2 * b.x; acc.y += w1 * a.y + w2 * b.y; acc.z += w1 * a.z + w2 * b.z;
			rs += w1 + w2;
		}
	}
	*r = make_float3(quad_sum(acc.x), quad_sum(acc.y), quad_sum(acc.z));
	*rowsum = quad_sum(rs);
}

// vertices per block: 256 threads = 64 quads.  (Small blocks on purpose: inside a training step these launches run beside the Chamfer
// search / the texture pass's trailing weight-gradient kernels, and a 16-wave block found room on a CU only when one of theirs retired --
// the forward took 100 us beside nn_kernel against 18 us alone.)
constexpr int SMOOTH_VPB = 64;

// forward per vertex: lap = (L V)_i * nw_i - V_i; block partial sums of |lap| and of the half edge-length sums.
// Saves nw_i (rowsum>0 ? 1/rowsum : rowsum) and u_i = lap_i/|lap_i| scaled later in backward.
__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
														   const float* __restrict__ fw, const int32_t* __restrict__ vf_off,
														   const int32_t* __restrict__ vf_items, const int32_t* __restrict__ nbr_off,
														   const int32_t* __restrict__ nbr_idx, int n_verts, int n_faces,
														   float* __restrict__ nw_out, float* __restrict__ lapdir_out,
														   float* __restrict__ partial /* [n_meshes][gridDim.x][2] */) {
	const int m = blockIdx.y;
	const int i = blockIdx.x * SMOOTH_VPB + ((int)threadIdx.x >> 2), t = threadIdx.x & 3;
	const bool live = i < n_verts;
	const float* vp = verts + (int64_t)m * n_verts * 3;
	float3 r;
	float rs;
	apply_L([&](int j) { return ld3(vp + 3 * j); }, faces, fw + (int64_t)m * n_faces * 3, vf_off, vf_items, i, t, live, &r, &rs);
	float lap_n = 0.f, edge_s = 0.f;
	if (live) {
		const float3 v = ld3(vp + 3 * i);
		const int hi = nbr_off[i + 1];
		for (int e = nbr_off[i] + t; e < hi; e += 4) {
			const float3 d = sub3(v, ld3(vp + 3 * nbr_idx[e]));
			edge_s += d.x * d.x + d.y * d.y + d.z * d.z;  // every undirected edge is visited from both ends
		}
		if (t == 0) {
			const float nw = rs > 0.f ? 1.0f / rs : rs;
			const float3 lap = make_float3(r.x * nw - v.x, r.y * nw - v.y, r.z * nw - v.z);
			lap_n = norm3(lap);
			const float inv = lap_n > 0.f ? 1.0f / lap_n : 0.f;
			const int64_t o = (int64_t)m * n_verts + i;
			nw_out[o] = nw;
			lapdir_out[o * 3 + 0] = lap.x * inv; lapdir_out[o * 3 + 1] = lap.y * inv; lapdir_out[o * 3 + 2] = lap.z * inv;
		}
	}
	__shared__ float red[2][4];
	const float a = wave_sum(lap_n), b = wave_sum(edge_s);
	if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
	__syncthreads();
	if (threadIdx.x < 2) {
		float sum = 0.f;
		for (int w = 0; w < 4; ++w) sum += red[threadIdx.x][w];
		partial[((int64_t)m * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = sum;
	}
}

__global__ void smooth_finalize_kernel(const float* __restrict__ partial, int n_meshes, int nblk, int n_verts, int n_edges,
									   float* __restrict__ loss_edge, float* __restrict__ loss_lap, float w_edge, float w_lap,
									   float* __restrict__ loss_sum) {
	// single wave; deterministic order
	float lap = 0.f, edge = 0.f;
	for (int k = threadIdx.x; k < n_meshes * nblk; k += 64) { lap += partial[k * 2 + 0]; edge += partial[k * 2 + 1]; }
	lap = wave_sum(lap);
	edge = wave_sum(edge);
	if (threadIdx.x == 0) {
		const float l = lap / (float)n_verts / (float)n_meshes, e = 0.5f * edge / (float)n_edges / (float)n_meshes;
		if (loss_lap) *loss_lap = l;
		if (loss_edge) *loss_edge = e;
		if (loss_sum) *loss_sum = w_lap * l + w_edge * e;   // MeshSmoothnessLoss: 0.1 * laplacian + 10 * edge (losses.py:95-99)
	}
}

// backward: with u_i = g_lap/(V N) * lapdir_i and q_i = nw_i * u_i:   dV_i = (L q)_i - u_i  +  g_edge/(E N) * 2 * sum_j (v_i - v_j)
// (q is formed where it is read -- nw_j and lapdir_j instead of a stored q_j: one launch and one (N, V, 3) buffer less)
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
														   const float* __restrict__ fw, const int32_t* __restrict__ vf_off,
														   const int32_t* __restrict__ vf_items, const int32_t* __restrict__ nbr_off,
														   const int32_t* __restrict__ nbr_idx, const float* __restrict__ nw,
														   const float* __restrict__ lapdir, const float* __restrict__ g_edge,
														   const float* __restrict__ g_lap, float s_edge, float s_lap, int n_meshes, int n_verts,
														   int n_faces, int n_edges, float* __restrict__ d_verts) {
	const int m = blockIdx.y;
	const int i = blockIdx.x * SMOOTH_VPB + ((int)threadIdx.x >> 2), t = threadIdx.x & 3;
	const bool live = i < n_verts;
	const float* vp = verts + (int64_t)m * n_verts * 3;
	const float* nwp = nw + (int64_t)m * n_verts;
	const float* ldp = lapdir + (int64_t)m * n_verts * 3;
	const float su = (*g_lap) * s_lap / (float)n_verts / (float)n_meshes;
	float3 r;
	float rs;
	apply_L([&](int j) { const float sj = su * nwp[j]; const float3 u = ld3(ldp + 3 * j); return make_float3(sj * u.x, sj * u.y, sj * u.z); },
			faces, fw + (int64_t)m * n_faces * 3, vf_off, vf_items, i, t, live, &r, &rs);
	float3 es = make_float3(0.f, 0.f, 0.f);
	float3 v = es;
	if (live) {
		v = ld3(vp + 3 * i);
		const int hi = nbr_off[i + 1];
		for (int e = nbr_off[i] + t; e < hi; e += 4) {
			const float3 d = sub3(v, ld3(vp + 3 * nbr_idx[e]));
			es.x += d.x; es.y += d.y; es.z += d.z;
		}
	}
	es = make_float3(quad_sum(es.x), quad_sum(es.y), quad_sum(es.z));
	if (live && t == 0) {
		const int64_t o = (int64_t)m * n_verts + i;
		const float se = 2.0f * (*g_edge) * s_edge / (float)n_edges / (float)n_meshes;
		d_verts[o * 3 + 0] = r.x - su * ldp[3 * i + 0] + se * es.x;
		d_verts[o * 3 + 1] = r.y - su * ldp[3 * i + 1] + se * es.y;
		d_verts[o * 3 + 2] = r.z - su * ldp[3 * i + 2] + se * es.z;
	}
}

struct SmoothWs {
	float* fw;       // (n_meshes, n_faces, 3)
	float* nw;       // (n_meshes, n_verts)
	float* lapdir;   // (n_meshes, n_verts, 3)
	float* partial;  // (n_meshes, nblk, 2)
	int nblk;
	int64_t bytes;
};

static void carve_smooth(int64_t n_meshes, int64_t n_verts, int64_t n_faces, void* ws, SmoothWs* o) {
	Carver c(ws);
	o->nblk = (int)cdiv(n_verts, SMOOTH_VPB);
	o->fw = c.take<float>(n_meshes * n_faces * 3);
	o->nw = c.take<float>(n_meshes * n_verts);
	o->lapdir = c.take<float>(n_meshes * n_verts * 3);
	o->partial = c.take<float>(n_meshes * o->nblk * 2);
	o->bytes = c.off;
}

}  // namespace geom
}  // namespace find

using namespace find;
using namespace find::geom;

static inline bool bad_dims(int64_t a, int64_t b) { return a < 1 || b < 1 || a >= (1 << 16) || b >= (1ll << 30); }

extern "C" int find_face_areas(const float* verts, const int32_t* faces, int64_t faces_batch, int64_t n_meshes, int64_t n_verts,
							   int64_t n_faces, float* areas, void* stream) {
	FIND_REQUIRE(verts && faces && areas, "find_face_areas: NULL argument");
	FIND_REQUIRE(!bad_dims(n_meshes, n_faces) && n_verts >= 1, "find_face_areas: bad sizes");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_face_areas: faces_batch must be 1 or n_meshes");
	hipLaunchKernelGGL(face_areas_kernel, dim3((unsigned)cdiv(n_faces, 256), (unsigned)n_meshes), dim3(256), 0, (hipStream_t)stream, verts,
					   faces, faces_batch == 1 ? 0 : n_faces * 3, (int)n_verts, (int)n_faces, areas);
	FIND_LAUNCH_CHECK("face_areas_kernel");
	return FIND_OK;
}

extern "C" int find_sample_points_fwd(const float* verts, const int32_t* faces, int64_t faces_batch, const int32_t* face_idx,
									  const float* uv, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_samples,
									  float* out, const float* attr, float* attr_out, void* stream) {
	FIND_REQUIRE(verts && faces && face_idx && uv && out, "find_sample_points_fwd: NULL argument");
	FIND_REQUIRE((attr == nullptr) == (attr_out == nullptr), "find_sample_points_fwd: attr and attr_out must both be given or both NULL");
	FIND_REQUIRE(!bad_dims(n_meshes, n_samples) && n_verts >= 1 && n_faces >= 1, "find_sample_points_fwd: bad sizes");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_sample_points_fwd: faces_batch must be 1 or n_meshes");
	hipLaunchKernelGGL(sample_fwd_kernel, dim3((unsigned)cdiv(n_samples, 256), (unsigned)n_meshes), dim3(256), 0, (hipStream_t)stream, verts,
					   faces, faces_batch == 1 ? 0 : n_faces * 3, face_idx, uv, (int)n_verts, (int)n_samples, out, attr, attr_out);
	FIND_LAUNCH_CHECK("sample_fwd_kernel");
	return FIND_OK;
}

extern "C" int find_sample_points_bwd(const int32_t* faces, int64_t faces_batch, const int32_t* face_idx, const float* uv,
									  const float* d_out, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_samples,
									  float* d_verts, void* stream) {
	FIND_REQUIRE(faces && face_idx && uv && d_out && d_verts, "find_sample_points_bwd: NULL argument");
	FIND_REQUIRE(!bad_dims(n_meshes, n_samples) && n_verts >= 1 && n_faces >= 1, "find_sample_points_bwd: bad sizes");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_sample_points_bwd: faces_batch must be 1 or n_meshes");
	hipLaunchKernelGGL(sample_bwd_kernel, dim3((unsigned)cdiv(n_samples, 256), (unsigned)n_meshes), dim3(256), 0, (hipStream_t)stream, faces,
					   faces_batch == 1 ? 0 : n_faces * 3, face_idx, uv, d_out, (int)n_verts, (int)n_samples, d_verts);
	FIND_LAUNCH_CHECK("sample_bwd_kernel");
	return FIND_OK;
}


// ------------------------------------------------------------------------------------------------ nearest neighbour through a uniform grid
// The brute-force kernel above looks at every (query, target) pair: 8 flops x P1 x P2.  For clouds of a few thousand points and more, the
// targets of a cloud are first sorted into a NG^3 uniform grid over their bounding box (one workgroup per cloud and direction: bounding
// box, cell histogram and running sum in LDS, scatter), and a query then scans the cells around its own in growing cubes until the best
// distance found is smaller than the distance to anything outside the cube scanned.  EXACT, and the same answer as the brute-force kernel
// to the bit: a pair's distance is nn_dist's three roundings on the same operands, the result is the smallest 64-bit key
// (distance bits << 32 | index) seen -- lowest index among equal distances, whatever order the cells hand the targets out in --, and the
// stopping rule keeps a margin far above that rounding, so a cube is only left when nothing outside it can win or tie.
constexpr int NG = 16;                 // cells per axis
constexpr int NG3 = NG * NG * NG;
struct GridDir {
	const float* y;          // targets (n, p2_max, 3)
	const int32_t* y_len;
	int p2_max;
	float4* sorted;          // (n, p2_max): x, y, z, original index (bits)
	int32_t* cell_start;     // (n, NG3 + 1)
	float* box;              // (n, 8): origin xyz, inverse cell sizes xyz
};
struct GridBuildArgs { GridDir d[2]; };

__global__ __launch_bounds__(1024) void nn_grid_build_kernel(const GridBuildArgs a) {
	__shared__ int cnt[NG3];
	__shared__ float red[6][16];
	__shared__ int wsum[16];
	__shared__ float bx[8];
	const int dir = blockIdx.y, n = blockIdx.x;
	const float* __restrict__ y = dir ? a.d[1].y : a.d[0].y;
	const int32_t* y_len = dir ? a.d[1].y_len : a.d[0].y_len;
	const int p2_max = dir ? a.d[1].p2_max : a.d[0].p2_max;
	float4* sorted = (dir ? a.d[1].sorted : a.d[0].sorted) + (int64_t)n * p2_max;
	int32_t* cell_start = (dir ? a.d[1].cell_start : a.d[0].cell_start) + (int64_t)n * (NG3 + 1);
	float* box = (dir ? a.d[1].box : a.d[0].box) + (int64_t)n * 8;
	const int p2 = y_len ? y_len[n] : p2_max;
	const float* yp = y + (int64_t)n * p2_max * 3;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// ---- bounding box
	float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (int j = tid; j < p2; j += 1024)
#pragma unroll
		for (int c = 0; c < 3; ++c) { const float v = yp[(int64_t)j * 3 + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
#pragma unroll
	for (int c = 0; c < 3; ++c)
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], d, 64)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], d, 64)); }
	if (lane == 0)
#pragma unroll
		for (int c = 0; c < 3; ++c) { red[c][wave] = lo[c]; red[3 + c][wave] = hi[c]; }
	for (int i = tid; i < NG3; i += 1024) cnt[i] = 0;
	__syncthreads();
	if (tid == 0) {
		float ext[3], emax = 0.f;
		for (int c = 0; c < 3; ++c) {
			float l = INFINITY, h = -INFINITY;
			for (int w = 0; w < 16; ++w) { l = fminf(l, red[c][w]); h = fmaxf(h, red[3 + c][w]); }
			if (!(l <= h)) { l = 0.f; h = 0.f; }   // empty cloud
			bx[c] = l;
			ext[c] = h - l;
			emax = fmaxf(emax, ext[c]);
		}
		// NG cells along every axis of the bounding box (a foot is three times as long as it is wide: cubes would leave most of the grid
		// empty), none thinner than a hundredth of the longest side
		for (int c = 0; c < 3; ++c) {
			const float cell = fmaxf(fmaxf(ext[c], 0.01f * emax) * (1.0f / NG) * 1.0001f, 1e-12f);
			bx[3 + c] = 1.0f / cell;
			box[c] = bx[c]; box[3 + c] = 1.0f / cell;
		}
	}
	__syncthreads();
	const float ox = bx[0], oy = bx[1], oz = bx[2], ivx = bx[3], ivy = bx[4], ivz = bx[5];
	auto cell_of = [&](float x, float yv, float z) {
		const int cx = min(max((int)((x - ox) * ivx), 0), NG - 1), cy = min(max((int)((yv - oy) * ivy), 0), NG - 1), cz = min(max((int)((z - oz) * ivz), 0), NG - 1);
		return (cz * NG + cy) * NG + cx;
	};
	// ---- histogram, running sum (four cells per thread, then across the block), scatter
	for (int j = tid; j < p2; j += 1024) atomicAdd(&cnt[cell_of(yp[(int64_t)j * 3], yp[(int64_t)j * 3 + 1], yp[(int64_t)j * 3 + 2])], 1);
	__syncthreads();
	static_assert(NG3 == 4 * 1024, "four cells per thread");
	const int c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
	const int own = c0 + c1 + c2 + c3;
	int inc = own;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
	if (lane == 63) wsum[wave] = inc;
	__syncthreads();
	int before = 0;
	for (int w = 0; w < wave; ++w) before += wsum[w];
	const int ex = before + inc - own;
	__syncthreads();
	cnt[4 * tid] = ex; cnt[4 * tid + 1] = ex + c0; cnt[4 * tid + 2] = ex + c0 + c1; cnt[4 * tid + 3] = ex + c0 + c1 + c2;
	cell_start[4 * tid] = ex; cell_start[4 * tid + 1] = ex + c0; cell_start[4 * tid + 2] = ex + c0 + c1; cell_start[4 * tid + 3] = ex + c0 + c1 + c2;
	if (tid == 1023) cell_start[NG3] = ex + own;
	__syncthreads();
	for (int j = tid; j < p2; j += 1024) {
		const float x = yp[(int64_t)j * 3], yv = yp[(int64_t)j * 3 + 1], z = yp[(int64_t)j * 3 + 2];
		const int pos = atomicAdd(&cnt[cell_of(x, yv, z)], 1);   // (the order inside a cell is whatever the atomics make it: the keys decide)
		sorted[pos] = make_float4(x, yv, z, __int_as_float(j));
	}
}

struct GridQueryDir {
	const float* x;          // queries (n, p1_max, 3)
	const int32_t* x_len;
	int p1_max, p2_max;
	const float4* sorted;
	const int32_t* cell_start;
	const float* box;
	unsigned long long* key; // (n, p1_max)
};
struct GridQueryArgs { GridQueryDir d[2]; unsigned* counter; };

__global__ __launch_bounds__(256) void nn_grid_query_kernel(const GridQueryArgs a) {
	const int dir = blockIdx.z, n = blockIdx.y;
	if (a.counter && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) *a.counter = 0u;
	const int p1_max = dir ? a.d[1].p1_max : a.d[0].p1_max, p2_max = dir ? a.d[1].p2_max : a.d[0].p2_max;
	const int qi = blockIdx.x * 256 + threadIdx.x;
	if (qi >= p1_max) return;
	const float* __restrict__ x = dir ? a.d[1].x : a.d[0].x;
	const int32_t* x_len = dir ? a.d[1].x_len : a.d[0].x_len;
	const float4* __restrict__ sorted = (dir ? a.d[1].sorted : a.d[0].sorted) + (int64_t)n * p2_max;
	const int32_t* __restrict__ cs = (dir ? a.d[1].cell_start : a.d[0].cell_start) + (int64_t)n * (NG3 + 1);
	const float* box = (dir ? a.d[1].box : a.d[0].box) + (int64_t)n * 8;
	unsigned long long* kout = (dir ? a.d[1].key : a.d[0].key) + (int64_t)n * p1_max;
	const int p1 = x_len ? x_len[n] : p1_max;
	if (qi >= p1) { kout[qi] = ~0ull; return; }
	const float* q = x + ((int64_t)n * p1_max + qi) * 3;
	const float qx = q[0], qy = q[1], qz = q[2];
	const float ox = box[0], oy = box[1], oz = box[2], ivx = box[3], ivy = box[4], ivz = box[5];
	const float clx = 1.0f / ivx, cly = 1.0f / ivy, clz = 1.0f / ivz;
	const float rx = qx - ox, ry = qy - oy, rz = qz - oz;
	const int cx = min(max((int)(rx * ivx), 0), NG - 1), cy = min(max((int)(ry * ivy), 0), NG - 1), cz = min(max((int)(rz * ivz), 0), NG - 1);
	unsigned long long best = ~0ull;
	auto scan_row = [&](int z, int yy, int x0, int x1) {   // cells (x0..x1, yy, z): contiguous in the sorted order
		const int c = (z * NG + yy) * NG;
		const int j0 = cs[c + x0], j1 = cs[c + x1 + 1];
		for (int j = j0; j < j1; ++j) {
			const float4 t = sorted[j];
			best = min(best, nn_key(nn_dist(qx, qy, qz, t.x, t.y, t.z), __float_as_int(t.w)));
		}
	};
	for (int r = 0; r < NG; ++r) {
		const int x0 = max(cx - r, 0), x1 = min(cx + r, NG - 1), y0 = max(cy - r, 0), y1 = min(cy + r, NG - 1), z0 = max(cz - r, 0), z1 = min(cz + r, NG - 1);
		// the shell of the cube of radius r (everything inside it was scanned by the smaller cubes)
		for (int z = z0; z <= z1; ++z)
			for (int yy = y0; yy <= y1; ++yy) {
				const bool face = (z == cz - r) || (z == cz + r) || (yy == cy - r) || (yy == cy + r);
				if (face || r == 0) scan_row(z, yy, x0, x1);
				else {
					if (cx - r >= 0) scan_row(z, yy, cx - r, cx - r);
					if (cx + r <= NG - 1) scan_row(z, yy, cx + r, cx + r);
				}
			}
		if (x0 == 0 && x1 == NG - 1 && y0 == 0 && y1 == NG - 1 && z0 == 0 && z1 == NG - 1) break;   // the whole grid
		// anything not scanned yet lies beyond a face of the cube that is inside the grid: at least `gap` away along that axis
		float gap = INFINITY;
		// Distances to the cube's faces are formed RELATIVE to the box origin (q - o once, then minus k cells): in absolute coordinates
		// one ulp of an origin far from zero (a scan in millimetres placed a metre away) would exceed any fixed share of a cell.  The
		// margin covers the rounding of q - o and t - o on both sides of a cell boundary: a ten-thousandth of a cell plus a few ulps of
		// the larger of |origin| and |query|.
		const float mx = 1e-4f * clx + 6e-7f * fmaxf(fabsf(ox), fabsf(qx)), my = 1e-4f * cly + 6e-7f * fmaxf(fabsf(oy), fabsf(qy)),
					mz = 1e-4f * clz + 6e-7f * fmaxf(fabsf(oz), fabsf(qz));
		if (cx - r > 0) gap = fminf(gap, rx - (cx - r) * clx - mx);
		if (cx + r < NG - 1) gap = fminf(gap, (cx + r + 1) * clx - rx - mx);
		if (cy - r > 0) gap = fminf(gap, ry - (cy - r) * cly - my);
		if (cy + r < NG - 1) gap = fminf(gap, (cy + r + 1) * cly - ry - my);
		if (cz - r > 0) gap = fminf(gap, rz - (cz - r) * clz - mz);
		if (cz + r < NG - 1) gap = fminf(gap, (cz + r + 1) * clz - rz - mz);
		gap = fmaxf(gap, 0.f);
		if (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) < gap * gap * 0.9999f) break;
	}
	kout[qi] = best;
}

// queries per lane and target splits of a launch: enough blocks to give every SIMD work at batch 1, whole waves of blocks at batch 16
static void nn_shape(int64_t blocks_np1, int64_t p2_max, int* np, int* splits) {
	*np = blocks_np1 >= 4096 ? 2 : 1;
	*splits = 1;
	while (blocks_np1 * *splits < 512 && *splits < 8 && p2_max / (*splits * 2) >= 256) *splits *= 2;
}

static int launch_nn(const NnArgs& a, int np, int64_t p_max, int64_t n, hipStream_t s) {
	const dim3 grid((unsigned)cdiv(p_max, 128 * np), (unsigned)n, (unsigned)(a.n_dirs * a.splits));
	if (np == 2) hipLaunchKernelGGL(nn_kernel<2>, grid, dim3(256), 0, s, a);
	else hipLaunchKernelGGL(nn_kernel<1>, grid, dim3(256), 0, s, a);
	return check_launch("nn_kernel");
}

extern "C" int find_nn_fwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n, int64_t p1_max,
						   int64_t p2_max, float* dist, int32_t* idx, void* stream) {
	FIND_REQUIRE(x && y && dist && idx, "find_nn_fwd: NULL argument");
	FIND_REQUIRE(!bad_dims(n, p1_max) && p2_max >= 1 && p2_max < (1ll << 30), "find_nn_fwd: bad sizes");
	NnArgs a = {};
	a.d[0] = NnDir{x, x_len, y, y_len, (int)p1_max, (int)p2_max, dist, idx, nullptr};
	a.n_dirs = 1;
	int np, splits;
	nn_shape(cdiv(p1_max, 128) * n, p2_max, &np, &splits);
	a.splits = 1;   // the direct outputs cannot be merged across blocks
	return launch_nn(a, np, p1_max, n, (hipStream_t)stream);
}

extern "C" int find_nn_bwd(const float* x, const int32_t* x_len, const float* y, const int32_t* idx, const float* w, int64_t n,
						   int64_t p1_max, int64_t p2_max, float* d_x, float* d_y, void* stream) {
	FIND_REQUIRE(x && y && idx && w, "find_nn_bwd: NULL argument");
	FIND_REQUIRE(d_x || d_y, "find_nn_bwd: both gradient outputs NULL");
	FIND_REQUIRE(!bad_dims(n, p1_max) && p2_max >= 1, "find_nn_bwd: bad sizes");
	hipLaunchKernelGGL(nn_bwd_kernel, dim3((unsigned)cdiv(p1_max, 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, x, x_len, y, idx, w,
					   (int)p1_max, (int)p2_max, d_x, d_y);
	FIND_LAUNCH_CHECK("nn_bwd_kernel");
	return FIND_OK;
}

struct ChamferWs {
	unsigned long long* kx;  // (n, p1_max)
	unsigned long long* ky;  // (n, p2_max)
	int64_t key_bytes;       // of kx + ky: what a split launch presets to ~0
	float* partial;          // (2, n) per-cloud means of the reduction
	unsigned* counter;
	float4 *sorted_x, *sorted_y;
	int32_t *cells_x, *cells_y;
	float *box_x, *box_y;
	int64_t bytes;
};
// Per cloud; below, all pairs.  Alone, the grid wins from ~2000 points on (16 x 5000 x 5000, forward + backward: 0.097 against 0.151 ms;
// 16 x 10000 x 10000: 0.18 against 0.50 ms).  But the training step runs its Chamfer term BESIDE the texture pass's MLP chain (a second
// stream, model_with_loss.py), and there the grid's dependent, scattered loads fare badly -- 430 us instead of the brute-force kernel's
// 134 us of packed arithmetic, the step 2.4 - 2.6 instead of 2.19 ms -- so the 5000-sample training clouds stay with all pairs and the
// grid serves the evaluation sizes (eval_3d.py:148: 10 000 samples).  Bits of the profiling switch (find_debug_raster_ablate): 512 = never
// the grid, 1024 = the grid from 64 points on (tests).
constexpr int64_t GRID_MIN_POINTS = 8192;
static void carve_chamfer(int64_t n, int64_t p1_max, int64_t p2_max, void* ws, ChamferWs* o) {
	Carver c(ws);
	o->kx = c.take<unsigned long long>(n * p1_max);
	o->ky = c.take<unsigned long long>(n * p2_max);
	o->key_bytes = c.off;
	o->partial = c.take<float>(2 * n);
	o->counter = c.take<unsigned>(1);
	// the uniform grids of the two clouds (nn_grid_*_kernel; used from GRID_MIN_POINTS points per cloud on)
	o->sorted_x = c.take<float4>(n * p1_max);
	o->sorted_y = c.take<float4>(n * p2_max);
	o->cells_x = c.take<int32_t>(n * (NG3 + 1));
	o->cells_y = c.take<int32_t>(n * (NG3 + 1));
	o->box_x = c.take<float>(n * 8);
	o->box_y = c.take<float>(n * 8);
	o->bytes = c.off;
}

extern "C" int64_t find_chamfer_ws_bytes(int64_t n, int64_t p1_max, int64_t p2_max) {
	if (bad_dims(n, p1_max) || bad_dims(n, p2_max)) return -1;
	ChamferWs w;
	carve_chamfer(n, p1_max, p2_max, nullptr, &w);
	return w.bytes;
}

extern "C" int find_chamfer_fwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n, int64_t p1_max, int64_t p2_max,
								float* loss, void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(x && y && loss && ws, "find_chamfer_fwd: NULL argument");
	FIND_REQUIRE(!bad_dims(n, p1_max) && !bad_dims(n, p2_max), "find_chamfer_fwd: bad sizes");
	ChamferWs w;
	carve_chamfer(n, p1_max, p2_max, ws, &w);
	if (ws_bytes < w.bytes) { set_error("find_chamfer_fwd: workspace too small"); return FIND_EWORKSPACE; }
	hipStream_t s = (hipStream_t)stream;
	const int64_t grid_from = (g_raster_ablate & 1024) ? 64 : GRID_MIN_POINTS;
	if (p1_max >= grid_from && p2_max >= grid_from && !(g_raster_ablate & 512)) {
		// targets of direction 0 (x -> y) are y's points, of direction 1 x's
		GridBuildArgs b = {};
		b.d[0] = GridDir{y, y_len, (int)p2_max, w.sorted_y, w.cells_y, w.box_y};
		b.d[1] = GridDir{x, x_len, (int)p1_max, w.sorted_x, w.cells_x, w.box_x};
		hipLaunchKernelGGL(nn_grid_build_kernel, dim3((unsigned)n, 2), dim3(1024), 0, s, b);
		FIND_LAUNCH_CHECK("nn_grid_build_kernel");
		GridQueryArgs q = {};
		q.d[0] = GridQueryDir{x, x_len, (int)p1_max, (int)p2_max, w.sorted_y, w.cells_y, w.box_y, w.kx};
		q.d[1] = GridQueryDir{y, y_len, (int)p2_max, (int)p1_max, w.sorted_x, w.cells_x, w.box_x, w.ky};
		q.counter = w.counter;
		hipLaunchKernelGGL(nn_grid_query_kernel, dim3((unsigned)cdiv(std::max(p1_max, p2_max), 256), (unsigned)n, 2), dim3(256), 0, s, q);
		FIND_LAUNCH_CHECK("nn_grid_query_kernel");
		hipLaunchKernelGGL(chamfer_reduce_kernel, dim3((unsigned)n, 2), dim3(1024), 0, s, w.kx, x_len, (int)p1_max, w.ky, y_len, (int)p2_max, (int)n, w.partial, w.counter,
						   loss);
		FIND_LAUNCH_CHECK("chamfer_reduce_kernel");
		return FIND_OK;
	}
	NnArgs a = {};
	a.d[0] = NnDir{x, x_len, y, y_len, (int)p1_max, (int)p2_max, nullptr, nullptr, w.kx};
	a.d[1] = NnDir{y, y_len, x, x_len, (int)p2_max, (int)p1_max, nullptr, nullptr, w.ky};
	a.n_dirs = 2;
	a.counter = w.counter;
	const int64_t p_max = std::max(p1_max, p2_max);
	int np;
	nn_shape(cdiv(p_max, 128) * n * 2, std::min(p1_max, p2_max), &np, &a.splits);
	if (a.splits > 1) {
		hipError_t e = hipMemsetAsync(ws, 0xff, (size_t)w.key_bytes, s);
		if (e != hipSuccess) { set_error("find_chamfer_fwd: hipMemsetAsync: %s", hipGetErrorString(e)); return FIND_ELAUNCH; }
	}
	int rc = launch_nn(a, np, p_max, n, s);
	if (rc != FIND_OK) return rc;
	hipLaunchKernelGGL(chamfer_reduce_kernel, dim3((unsigned)n, 2), dim3(1024), 0, s, w.kx, x_len, (int)p1_max, w.ky, y_len, (int)p2_max, (int)n, w.partial, w.counter,
					   loss);
	FIND_LAUNCH_CHECK("chamfer_reduce_kernel");
	return FIND_OK;
}

extern "C" int find_chamfer_bwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n, int64_t p1_max, int64_t p2_max,
								const float* g_loss, const void* ws, int64_t ws_bytes, float* d_x, float* d_y, void* stream) {
	FIND_REQUIRE(x && y && g_loss && ws, "find_chamfer_bwd: NULL argument");
	FIND_REQUIRE(d_x || d_y, "find_chamfer_bwd: both gradient outputs NULL");
	FIND_REQUIRE(!bad_dims(n, p1_max) && !bad_dims(n, p2_max), "find_chamfer_bwd: bad sizes");
	ChamferWs w;
	carve_chamfer(n, p1_max, p2_max, const_cast<void*>(ws), &w);
	if (ws_bytes < w.bytes) { set_error("find_chamfer_bwd: workspace too small"); return FIND_EWORKSPACE; }
	hipLaunchKernelGGL(chamfer_bwd_kernel, dim3((unsigned)cdiv(std::max(p1_max, p2_max), 256), (unsigned)n, 2), dim3(256), 0, (hipStream_t)stream, x, x_len,
					   (int)p1_max, y, y_len, (int)p2_max, w.kx, w.ky, g_loss, (int)n, d_x, d_y);
	FIND_LAUNCH_CHECK("chamfer_bwd_kernel");
	return FIND_OK;
}

extern "C" int64_t find_sample_surface_ws_bytes(int64_t n_meshes, int64_t n_faces) {
	if (bad_dims(n_meshes, n_faces)) return -1;
	return align_up(n_meshes * n_faces * (int64_t)sizeof(float), 256);
}

extern "C" int find_sample_surface_fwd(const float* verts, const int32_t* faces, int64_t faces_batch, const float* rnd, int64_t n_meshes, int64_t n_verts,
									   int64_t n_faces, int64_t n_samples, int32_t* face_idx, float* uv, float* out, const float* attr, float* attr_out,
									   void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(verts && faces && rnd && face_idx && uv && out && ws, "find_sample_surface_fwd: NULL argument");
	FIND_REQUIRE((attr == nullptr) == (attr_out == nullptr), "find_sample_surface_fwd: attr and attr_out must both be given or both NULL");
	FIND_REQUIRE(!bad_dims(n_meshes, n_samples) && !bad_dims(n_meshes, n_faces) && n_verts >= 1, "find_sample_surface_fwd: bad sizes");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_sample_surface_fwd: faces_batch must be 1 or n_meshes");
	if (ws_bytes < find_sample_surface_ws_bytes(n_meshes, n_faces)) { set_error("find_sample_surface_fwd: workspace too small"); return FIND_EWORKSPACE; }
	hipStream_t s = (hipStream_t)stream;
	const int64_t fstride = faces_batch == 1 ? 0 : n_faces * 3;
	hipLaunchKernelGGL(face_areas_kernel, dim3((unsigned)cdiv(n_faces, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, fstride, (int)n_verts, (int)n_faces,
					   (float*)ws);
	hipLaunchKernelGGL(area_scan_kernel, dim3((unsigned)n_meshes), dim3(1024), 0, s, (int)n_faces, (float*)ws);
	hipLaunchKernelGGL(sample_surface_kernel, dim3((unsigned)cdiv(n_samples, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, fstride, (const float*)ws, rnd,
					   (int)n_verts, (int)n_faces, (int)n_samples, face_idx, uv, out, attr, attr_out);
	FIND_LAUNCH_CHECK("sample_surface");
	return FIND_OK;
}

extern "C" int find_sample_surface_again(const float* verts, const int32_t* faces, int64_t faces_batch, const float* rnd, int64_t n_meshes, int64_t n_verts,
										 int64_t n_faces, int64_t n_samples, int32_t* face_idx, float* uv, float* out, const float* attr, float* attr_out,
										 const void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(verts && faces && rnd && face_idx && uv && out && ws, "find_sample_surface_again: NULL argument");
	FIND_REQUIRE((attr == nullptr) == (attr_out == nullptr), "find_sample_surface_again: attr and attr_out must both be given or both NULL");
	FIND_REQUIRE(!bad_dims(n_meshes, n_samples) && !bad_dims(n_meshes, n_faces) && n_verts >= 1, "find_sample_surface_again: bad sizes");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_sample_surface_again: faces_batch must be 1 or n_meshes");
	if (ws_bytes < find_sample_surface_ws_bytes(n_meshes, n_faces)) { set_error("find_sample_surface_again: workspace too small"); return FIND_EWORKSPACE; }
	const int64_t fstride = faces_batch == 1 ? 0 : n_faces * 3;
	hipLaunchKernelGGL(sample_surface_kernel, dim3((unsigned)cdiv(n_samples, 256), (unsigned)n_meshes), dim3(256), 0, (hipStream_t)stream, verts, faces, fstride,
					   (const float*)ws, rnd, (int)n_verts, (int)n_faces, (int)n_samples, face_idx, uv, out, attr, attr_out);
	FIND_LAUNCH_CHECK("sample_surface (again)");
	return FIND_OK;
}

extern "C" int find_masked_mse_fwd(const float* pred, const float* target, int64_t n_pts, float* loss, void* stream) {
	FIND_REQUIRE(pred && target && loss, "find_masked_mse_fwd: NULL argument");
	FIND_REQUIRE(n_pts >= 1 && n_pts < (1ll << 40), "find_masked_mse_fwd: bad sizes");
	hipLaunchKernelGGL(masked_mse_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, target, n_pts, loss);
	FIND_LAUNCH_CHECK("masked_mse_fwd_kernel");
	return FIND_OK;
}

extern "C" int find_masked_mse_bwd(const float* pred, const float* target, int64_t n_pts, const float* g_loss, float* d_pred, void* stream) {
	FIND_REQUIRE(pred && target && g_loss && d_pred, "find_masked_mse_bwd: NULL argument");
	FIND_REQUIRE(n_pts >= 1 && n_pts < (1ll << 40), "find_masked_mse_bwd: bad sizes");
	hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3((unsigned)cdiv(n_pts, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, n_pts, g_loss, d_pred);
	FIND_LAUNCH_CHECK("masked_mse_bwd_kernel");
	return FIND_OK;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }   // (NULL counts as aligned)

extern "C" int64_t find_image_mse_ws_bytes(void) { return IMSE_BLOCKS * (int64_t)sizeof(float); }

extern "C" int find_image_mse_fwd(const float* a, const float* a_mask, const float* b, const float* b_mask, int64_t n_pix, int64_t channels, float* loss,
								  void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(a && b && loss && ws, "find_image_mse_fwd: NULL argument");
	FIND_REQUIRE(n_pix >= 1 && n_pix < (1ll << 40) && channels >= 1 && channels <= 16, "find_image_mse_fwd: bad sizes");
	if (ws_bytes < find_image_mse_ws_bytes()) { set_error("find_image_mse_fwd: workspace too small"); return FIND_EWORKSPACE; }
	const int nblk = (int)std::min<int64_t>(IMSE_BLOCKS, cdiv(n_pix, 1024));
	hipStream_t s = (hipStream_t)stream;
	const bool vec = (n_pix & 3) == 0 && (channels == 3 || channels == 1) && al16(a) && al16(b) && al16(a_mask) && al16(b_mask);
	if (vec && channels == 3) hipLaunchKernelGGL((image_mse_fwd_kernel<3, true>), dim3((unsigned)nblk), dim3(1024), 0, s, a, a_mask, b, b_mask, n_pix, 3, (float*)ws);
	else if (vec) hipLaunchKernelGGL((image_mse_fwd_kernel<1, true>), dim3((unsigned)nblk), dim3(1024), 0, s, a, a_mask, b, b_mask, n_pix, 1, (float*)ws);
	else hipLaunchKernelGGL((image_mse_fwd_kernel<1, false>), dim3((unsigned)nblk), dim3(1024), 0, s, a, a_mask, b, b_mask, n_pix, (int)channels, (float*)ws);
	hipLaunchKernelGGL(image_mse_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, nblk, 1.0f / ((float)n_pix * (float)channels), loss);
	FIND_LAUNCH_CHECK("image_mse_fwd_kernel");
	return FIND_OK;
}

extern "C" int find_image_mse_bwd(const float* a, const float* a_mask, const float* b, const float* b_mask, int64_t n_pix, int64_t channels,
								  const float* g_loss, float* d_a, float* d_a_mask, void* stream) {
	FIND_REQUIRE(a && b && g_loss && (d_a || d_a_mask), "find_image_mse_bwd: NULL argument");
	FIND_REQUIRE(n_pix >= 1 && n_pix < (1ll << 40) && channels >= 1 && channels <= 16, "find_image_mse_bwd: bad sizes");
	FIND_REQUIRE(!d_a_mask || a_mask, "find_image_mse_bwd: d_a_mask without a_mask");
	hipStream_t s = (hipStream_t)stream;
	const bool vec = (n_pix & 3) == 0 && (channels == 3 || channels == 1) && al16(a) && al16(b) && al16(a_mask) && al16(b_mask) && al16(d_a) && al16(d_a_mask);
	if (vec && channels == 3)
		hipLaunchKernelGGL((image_mse_bwd_kernel<3, true>), dim3((unsigned)cdiv(n_pix >> 2, 256)), dim3(256), 0, s, a, a_mask, b, b_mask, n_pix, 3, g_loss, d_a, d_a_mask);
	else if (vec)
		hipLaunchKernelGGL((image_mse_bwd_kernel<1, true>), dim3((unsigned)cdiv(n_pix >> 2, 256)), dim3(256), 0, s, a, a_mask, b, b_mask, n_pix, 1, g_loss, d_a, d_a_mask);
	else
		hipLaunchKernelGGL((image_mse_bwd_kernel<1, false>), dim3((unsigned)cdiv(n_pix, 256)), dim3(256), 0, s, a, a_mask, b, b_mask, n_pix, (int)channels, g_loss, d_a,
						   d_a_mask);
	FIND_LAUNCH_CHECK("image_mse_bwd_kernel");
	return FIND_OK;
}

extern "C" int64_t find_smooth_ws_bytes(int64_t n_meshes, int64_t n_verts, int64_t n_faces) {
	if (bad_dims(n_meshes, n_verts) || n_faces < 1) return -1;
	SmoothWs w;
	carve_smooth(n_meshes, n_verts, n_faces, nullptr, &w);
	return w.bytes;
}

static int smooth_fwd_body(const char* who, const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items, const int32_t* nbr_off,
						   const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_edges, float* loss_edge, float* loss_lap,
						   float w_edge, float w_lap, float* loss_sum, void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(verts && faces && vf_off && vf_items && nbr_off && nbr_idx && ws, "%s: NULL argument", who);
	FIND_REQUIRE(!bad_dims(n_meshes, n_verts) && n_faces >= 1 && n_edges >= 1, "%s: bad sizes", who);
	SmoothWs w;
	carve_smooth(n_meshes, n_verts, n_faces, ws, &w);
	if (ws_bytes < w.bytes) { set_error("%s: workspace too small", who); return FIND_EWORKSPACE; }
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(cot_weights_kernel, dim3((unsigned)cdiv(n_faces, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, (int)n_verts, (int)n_faces, w.fw);
	hipLaunchKernelGGL(smooth_fwd_kernel, dim3((unsigned)w.nblk, (unsigned)n_meshes), dim3(256), 0, s, verts, faces, w.fw, vf_off, vf_items, nbr_off,
					   nbr_idx, (int)n_verts, (int)n_faces, w.nw, w.lapdir, w.partial);
	hipLaunchKernelGGL(smooth_finalize_kernel, dim3(1), dim3(64), 0, s, w.partial, (int)n_meshes, w.nblk, (int)n_verts, (int)n_edges, loss_edge, loss_lap,
					   w_edge, w_lap, loss_sum);
	FIND_LAUNCH_CHECK(who);
	return FIND_OK;
}

static int smooth_bwd_body(const char* who, const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items, const int32_t* nbr_off,
						   const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_edges, const float* g_edge, const float* g_lap,
						   float s_edge, float s_lap, void* ws, int64_t ws_bytes, float* d_verts, void* stream) {
	FIND_REQUIRE(verts && faces && vf_off && vf_items && nbr_off && nbr_idx && g_edge && g_lap && ws && d_verts, "%s: NULL argument", who);
	FIND_REQUIRE(!bad_dims(n_meshes, n_verts) && n_faces >= 1 && n_edges >= 1, "%s: bad sizes", who);
	SmoothWs w;
	carve_smooth(n_meshes, n_verts, n_faces, ws, &w);
	if (ws_bytes < w.bytes) { set_error("%s: workspace too small", who); return FIND_EWORKSPACE; }
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(smooth_bwd_kernel, dim3((unsigned)cdiv(n_verts, SMOOTH_VPB), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, w.fw, vf_off, vf_items, nbr_off, nbr_idx,
					   w.nw, w.lapdir, g_edge, g_lap, s_edge, s_lap, (int)n_meshes, (int)n_verts, (int)n_faces, (int)n_edges, d_verts);
	FIND_LAUNCH_CHECK(who);
	return FIND_OK;
}

extern "C" int find_smooth_fwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
							   const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
							   int64_t n_edges, float* loss_edge, float* loss_lap, void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(loss_edge && loss_lap, "find_smooth_fwd: NULL argument");
	return smooth_fwd_body("find_smooth_fwd", verts, faces, vf_off, vf_items, nbr_off, nbr_idx, n_meshes, n_verts, n_faces, n_edges, loss_edge, loss_lap, 0.f, 0.f,
						   nullptr, ws, ws_bytes, stream);
}

extern "C" int find_smooth_bwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
							   const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
							   int64_t n_edges, const float* g_edge, const float* g_lap, void* ws, int64_t ws_bytes, float* d_verts,
							   void* stream) {
	return smooth_bwd_body("find_smooth_bwd", verts, faces, vf_off, vf_items, nbr_off, nbr_idx, n_meshes, n_verts, n_faces, n_edges, g_edge, g_lap, 1.f, 1.f, ws,
						   ws_bytes, d_verts, stream);
}

extern "C" int find_smooth_loss_fwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items, const int32_t* nbr_off,
									const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_edges, float w_edge, float w_lap,
									float* loss, void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(loss, "find_smooth_loss_fwd: NULL argument");
	return smooth_fwd_body("find_smooth_loss_fwd", verts, faces, vf_off, vf_items, nbr_off, nbr_idx, n_meshes, n_verts, n_faces, n_edges, nullptr, nullptr, w_edge,
						   w_lap, loss, ws, ws_bytes, stream);
}

extern "C" int find_smooth_loss_bwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items, const int32_t* nbr_off,
									const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_edges, float w_edge, float w_lap,
									const float* g_loss, void* ws, int64_t ws_bytes, float* d_verts, void* stream) {
	return smooth_bwd_body("find_smooth_loss_bwd", verts, faces, vf_off, vf_items, nbr_off, nbr_idx, n_meshes, n_verts, n_faces, n_edges, g_loss, g_loss, w_edge,
						   w_lap, ws, ws_bytes, d_verts, stream);
}
