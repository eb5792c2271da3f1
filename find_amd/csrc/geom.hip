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

// ------------------------------------------------------------------------------------------- nearest neighbour
// A lane owns NQ query points; the four waves of a block own the SAME 64*NQ queries and each scans one quarter of every
// target tile (targets stream through LDS in tiles of NN_TILE, x,y,z,pad float4, a wave-uniform ds_read_b128 broadcasts one
// target to all lanes), so that FIND's small clouds (5-10 k points x 16 feet) still put more than two waves on every SIMD.
// Strict '<' on ascending j keeps the lowest index on ties inside a wave; the four partial results are merged through LDS by
// (distance, index), which is the same rule.
constexpr int NN_TILE = 1024;
constexpr int NQ = 2;

__global__ __launch_bounds__(256) void nn_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ x_len,
													  const float* __restrict__ y, const int32_t* __restrict__ y_len, int p1_max,
													  int p2_max, float* __restrict__ dist, int32_t* __restrict__ idx) {
	__shared__ float4 ty[NN_TILE];
	__shared__ float pbest[4][64 * NQ];
	__shared__ int pidx[4][64 * NQ];
	const int n = blockIdx.y;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int p1 = x_len ? x_len[n] : p1_max;
	const int p2 = y_len ? y_len[n] : p2_max;
	const float* xp = x + (int64_t)n * p1_max * 3;
	const float* yp = y + (int64_t)n * p2_max * 3;
	float qx[NQ], qy[NQ], qz[NQ], best[NQ];
	int bi[NQ], qi[NQ];
#pragma unroll
	for (int k = 0; k < NQ; ++k) {
		qi[k] = (blockIdx.x * NQ + k) * 64 + lane;
		const int i = min(qi[k], p1_max - 1);
		qx[k] = xp[i * 3 + 0]; qy[k] = xp[i * 3 + 1]; qz[k] = xp[i * 3 + 2];
		best[k] = INFINITY; bi[k] = -1;
	}
	for (int j0 = 0; j0 < p2; j0 += NN_TILE) {
		const int cnt = min(NN_TILE, p2 - j0);
		__syncthreads();
		for (int j = threadIdx.x; j < cnt; j += 256) {
			const float* s = yp + (int64_t)(j0 + j) * 3;
			ty[j] = make_float4(s[0], s[1], s[2], 0.f);
		}
		__syncthreads();
		const int ja = wave * (NN_TILE / 4), jb = min(cnt, ja + NN_TILE / 4);
#pragma unroll 4
		for (int j = ja; j < jb; ++j) {
			const float4 t = ty[j];
#pragma unroll
			for (int k = 0; k < NQ; ++k) {
				const float dx = qx[k] - t.x, dy = qy[k] - t.y, dz = qz[k] - t.z;
				const float d = dx * dx + dy * dy + dz * dz;
				if (d < best[k]) { best[k] = d; bi[k] = j0 + j; }
			}
		}
	}
#pragma unroll
	for (int k = 0; k < NQ; ++k) { pbest[wave][k * 64 + lane] = best[k]; pidx[wave][k * 64 + lane] = bi[k]; }
	__syncthreads();
	if (wave == 0) {
#pragma unroll
		for (int k = 0; k < NQ; ++k) {
			float b = best[k];
			int ib = bi[k];
#pragma unroll
			for (int w = 1; w < 4; ++w) {
				const float ob = pbest[w][k * 64 + lane];
				const int oi = pidx[w][k * 64 + lane];
				if (oi >= 0 && (ib < 0 || ob < b || (ob == b && oi < ib))) { b = ob; ib = oi; }
			}
			if (qi[k] < p1_max) {
				const bool valid = qi[k] < p1 && p2 > 0;
				dist[(int64_t)n * p1_max + qi[k]] = valid ? b : 0.f;
				idx[(int64_t)n * p1_max + qi[k]] = valid ? ib : -1;
			}
		}
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
__device__ __forceinline__ void apply_L(const float* __restrict__ q, const int32_t* __restrict__ faces, const float* __restrict__ fw,
										const int32_t* __restrict__ vf_off, const int32_t* __restrict__ vf_items, int i,
										float3* r, float* rowsum) {
	float3 acc = make_float3(0.f, 0.f, 0.f);
	float rs = 0.f;
	for (int e = vf_off[i]; e < vf_off[i + 1]; ++e) {
		const int item = vf_items[e];
		const int f = item / 3, c = item - f * 3;
		const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
		const int j1 = faces[f * 3 + c1], j2 = faces[f * 3 + c2];
		const float w1 = fw[f * 3 + c2], w2 = fw[f * 3 + c1];
		const float3 a = ld3(q + 3 * j1), b = ld3(q + 3 * j2);
		acc.x += w1 * a.x + w2 * b.x; acc.y += w1 * a.y + w2 * b.y; acc.z += w1 * a.z + w2 * b.z;
		rs += w1 + w2;
	}
	*r = acc;
	*rowsum = rs;
}

// forward per vertex: lap = (L V)_i * nw_i - V_i; block partial sums of |lap| and of the half edge-length sums.
// Saves nw_i (rowsum>0 ? 1/rowsum : rowsum) and u_i = lap_i/|lap_i| scaled later in backward.
__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
														  const float* __restrict__ fw, const int32_t* __restrict__ vf_off,
														  const int32_t* __restrict__ vf_items, const int32_t* __restrict__ nbr_off,
														  const int32_t* __restrict__ nbr_idx, int n_verts, int n_faces,
														  float* __restrict__ nw_out, float* __restrict__ lapdir_out,
														  float* __restrict__ partial /* [n_meshes][gridDim.x][2] */) {
	const int m = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const float* vp = verts + (int64_t)m * n_verts * 3;
	float lap_n = 0.f, edge_s = 0.f;
	if (i < n_verts) {
		float3 r;
		float rs;
		apply_L(vp, faces, fw + (int64_t)m * n_faces * 3, vf_off, vf_items, i, &r, &rs);
		const float nw = rs > 0.f ? 1.0f / rs : rs;
		const float3 v = ld3(vp + 3 * i);
		const float3 lap = make_float3(r.x * nw - v.x, r.y * nw - v.y, r.z * nw - v.z);
		lap_n = norm3(lap);
		const float inv = lap_n > 0.f ? 1.0f / lap_n : 0.f;
		const int64_t o = (int64_t)m * n_verts + i;
		nw_out[o] = nw;
		lapdir_out[o * 3 + 0] = lap.x * inv; lapdir_out[o * 3 + 1] = lap.y * inv; lapdir_out[o * 3 + 2] = lap.z * inv;
		for (int e = nbr_off[i]; e < nbr_off[i + 1]; ++e) {
			const float3 d = sub3(v, ld3(vp + 3 * nbr_idx[e]));
			edge_s += d.x * d.x + d.y * d.y + d.z * d.z;  // every undirected edge is visited from both ends
		}
	}
	__shared__ float red[2][4];
	const float a = wave_sum(lap_n), b = wave_sum(edge_s);
	if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
	__syncthreads();
	if (threadIdx.x == 0) {
		float* p = partial + ((int64_t)m * gridDim.x + blockIdx.x) * 2;
		p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
		p[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
	}
}

__global__ void smooth_finalize_kernel(const float* __restrict__ partial, int n_meshes, int nblk, int n_verts, int n_edges,
									   float* __restrict__ loss_edge, float* __restrict__ loss_lap) {
	// single wave; deterministic order
	float lap = 0.f, edge = 0.f;
	for (int k = threadIdx.x; k < n_meshes * nblk; k += 64) { lap += partial[k * 2 + 0]; edge += partial[k * 2 + 1]; }
	lap = wave_sum(lap);
	edge = wave_sum(edge);
	if (threadIdx.x == 0) {
		*loss_lap = lap / (float)n_verts / (float)n_meshes;
		*loss_edge = 0.5f * edge / (float)n_edges / (float)n_meshes;
	}
}

// backward: with u_i = g_lap/(V N) * lapdir_i and q_i = nw_i * u_i:   dV_i = (L q)_i - u_i  +  g_edge/(E N) * 2 * sum_j (v_i - v_j)
__global__ void smooth_bwd_q_kernel(const float* __restrict__ nw, const float* __restrict__ lapdir, const float* __restrict__ g_lap,
									int n_meshes, int n_verts, float* __restrict__ q) {
	const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (o >= (int64_t)n_meshes * n_verts) return;
	const float s = (*g_lap) / (float)n_verts / (float)n_meshes * nw[o];
	q[o * 3 + 0] = s * lapdir[o * 3 + 0]; q[o * 3 + 1] = s * lapdir[o * 3 + 1]; q[o * 3 + 2] = s * lapdir[o * 3 + 2];
}

__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
														  const float* __restrict__ fw, const int32_t* __restrict__ vf_off,
														  const int32_t* __restrict__ vf_items, const int32_t* __restrict__ nbr_off,
														  const int32_t* __restrict__ nbr_idx, const float* __restrict__ q,
														  const float* __restrict__ lapdir, const float* __restrict__ g_edge,
														  const float* __restrict__ g_lap, int n_meshes, int n_verts, int n_faces,
														  int n_edges, float* __restrict__ d_verts) {
	const int m = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_verts) return;
	const float* vp = verts + (int64_t)m * n_verts * 3;
	const float* qp = q + (int64_t)m * n_verts * 3;
	float3 r;
	float rs;
	apply_L(qp, faces, fw + (int64_t)m * n_faces * 3, vf_off, vf_items, i, &r, &rs);
	const int64_t o = (int64_t)m * n_verts + i;
	const float su = (*g_lap) / (float)n_verts / (float)n_meshes;
	float3 g = make_float3(r.x - su * lapdir[o * 3 + 0], r.y - su * lapdir[o * 3 + 1], r.z - su * lapdir[o * 3 + 2]);
	const float se = 2.0f * (*g_edge) / (float)n_edges / (float)n_meshes;
	const float3 v = ld3(vp + 3 * i);
	float3 es = make_float3(0.f, 0.f, 0.f);
	for (int e = nbr_off[i]; e < nbr_off[i + 1]; ++e) {
		const float3 d = sub3(v, ld3(vp + 3 * nbr_idx[e]));
		es.x += d.x; es.y += d.y; es.z += d.z;
	}
	d_verts[o * 3 + 0] = g.x + se * es.x;
	d_verts[o * 3 + 1] = g.y + se * es.y;
	d_verts[o * 3 + 2] = g.z + se * es.z;
}

struct SmoothWs {
	float* fw;       // (n_meshes, n_faces, 3)
	float* nw;       // (n_meshes, n_verts)
	float* lapdir;   // (n_meshes, n_verts, 3)
	float* q;        // (n_meshes, n_verts, 3)
	float* partial;  // (n_meshes, nblk, 2)
	int nblk;
	int64_t bytes;
};

static void carve_smooth(int64_t n_meshes, int64_t n_verts, int64_t n_faces, void* ws, SmoothWs* o) {
	Carver c(ws);
	o->nblk = (int)cdiv(n_verts, 256);
	o->fw = c.take<float>(n_meshes * n_faces * 3);
	o->nw = c.take<float>(n_meshes * n_verts);
	o->lapdir = c.take<float>(n_meshes * n_verts * 3);
	o->q = c.take<float>(n_meshes * n_verts * 3);
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

extern "C" int find_nn_fwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n, int64_t p1_max,
						   int64_t p2_max, float* dist, int32_t* idx, void* stream) {
	FIND_REQUIRE(x && y && dist && idx, "find_nn_fwd: NULL argument");
	FIND_REQUIRE(!bad_dims(n, p1_max) && p2_max >= 1 && p2_max < (1ll << 30), "find_nn_fwd: bad sizes");
	hipLaunchKernelGGL(nn_fwd_kernel, dim3((unsigned)cdiv(p1_max, 64 * NQ), (unsigned)n), dim3(256), 0, (hipStream_t)stream, x, x_len, y, y_len,
					   (int)p1_max, (int)p2_max, dist, idx);
	FIND_LAUNCH_CHECK("nn_fwd_kernel");
	return FIND_OK;
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

extern "C" int64_t find_smooth_ws_bytes(int64_t n_meshes, int64_t n_verts, int64_t n_faces) {
	if (bad_dims(n_meshes, n_verts) || n_faces < 1) return -1;
	SmoothWs w;
	carve_smooth(n_meshes, n_verts, n_faces, nullptr, &w);
	return w.bytes;
}

extern "C" int find_smooth_fwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
							   const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
							   int64_t n_edges, float* loss_edge, float* loss_lap, void* ws, int64_t ws_bytes, void* stream) {
	FIND_REQUIRE(verts && faces && vf_off && vf_items && nbr_off && nbr_idx && loss_edge && loss_lap && ws, "find_smooth_fwd: NULL argument");
	FIND_REQUIRE(!bad_dims(n_meshes, n_verts) && n_faces >= 1 && n_edges >= 1, "find_smooth_fwd: bad sizes");
	SmoothWs w;
	carve_smooth(n_meshes, n_verts, n_faces, ws, &w);
	if (ws_bytes < w.bytes) { set_error("find_smooth_fwd: workspace too small"); return FIND_EWORKSPACE; }
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(cot_weights_kernel, dim3((unsigned)cdiv(n_faces, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, (int)n_verts, (int)n_faces, w.fw);
	hipLaunchKernelGGL(smooth_fwd_kernel, dim3((unsigned)w.nblk, (unsigned)n_meshes), dim3(256), 0, s, verts, faces, w.fw, vf_off, vf_items, nbr_off,
					   nbr_idx, (int)n_verts, (int)n_faces, w.nw, w.lapdir, w.partial);
	hipLaunchKernelGGL(smooth_finalize_kernel, dim3(1), dim3(64), 0, s, w.partial, (int)n_meshes, w.nblk, (int)n_verts, (int)n_edges, loss_edge, loss_lap);
	FIND_LAUNCH_CHECK("smooth_fwd");
	return FIND_OK;
}

extern "C" int find_smooth_bwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
							   const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
							   int64_t n_edges, const float* g_edge, const float* g_lap, void* ws, int64_t ws_bytes, float* d_verts,
							   void* stream) {
	FIND_REQUIRE(verts && faces && vf_off && vf_items && nbr_off && nbr_idx && g_edge && g_lap && ws && d_verts, "find_smooth_bwd: NULL argument");
	FIND_REQUIRE(!bad_dims(n_meshes, n_verts) && n_faces >= 1 && n_edges >= 1, "find_smooth_bwd: bad sizes");
	SmoothWs w;
	carve_smooth(n_meshes, n_verts, n_faces, ws, &w);
	if (ws_bytes < w.bytes) { set_error("find_smooth_bwd: workspace too small"); return FIND_EWORKSPACE; }
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(smooth_bwd_q_kernel, dim3((unsigned)cdiv(n_meshes * n_verts, 256)), dim3(256), 0, s, w.nw, w.lapdir, g_lap, (int)n_meshes, (int)n_verts, w.q);
	hipLaunchKernelGGL(smooth_bwd_kernel, dim3((unsigned)w.nblk, (unsigned)n_meshes), dim3(256), 0, s, verts, faces, w.fw, vf_off, vf_items, nbr_off, nbr_idx,
					   w.q, w.lapdir, g_edge, g_lap, (int)n_meshes, (int)n_verts, (int)n_faces, (int)n_edges, d_verts);
	FIND_LAUNCH_CHECK("smooth_bwd");
	return FIND_OK;
}
