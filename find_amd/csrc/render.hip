// Differentiable mesh render for FIND on gfx950: projection, tile rasteriser fused with the soft-silhouette and
// Phong/softmax-blend shaders, and the backward passes.
//
// Replaces FootRenderer.forward / rasterize (reference src/model/renderer.py:208-245, 247-383): PyTorch3D's
// FoVPerspectiveCameras transform, rasterize_meshes (K=100 soft-silhouette pass + K=1 RGB pass), SoftSilhouetteShader,
// SoftPhongShader + softmax_rgb_blend (math restated in-repo at renderer.py:23-72).
//
// Design (HBM/VALU-bound, no MFMA): the K=100 fragment buffers PyTorch3D materialises (2.8 kB/pixel) never exist.
//   1. project_kernel   : world -> (x_ndc, y_ndc, z_view) per (image, vertex)                       [coalesced stream]
//   2. face_setup_kernel: per (image, face) cull + 48-B face record + tile-bbox packed in one uint32
//   3. raster_tile_kernel: persistent workgroups take 16x16-pixel tiles from an atomic counter (covered tiles cost far more
//      than empty ones).  The packed bboxes are scanned 256 faces at a time (4 B/face, L2-resident), hits are
//      ballot-compacted IN FACE ORDER into LDS (deterministic), their records staged in LDS; each wave owns an 8x8-pixel
//      quadrant and evaluates only the faces whose blurred bbox touches it; pixels keep the silhouette product and the
//      nearest inside fragment in registers.  Silhouette candidates are also appended to a per-workgroup scratch so that a
//      pixel with more than faces_per_pixel of them keeps exactly the K nearest in depth (lane-parallel bisection; the
//      K-th depth is saved for the backward).  The tail shades the nearest fragment (Phong + blend), writes mask / image.
//   4. backward: silhouette gradient is FACE-centric (one thread walks the blurred bbox of its face and accumulates the
//      six NDC gradients in registers: no per-fragment atomics); projection backward is a deterministic sum over views.
// Conventions: SURVEY.md Appendix A.2-A.4 (row-vector transforms, NDC +x left / +y up, image = mesh*n_views + view).
#include <type_traits>

#include "common.h"

namespace find {
int g_raster_ablate = 0;
namespace render {

constexpr int TS_WG = 16;       // tile edge in pixels: one workgroup per tile, a quadrant per wave (raster_tile_kernel<4>) ...
constexpr int TS_WAVE = 8;      // ... or one wave per tile (raster_tile_kernel<1>: find_debug_raster_ablate bit 128, measured slower)
constexpr float KEPS = 1e-8f;
constexpr uint32_t TB_EMPTY = 0x000000FFu;  // tx0 = 255 > tx1 = 0
constexpr int KU = 16;            // list entries in flight per lane in the K-nearest passes (they are L2-latency-bound)
constexpr int KN_CAP = 4096;      // silhouette candidates kept per pixel for the K-nearest rule (more: unresolved, flags[1]); the pole of a 10 002-vertex lat-long scan @256^2 collects ~2000
constexpr int RING = 8;           // candidates a lane collects in LDS before it writes them out: 2 x 32 contiguous bytes per flush
constexpr int RASTER_WGS = 1024;  // persistent rasteriser workgroups (4 per CU: 95 VGPRs, 38 KB LDS); each owns 256 x KN_CAP x 8 B of scratch

struct Ws {
	float* vproj;     // (n_img, V, 3)
	float4* frec;     // (n_img, F, 3) float4: [x0 y0 x1 y1][x2 y2 z0 z1][z2 - - -]
	uint32_t* tb;     // (n_img, F) packed tile bbox
	uint32_t* tbb;    // (n_img, ceil(F / 64)) packed tile bbox of every run of 64 consecutive faces
	float* normals;   // (n_meshes, V, 3) area-weighted vertex normals (world)
	int32_t* p2f;     // (n_img, H, W) nearest inside face (local id) or -1   [saved for backward]
	float* bary;      // (n_img, H, W, 3) its perspective-correct barycentrics
	float* d_vproj;   // (n_img, V, 3) backward accumulator
	float* d_normals; // (n_meshes, V, 3) backward accumulator
	float* raw_normals; // (n_meshes, V, 3) un-normalised vertex-normal sums (backward)
	int32_t* flags;   // [0] straddling faces seen, [1] overflow pixels left unresolved (> KN_CAP candidates), [3] tile counter
	float* zthr;      // (n_img, H, W) depth of the K-th nearest silhouette candidate (+inf: every candidate counts)  [backward]
	float* alpha;     // (n_img, H, W) prod (1 - p_k) over the blended candidates  [backward: 1 - mask has lost it wherever the mask rounds to 1]
	float2* scratch;  // (raster workgroups, 2, 256, KN_CAP) per-pixel candidate lists of the tile in flight: depths, then 1 - p; one contiguous run per pixel
	int32_t* tile_any; // (n_img, tiles) 1 = some face's blurred bbox touches the tile
	int32_t* tile_cnt; // (n_img, tiles) estimate of how many do (sampled)
	int32_t* tile_order; // (n_img * tiles) tile ids, the tiles with the most faces first
	int64_t raster_wgs;
	int64_t bytes;
};

static void carve(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t V, int64_t F, void* ws, Ws* o) {
	Carver c(ws);
	const int64_t n_img = n_meshes * n_views;
	const int64_t px = n_img * rp->image_h * rp->image_w;
	o->flags = c.take<int32_t>(64);
	o->vproj = c.take<float>(n_img * V * 3);
	o->frec = c.take<float4>(n_img * F * 3);
	o->tb = c.take<uint32_t>(n_img * F);
	o->tbb = c.take<uint32_t>(n_img * cdiv(F, 64));
	o->normals = c.take<float>(n_meshes * V * 3);
	o->p2f = c.take<int32_t>(px);
	o->bary = c.take<float>(px * 3);
	o->d_vproj = c.take<float>(n_img * V * 3);
	o->d_normals = c.take<float>(n_meshes * V * 3);
	o->raw_normals = c.take<float>(n_meshes * V * 3);
	o->zthr = c.take<float>(px);
	o->alpha = c.take<float>(px);
	const int64_t tiles = n_img * cdiv(rp->image_w, TS_WAVE) * cdiv(rp->image_h, TS_WAVE);   // (the finer of the two tilings)
	o->raster_wgs = std::min<int64_t>(tiles, RASTER_WGS);
	o->scratch = c.take<float2>(o->raster_wgs * KN_CAP * 256);
	o->tile_any = c.take<int32_t>(2 * tiles);
	o->tile_cnt = o->tile_any + tiles;
	o->tile_order = c.take<int32_t>(tiles);
	o->bytes = c.off;
}

// ------------------------------------------------------------------------------------------------ 1. projection
__global__ void project_kernel(const float* __restrict__ verts, const float* __restrict__ R, const float* __restrict__ T, float s,
							   int n_views, int V, float* __restrict__ vproj) {
	const int img = blockIdx.y;
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= V) return;
	const int mesh = img / n_views, view = img - mesh * n_views;
	const float* Rm = R + view * 9;
	const float* Tm = T + view * 3;
	const float* p = verts + ((int64_t)mesh * V + v) * 3;
	const float x = p[0] * Rm[0] + p[1] * Rm[3] + p[2] * Rm[6] + Tm[0];
	const float y = p[0] * Rm[1] + p[1] * Rm[4] + p[2] * Rm[7] + Tm[1];
	const float z = p[0] * Rm[2] + p[1] * Rm[5] + p[2] * Rm[8] + Tm[2];
	float* o = vproj + ((int64_t)img * V + v) * 3;
	o[0] = s * x / z;
	o[1] = s * y / z;
	o[2] = z;
}

// d_vproj (dx_ndc, dy_ndc, dz_view) -> d_verts (world), summed over the views of each mesh (deterministic, no atomics)
__global__ void project_bwd_kernel(const float* __restrict__ verts, const float* __restrict__ R, const float* __restrict__ T, float s,
								   int n_views, int V, const float* __restrict__ d_vproj, float* __restrict__ d_verts, int accumulate) {
	const int mesh = blockIdx.y;
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= V) return;
	const float* p = verts + ((int64_t)mesh * V + v) * 3;
	float gx = 0.f, gy = 0.f, gz = 0.f;
	for (int m = 0; m < n_views; ++m) {
		const float* Rm = R + m * 9;
		const float* Tm = T + m * 3;
		const float x = p[0] * Rm[0] + p[1] * Rm[3] + p[2] * Rm[6] + Tm[0];
		const float y = p[0] * Rm[1] + p[1] * Rm[4] + p[2] * Rm[7] + Tm[1];
		const float z = p[0] * Rm[2] + p[1] * Rm[5] + p[2] * Rm[8] + Tm[2];
		const float* g = d_vproj + (((int64_t)mesh * n_views + m) * V + v) * 3;
		const float iz = 1.0f / z;
		const float dvx = g[0] * s * iz, dvy = g[1] * s * iz;
		const float dvz = g[2] - (g[0] * s * x + g[1] * s * y) * iz * iz;
		gx += dvx * Rm[0] + dvy * Rm[1] + dvz * Rm[2];
		gy += dvx * Rm[3] + dvy * Rm[4] + dvz * Rm[5];
		gz += dvx * Rm[6] + dvy * Rm[7] + dvz * Rm[8];
	}
	float* o = d_verts + ((int64_t)mesh * V + v) * 3;
	if (accumulate) { o[0] += gx; o[1] += gy; o[2] += gz; }
	else { o[0] = gx; o[1] = gy; o[2] = gz; }
}

// ------------------------------------------------------------------------------------------------ 2. face setup
__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
	return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// pixel index range [lo, hi] whose centres 1-(2i+1)/S may fall in NDC [cmin, cmax] (one pixel of slack; exact test per pixel)
__device__ __forceinline__ void pix_range(float cmin, float cmax, int S, int* lo, int* hi) {
	const float a = ((1.0f - cmax) * S - 1.0f) * 0.5f;
	const float b = ((1.0f - cmin) * S - 1.0f) * 0.5f;
	*lo = max((int)floorf(a) - 1, 0);
	*hi = min((int)ceilf(b) + 1, S - 1);
}

__global__ void face_setup_kernel(const float* __restrict__ vproj, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
								  int n_views, int V, int F, int H, int W, float blur_radius, float z_clip,
								  float4* __restrict__ frec, uint32_t* __restrict__ tb, uint32_t* __restrict__ tbb, int32_t* __restrict__ flags,
								  int32_t* __restrict__ tile_any, int32_t* __restrict__ tile_cnt, int tiles_x, int tiles_per_img, int TS) {
	const int img = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	const int mesh = img / n_views;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)min(f, F - 1) * 3;
	uint32_t packed = TB_EMPTY;
	const int64_t o = (int64_t)img * F + f;
	if (f < F && fp[0] >= 0) {
		const float* vp = vproj + (int64_t)img * V * 3;
		const float x0 = vp[3 * fp[0]], y0 = vp[3 * fp[0] + 1], z0 = vp[3 * fp[0] + 2];
		const float x1 = vp[3 * fp[1]], y1 = vp[3 * fp[1] + 1], z1 = vp[3 * fp[1] + 2];
		const float x2 = vp[3 * fp[2]], y2 = vp[3 * fp[2] + 1], z2 = vp[3 * fp[2] + 2];
		frec[o * 3 + 0] = make_float4(x0, y0, x1, y1);
		frec[o * 3 + 1] = make_float4(x2, y2, z0, z1);
		frec[o * 3 + 2] = make_float4(z2, 0.f, 0.f, 0.f);
		const bool all_behind = z0 < z_clip && z1 < z_clip && z2 < z_clip;
		const bool any_behind = z0 < z_clip || z1 < z_clip || z2 < z_clip;
		if (any_behind && !all_behind) atomicAdd(&flags[0], 1);  // straddling the clip plane: PyTorch3D would clip it (clip.py)
		const float zmax = fmaxf(z0, fmaxf(z1, z2));
		const float area = edge_fn(x0, y0, x1, y1, x2, y2);
		const bool degenerate = area <= KEPS && area >= -KEPS;
		if (!all_behind && !(zmax < 0.f) && !degenerate) {
			const float br = sqrtf(blur_radius);
			int xlo, xhi, ylo, yhi;
			pix_range(fminf(x0, fminf(x1, x2)) - br, fmaxf(x0, fmaxf(x1, x2)) + br, W, &xlo, &xhi);
			pix_range(fminf(y0, fminf(y1, y2)) - br, fmaxf(y0, fmaxf(y1, y2)) + br, H, &ylo, &yhi);
			if (xlo <= xhi && ylo <= yhi) {
				packed = (uint32_t)(xlo / TS) | ((uint32_t)(xhi / TS) << 8) | ((uint32_t)(ylo / TS) << 16) | ((uint32_t)(yhi / TS) << 24);
				// count the faces of every tile this face can touch: the rasteriser skips the scan of a tile with none (most of an image)
				// and takes the crowded tiles first (tile_order_kernel)
				for (int ty = ylo / TS; ty <= yhi / TS; ++ty)
					for (int tx = xlo / TS; tx <= xhi / TS; ++tx) {
						const int64_t ti = (int64_t)img * tiles_per_img + ty * tiles_x + tx;
						if (tile_any[ti] == 0) tile_any[ti] = 1;  // benign race: every writer stores 1
						// how MANY faces touch the tile is estimated from every 16th face (a cost estimate for the tile order: an atomic
						// per face and tile serialises on the crowded tiles and made this kernel 20x slower)
						if ((f & 15) == 0) atomicAdd(tile_cnt + ti, 16);
					}
			}
		}
	}
	if (f < F) tb[o] = packed;
	// the tile bbox of the wave's 64 consecutive faces: the rasteriser tests these first and only opens the runs that reach its tile
	// (faces that are neighbours in the index are neighbours on the surface in any mesh that was not shuffled on purpose)
	int bx0 = packed & 255, bx1 = (packed >> 8) & 255, by0 = (packed >> 16) & 255, by1 = packed >> 24;
	if (packed == TB_EMPTY) { bx0 = 255; bx1 = 0; by0 = 255; by1 = 0; }
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		bx0 = min(bx0, __shfl_xor(bx0, d, 64)); bx1 = max(bx1, __shfl_xor(bx1, d, 64));
		by0 = min(by0, __shfl_xor(by0, d, 64)); by1 = max(by1, __shfl_xor(by1, d, 64));
	}
	if ((threadIdx.x & 63) == 0 && f < F)
		tbb[(int64_t)img * ((F + 63) / 64) + f / 64] = bx0 > bx1 ? TB_EMPTY : ((uint32_t)bx0 | ((uint32_t)bx1 << 8) | ((uint32_t)by0 << 16) | ((uint32_t)by1 << 24));
}


// ------------------------------------------------------------------------------------------------ shared fragment math
struct FaceRec {  // staged in LDS, one per candidate (80 bytes)
	float x0, y0, x1, y1, x2, y2, z0, z1, z2;
	float xmin, xmax, ymin, ymax;  // blurred NDC bbox
	float inv_area;
	float il01, il02, il12;        // 1 / |edge|^2 (0 for a degenerate edge: the projection parameter is then 1, as PyTorch3D)
	int f;
	int pad[2];
};

__device__ __forceinline__ float edge_inv_len2(float ax, float ay, float bx, float by) {
	const float dx = bx - ax, dy = by - ay;
	const float l2 = dx * dx + dy * dy;
	return l2 > KEPS ? 1.0f / l2 : 0.0f;
}

// Everything that is per face (not per pixel) is computed here, once: the per-pixel fragment math below then has no IEEE
// division left (a v_rcp_f32 for the two normalisations).
__device__ __forceinline__ void make_rec(const float4* fr, int f, float br, FaceRec* r) {
	const float4 a = fr[0], b = fr[1], c = fr[2];
	r->x0 = a.x; r->y0 = a.y; r->x1 = a.z; r->y1 = a.w; r->x2 = b.x; r->y2 = b.y; r->z0 = b.z; r->z1 = b.w; r->z2 = c.x;
	r->xmin = fminf(a.x, fminf(a.z, b.x)) - br; r->xmax = fmaxf(a.x, fmaxf(a.z, b.x)) + br;
	r->ymin = fminf(a.y, fminf(a.w, b.y)) - br; r->ymax = fmaxf(a.y, fmaxf(a.w, b.y)) + br;
	r->inv_area = 1.0f / (edge_fn(b.x, b.y, a.x, a.y, a.z, a.w) + KEPS);
	r->il01 = edge_inv_len2(a.x, a.y, a.z, a.w);
	r->il02 = edge_inv_len2(a.x, a.y, b.x, b.y);
	r->il12 = edge_inv_len2(a.z, a.w, b.x, b.y);
	r->f = f;
	r->pad[0] = r->pad[1] = 0;
}

// squared distance to segment ab; also returns the clamped parameter (PointLineDistanceForward); il = 1/|ab|^2 or 0
__device__ __forceinline__ float seg_dist(float px, float py, float ax, float ay, float bx, float by, float il, float* t_out) {
	const float bax = bx - ax, bay = by - ay;
	float t = 1.0f;
	if (il > 0.f) t = fminf(fmaxf((bax * (px - ax) + bay * (py - ay)) * il, 0.f), 1.f);
	const float qx = ax + t * bax - px, qy = ay + t * bay - py;
	*t_out = t;
	return qx * qx + qy * qy;
}

struct Frag {
	float w0, w1, w2;     // perspective-correct barycentrics, unclipped
	float pz_clip;        // depth from clipped barycentrics (silhouette pass)
	float pz;             // depth from unclipped barycentrics (RGB pass)
	float dist;           // unsigned squared distance to the triangle outline
	bool inside;
	int edge;             // nearest edge: 0 = v0v1, 1 = v0v2, 2 = v1v2
	float t;              // its clamped projection parameter
};

// geometry_utils: barycentric, perspective correction, clip, depth, point-triangle distance.  Returns false when the
// pixel is outside the blurred bbox.
__device__ __forceinline__ bool eval_frag(const FaceRec& r, float px, float py, Frag* o) {
	if (px > r.xmax || px < r.xmin || py > r.ymax || py < r.ymin) return false;
	float w0 = edge_fn(px, py, r.x1, r.y1, r.x2, r.y2) * r.inv_area;
	float w1 = edge_fn(px, py, r.x2, r.y2, r.x0, r.y0) * r.inv_area;
	float w2 = edge_fn(px, py, r.x0, r.y0, r.x1, r.y1) * r.inv_area;
	const float t0 = w0 * r.z1 * r.z2, t1 = r.z0 * w1 * r.z2, t2 = r.z0 * r.z1 * w2;
	const float iden = __builtin_amdgcn_rcpf(fmaxf(t0 + t1 + t2, KEPS));
	w0 = t0 * iden; w1 = t1 * iden; w2 = t2 * iden;
	o->w0 = w0; o->w1 = w1; o->w2 = w2;
	o->inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
	float c0 = fmaxf(w0, 0.f), c1 = fmaxf(w1, 0.f), c2 = fmaxf(w2, 0.f);
	const float isum = __builtin_amdgcn_rcpf(fmaxf(c0 + c1 + c2, 1e-5f));
	c0 *= isum; c1 *= isum; c2 *= isum;
	o->pz_clip = c0 * r.z0 + c1 * r.z1 + c2 * r.z2;
	o->pz = w0 * r.z0 + w1 * r.z1 + w2 * r.z2;
	float ta, tb, tc;
	const float e01 = seg_dist(px, py, r.x0, r.y0, r.x1, r.y1, r.il01, &ta);
	const float e02 = seg_dist(px, py, r.x0, r.y0, r.x2, r.y2, r.il02, &tb);
	const float e12 = seg_dist(px, py, r.x1, r.y1, r.x2, r.y2, r.il12, &tc);
	if (e01 <= e02 && e01 <= e12) { o->dist = e01; o->edge = 0; o->t = ta; }
	else if (e02 <= e01 && e02 <= e12) { o->dist = e02; o->edge = 1; o->t = tb; }
	else { o->dist = e12; o->edge = 2; o->t = tc; }
	return true;
}

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
	const float l = fmaxf(sqrtf(x * x + y * y + z * z), 1e-6f);
	x /= l; y /= l; z /= l;
}

// ------------------------------------------------------------------------------------------------ vertex normals
__global__ void normals_scatter_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
									   int V, int F, float* __restrict__ normals) {
	const int mesh = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= F) return;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)f * 3;
	if (fp[0] < 0) return;
	const float* vp = verts + (int64_t)mesh * V * 3;
	const float* a = vp + 3 * fp[0]; const float* b = vp + 3 * fp[1]; const float* c = vp + 3 * fp[2];
	const float ux = c[0] - b[0], uy = c[1] - b[1], uz = c[2] - b[2];
	const float wx = a[0] - b[0], wy = a[1] - b[1], wz = a[2] - b[2];
	const float nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
	float* np_ = normals + (int64_t)mesh * V * 3;
	for (int k = 0; k < 3; ++k) {
		atomicAdd(np_ + 3 * fp[k] + 0, nx); atomicAdd(np_ + 3 * fp[k] + 1, ny); atomicAdd(np_ + 3 * fp[k] + 2, nz);
	}
}

__global__ void normals_normalize_kernel(float* __restrict__ normals, int64_t n) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float x = normals[i * 3], y = normals[i * 3 + 1], z = normals[i * 3 + 2];
	normalize3(x, y, z);
	normals[i * 3] = x; normals[i * 3 + 1] = y; normals[i * 3 + 2] = z;
}

// ------------------------------------------------------------------------------------------------ tile order
// The persistent rasteriser takes tiles from a queue, and a tile's cost grows with the faces that touch it (a pole of the mesh:
// > 1000 candidates per pixel; most of an image: none).  Taken in image order, the expensive tiles of the last images arrive when
// nothing is left to overlap them with -- the average wave was alive for 63 % of the kernel (SQ_WAVE_CYCLES / waves / duration).
// Longest-processing-time-first: bucket the tiles by log2(face count), crowded buckets first.  One workgroup, counting sort.
constexpr int ORDER_BUCKETS = 24;
__global__ __launch_bounds__(1024) void tile_order_kernel(const int32_t* __restrict__ tile_any, const int32_t* __restrict__ tile_cnt, int total,
														   int32_t* __restrict__ order) {
	__shared__ int hist[ORDER_BUCKETS];
	__shared__ int start[ORDER_BUCKETS];
	const int tid = threadIdx.x;
	if (tid < ORDER_BUCKETS) hist[tid] = 0;
	__syncthreads();
	// crowded -> small index; touched tiles whose faces all escaped the sample sit just in front of the empty ones
	auto bucket = [&](int t) {
		if (tile_any[t] == 0) return ORDER_BUCKETS - 1;
		const int c = tile_cnt[t];
		return c <= 0 ? ORDER_BUCKETS - 2 : max(0, ORDER_BUCKETS - 3 - (31 - __builtin_clz(c)));
	};
	for (int t = tid; t < total; t += 1024) atomicAdd(&hist[bucket(t)], 1);
	__syncthreads();
	if (tid == 0) {
		int acc = 0;
		for (int b = 0; b < ORDER_BUCKETS; ++b) { start[b] = acc; acc += hist[b]; }
	}
	__syncthreads();
	for (int t = tid; t < total; t += 1024) order[atomicAdd(&start[bucket(t)], 1)] = t;  // order inside a bucket does not matter
}

// ------------------------------------------------------------------------------------------------ 3. tile rasteriser
struct TileArgs {
	find_render_params rp;
	const float4* frec;
	const uint32_t* tb;
	const uint32_t* tbb;     // per run of 64 faces
	const int32_t* faces;
	int64_t faces_mesh_stride;
	const float* verts;      // world (n_meshes,V,3)
	const float* normals;    // world
	const float* colors;     // (n_meshes,V,3) or null
	const float* cam;        // (n_views,3) camera centres
	int n_views, V, F, tiles_x;
	float* mask;             // (n_img,H,W) or null
	float* image;            // (n_img,H,W,3) or null
	int32_t* p2f_out;        // user-visible packed ids or null
	float* zbuf_out;         // or null
	int32_t* p2f_ws;         // local ids for backward
	float* bary_ws;
	int32_t* flags;
	float* zthr;
	float* alpha_ws;
	float2* scratch;
	const int32_t* tile_any;
	const int32_t* tile_order;
	int tiles_per_img, total_tiles;
	int ablate;              // profiling only (find_debug_raster_ablate): 1 no candidate lists, 2 no K-nearest pass, 4 no fragment math
};

// Persistent: workgroups take (image, tile) pairs from an atomic counter.  Silhouette candidates of every pixel are also
// appended, in face order, to the workgroup's scratch (depth, 1 - p): a pixel that ends with more than faces_per_pixel
// candidates is resolved at the end of its tile by its own wave -- rank by depth, ties to the earlier face (PyTorch3D's
// insertion into the per-pixel K-buffer), blend the K nearest, record the K-th depth for the backward pass.
//
// NW = waves that share a tile.  NW = 4 (what find_render_fwd launches): a 16x16 tile per workgroup, one 8x8 quadrant per wave, the tile's
// face list and records built together (workgroup barriers between the phases).  NW = 1 (find_debug_raster_ablate bit 128): every wave is
// its own worker on 8x8 tiles -- own list, own records, own queue slot, no workgroup barrier anywhere.  The reason to try it: with NW = 4
// the waves of a tile meet at a barrier after every batch of 256 faces, and the longest quadrant loop of a batch is 1.79 x the mean of the
// four at C3 (1.43 x per whole tile; diagnostic counters [26] - [28]).  Measured (same faces per pixel, values equal to rounding): 2.03 against 1.78 ms at 256^2,
// 6.27 against 4.25 ms at 512^2 -- four times the tiles means four times the scans and record stagings, each a chain of dependent loads
// that a lone wave cannot overlap with anything, and that costs more than the barriers did.  Kept as a switch, not as the default.
template <int NW>
__global__ __launch_bounds__(256) void raster_tile_kernel(const TileArgs a) {
	constexpr int TSZ = NW == 4 ? 16 : 8;   // tile edge
	constexpr int GB = NW * 64;             // threads of a group = faces per batch
	constexpr int NG = 4 / NW;              // groups per workgroup
	constexpr int LPR = 4;                  // face-bbox loads a lane keeps in flight per scan round (12 for NW = 1: slower still)
	__shared__ int list[NG][2 * GB];
	__shared__ FaceRec rec[NG][GB];
	__shared__ int wcount2[NG][2][4];
	__shared__ int runs[NG][GB];
	__shared__ int wit[4];
	__shared__ int s_tile[NG];
	// write-combining rings of the candidate lists: [slot][thread], so that a wave's appends (different slots per lane) never conflict
	__shared__ float ring_z[RING][256];
	__shared__ float ring_q[RING][256];

	const int H = a.rp.image_h, W = a.rp.image_w;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = NW == 4 ? 0 : wave, wg = NW == 4 ? wave : 0, gtid = NW == 4 ? tid : lane;   // group, wave inside it, thread inside it
	// barrier of the group: the workgroup's, or -- one wave: its LDS operations execute in order -- only a fence for the compiler
	auto gsync = [&]() {
		if constexpr (NW == 4) {
			__syncthreads();
		} else {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	};
	auto wtotal = [&](int par) {
		int t = 0;
#pragma unroll
		for (int w = 0; w < NW; ++w) t += wcount2[g][par][w];
		return t;
	};
	const float blur = a.rp.sil_blur_radius, br = sqrtf(blur);
	const float inv_sigma = 1.0f / a.rp.sil_sigma;
	const int K = a.rp.sil_faces_per_pixel;
	const bool want_sil = a.mask != nullptr;
	const bool want_rgb = a.image != nullptr || a.p2f_out != nullptr || a.zbuf_out != nullptr;
	// Candidate lists: every pixel (thread) owns ONE contiguous run of KN_CAP depths and one of KN_CAP (1 - p) values.  A lane collects
	// RING candidates in LDS and writes them out as whole 32-byte pieces (two 16-byte stores per array): the lists reach memory as
	// full sectors, once.  (First version: entry i of all 256 pixels interleaved, one 4-byte store per candidate -- lanes of a wave sit
	// at different i, so every store dirtied its own line: 5.3 GB written and 2.9 GB read per C3 launch for 0.5 GB of list data.)
	float* const scr_z = reinterpret_cast<float*>(a.scratch + (int64_t)blockIdx.x * KN_CAP * 256) + tid * KN_CAP;
	float* const scr_q = scr_z + KN_CAP * 256;
	auto flush_ring = [&](int base, int valid) {   // ring -> list entries [base, base + RING); slots >= valid become (+inf, 1): never selected
		float zv[RING], qv[RING];
#pragma unroll
		for (int u = 0; u < RING; ++u) {
			zv[u] = u < valid ? ring_z[u][tid] : INFINITY;
			qv[u] = u < valid ? ring_q[u][tid] : 1.0f;
		}
#pragma unroll
		for (int u = 0; u < RING; u += 4) {
			*reinterpret_cast<float4*>(scr_z + base + u) = make_float4(zv[u], zv[u + 1], zv[u + 2], zv[u + 3]);
			*reinterpret_cast<float4*>(scr_q + base + u) = make_float4(qv[u], qv[u + 1], qv[u + 2], qv[u + 3]);
		}
	};

	for (;;) {
		if (gtid == 0) s_tile[g] = atomicAdd(&a.flags[3], 1);
		gsync();
		const int t_q = s_tile[g];
		gsync();
		if (t_q >= a.total_tiles) break;
		const int t_id = a.tile_order[t_q];
		const int img = t_id / a.tiles_per_img, tile = t_id - img * a.tiles_per_img;
		const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
		// wave w owns the 8x8-pixel quadrant (w&1, w>>1) of the tile: faces are culled per wave against that quadrant
		const int qx0 = tile_x * TSZ + (wg & 1) * 8, qy0 = tile_y * TSZ + (wg >> 1) * 8;
		const int xi = qx0 + (lane & 7), yi = qy0 + (lane >> 3);
		const bool in_img = xi < W && yi < H;
		const float px = 1.0f - (2.0f * xi + 1.0f) / (float)W;
		const float py = 1.0f - (2.0f * yi + 1.0f) / (float)H;
		// NDC extent of the quadrant's pixel centres (x and y decrease with the pixel index)
		const float q_xhi = 1.0f - (2.0f * qx0 + 1.0f) / (float)W, q_xlo = 1.0f - (2.0f * (qx0 + 7) + 1.0f) / (float)W;
		const float q_yhi = 1.0f - (2.0f * qy0 + 1.0f) / (float)H, q_ylo = 1.0f - (2.0f * (qy0 + 7) + 1.0f) / (float)H;

		float alpha = 1.0f, z_lo = INFINITY, z_hi = 0.0f;
		int cnt = 0;
		int n_eval = 0, n_eval_prev = 0;   // diagnostics (ablate bit 64): (pixel, face) tests this lane ran
		float bz = INFINITY, bd = 0.f, bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
		int bf = -1;

		const uint32_t* tbp = a.tb + (int64_t)img * a.F;
		const float4* frp = a.frec + (int64_t)img * a.F * 3;

		auto shade_batch = [&](int nb) {
			// stage the records of list[0 .. nb) in LDS
			if (gtid < nb) make_rec(frp + (int64_t)list[g][gtid] * 3, list[g][gtid], br, &rec[g][gtid]);
			gsync();
			// candidates of this wave's quadrant: bbox-vs-quadrant test by 64 lanes at a time, then a scalar loop over the set
			// bits (increasing k: the order of the alpha product and of the candidate lists is that of the face list).
			// (Measured and dropped in round 2: 16-lane groups owning 4x4-pixel blocks, each walking its own mask of faces in the same
			// instruction stream -- 45 % of the lane-level tests produce a candidate instead of 26 %, 141 M tests instead of 245 M at C3, and
			// the render is no faster, 1.87 against 1.80 ms: the loop runs for the longest of the four masks, and in the tiles that decide the
			// kernel's duration -- the rim, the poles -- that is the quadrant's whole list.)
			for (int kb = 0; kb < nb; kb += 64) {
				bool ov = false;
				if (kb + lane < nb) {
					const FaceRec& rr = rec[g][kb + lane];
					ov = !(q_xlo > rr.xmax || q_xhi < rr.xmin || q_ylo > rr.ymax || q_yhi < rr.ymin);
				}
				unsigned long long qm = __ballot(ov);
				while (qm) {
					const int k = kb + __builtin_ctzll(qm);
					qm &= qm - 1;
					Frag fr;
					if (a.ablate & 64) ++n_eval;
					if ((a.ablate & 4) || !in_img || !eval_frag(rec[g][k], px, py, &fr)) continue;
					if (want_sil && fr.pz_clip >= 0.f && (fr.inside || fr.dist < blur)) {
						const float sd = fr.inside ? -fr.dist : fr.dist;
						const float prob = 1.0f / (1.0f + __expf(sd * inv_sigma));
						alpha *= (1.0f - prob);
						if (cnt < KN_CAP && !(a.ablate & 1)) {
							const int slot = cnt & (RING - 1);
							ring_z[slot][tid] = fr.pz_clip; ring_q[slot][tid] = 1.0f - prob;
							if (slot == RING - 1) flush_ring(cnt - (RING - 1), RING);
						}
						++cnt;
						z_lo = fminf(z_lo, fr.pz_clip); z_hi = fmaxf(z_hi, fr.pz_clip);  // depth range of the candidates (bisection bounds)
					}
					if (want_rgb && fr.inside && fr.pz >= 0.f && fr.pz < bz) {
						bz = fr.pz; bf = rec[g][k].f; bd = -fr.dist; bw0 = fr.w0; bw1 = fr.w1; bw2 = fr.w2;
					}
				}
			}
			gsync();
			if (NW == 4 && (a.ablate & 64)) {   // diagnostics [28]: sum over BATCHES of the longest quadrant loop (the barrier above waits for it)
				if (lane == 0) wit[wave] = n_eval - n_eval_prev;
				n_eval_prev = n_eval;
				__syncthreads();
				if (tid == 0) atomicAdd(&a.flags[28], max(max(wit[0], wit[1]), max(wit[2], wit[3])));
				__syncthreads();
			}
		};

		const int nF = a.tile_any[t_id] ? a.F : 0;  // nothing can touch this tile: straight to the background write
		// Two-level scan of the packed tile bboxes.  Level 1: the bboxes of the runs of 64 consecutive faces (face_setup_kernel), 256 runs
		// per round, compacted in order into `runs`.  Level 2: the faces of the runs that reach this tile, 16 runs = 1024 faces per
		// round: the four loads of a thread are in flight together (the scan is bound by the latency of this load, not by its bytes),
		// then four ordered compactions -- ballot per wave, exclusive offsets across the waves through LDS (one barrier each: the wave
		// counts alternate between two sets), the running length kept in a register.  Runs and faces stay in index order, so the
		// candidate order is the face order whichever runs were skipped.  (One level -- every tile reading every face's bbox -- was
		// 14 rounds per tile at 13 776 faces; a 16 x 16 tile is reached by a tenth of the runs of a mesh with coherent face order.)
		int nl = 0;   // == n_list, replicated in every thread
		int par = 0;
		const int n_runs = (nF + 63) >> 6;
		const uint32_t* tbbp = a.tbb + (int64_t)img * ((a.F + 63) >> 6);
		auto tile_hit = [&](uint32_t t) {
			const int tx0 = t & 255, tx1 = (t >> 8) & 255, ty0 = (t >> 16) & 255, ty1 = t >> 24;
			return tile_x >= tx0 && tile_x <= tx1 && tile_y >= ty0 && tile_y <= ty1;  // (TB_EMPTY: tx0 = 255 > tx1 = 0)
		};
		for (int rbase = 0; rbase < n_runs; rbase += GB) {
			int n_hit;
			{
				const bool rh = rbase + gtid < n_runs && tile_hit(tbbp[rbase + gtid]);
				const unsigned long long m = __ballot(rh);
				if (lane == 0) wcount2[g][par][wg] = __popcll(m);
				gsync();
				int off = 0;
				for (int w = 0; w < wg; ++w) off += wcount2[g][par][w];
				if (rh) runs[g][off + __popcll(m & ((1ull << lane) - 1ull))] = rbase + gtid;
				n_hit = wtotal(par);
				par ^= 1;
				gsync();
			}
			for (int j0 = 0; j0 < n_hit; j0 += LPR * NW) {
				uint32_t tbv[LPR];
				int fidx[LPR];
#pragma unroll
				for (int r = 0; r < LPR; ++r) {
					const int j = j0 + r * NW + wg;
					fidx[r] = j < n_hit ? runs[g][j] * 64 + lane : nF;
					tbv[r] = fidx[r] < nF ? tbp[fidx[r]] : TB_EMPTY;
				}
#pragma unroll
				for (int r = 0; r < LPR; ++r) {
					if (j0 + r * NW >= n_hit) break;  // uniform
					const bool hit = tile_hit(tbv[r]);
					const unsigned long long m = __ballot(hit);
					if (lane == 0) wcount2[g][par][wg] = __popcll(m);
					gsync();
					int off = nl;
					for (int w = 0; w < wg; ++w) off += wcount2[g][par][w];
					if (hit) list[g][off + __popcll(m & ((1ull << lane) - 1ull))] = fidx[r];
					nl += wtotal(par);
					par ^= 1;
					if (nl >= GB) {  // uniform
						gsync();
						shade_batch(GB);
						const int rest = nl - GB;
						int moved = 0;
						if (gtid < rest) moved = list[g][GB + gtid];
						gsync();
						if (gtid < rest) list[g][gtid] = moved;
						nl = rest;
						gsync();
					}
				}
			}
			gsync();   // `runs` is rewritten by the next round
		}
		if (nl > 0) { gsync(); shade_batch(nl); }

		// ---- K-nearest rule for the pixels that collected more than K candidates.  Lane-parallel and exact: every such
		// lane finds the K-th smallest depth of its OWN list (16-byte reads of its contiguous run) by a radix search
		// on the integer image of the depth (non-negative floats order like their bit patterns) with one counting pass per
		// step, then blends, in face order, the candidates in front of it and as many of those AT it as still fit.
		float thr = INFINITY;
		if (want_sil) {
			const bool over = in_img && cnt > K && !(a.ablate & 2);
			const unsigned long long ov_all = __ballot(over);
			if (ov_all) {
				// (no fence: every lane reads back only the list it wrote itself -- same thread, same addresses, program order.  An
				// agent-scope fence here costs 0.6 ms per C3 render: L2 write-back + L1 invalidate on a multi-XCD part.)
				const unsigned long long trunc = __ballot(over && cnt > KN_CAP);
				int wave_max_cnt = cnt;  // diagnostics: [5] largest candidate count seen
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) wave_max_cnt = max(wave_max_cnt, __shfl_xor(wave_max_cnt, d, 64));
				if (lane == 0) {  // diagnostics: [4] pixels with more than K candidates; [1] of those, left unresolved
					atomicAdd(&a.flags[4], (int)__popcll(ov_all));
					atomicMax(&a.flags[5], wave_max_cnt);
					if (trunc) atomicAdd(&a.flags[1], (int)__popcll(trunc));
				}
				const long long kt0 = (a.ablate & 64) ? wall_clock64() : 0;
				if ((a.ablate & 64) && lane == 0) {  // profiling: waves by their longest list, [8 + log2 bucket]; [20] sum of longest lists
					atomicAdd(&a.flags[8 + min(7, max(0, 31 - __builtin_clz(max(wave_max_cnt, 1)) - 6))], 1);
					atomicAdd(&a.flags[20], wave_max_cnt);
					atomicAdd(&a.flags[21], (int)__popcll(ov_all));
				}
				if (over && cnt <= KN_CAP) {
					// the candidates still in the ring join the list, padded to a whole piece with (+inf, 1) entries that no pass selects
					if (cnt & (RING - 1)) flush_ring(cnt & ~(RING - 1), cnt & (RING - 1));
					const float4* z4 = reinterpret_cast<const float4*>(scr_z);
					const float4* q4 = reinterpret_cast<const float4*>(scr_q);
					const int n4 = (cnt + 3) >> 2;   // 16-byte pieces to read (the padding of the last one is inert)
					// one pass over the lane's depths, KU entries in flight, fn(bits of the depth)
					auto scan_z = [&](auto&& fn) {
						int i = 0;
						for (; i + KU / 4 <= n4; i += KU / 4) {
							float4 v[KU / 4];
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) v[u] = z4[i + u];
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) {
								fn(__float_as_uint(v[u].x + 0.0f)); fn(__float_as_uint(v[u].y + 0.0f));
								fn(__float_as_uint(v[u].z + 0.0f)); fn(__float_as_uint(v[u].w + 0.0f));
							}
						}
						for (; i < n4; ++i) {
							const float4 v = z4[i];
							fn(__float_as_uint(v.x + 0.0f)); fn(__float_as_uint(v.y + 0.0f)); fn(__float_as_uint(v.z + 0.0f)); fn(__float_as_uint(v.w + 0.0f));
						}
					};
					unsigned lo = __float_as_uint(z_lo + 0.0f), hi = __float_as_uint(z_hi + 0.0f);
					// Radix search for the K-th smallest depth.  Invariant: every candidate lies in [lo, hi] or was counted in c_lo
					// (candidates in front of lo) or lies behind hi; the K-th smallest is inside [lo, hi].  A level histograms
					// (z - lo) >> shift into 32 bins in ONE read of the list and keeps the bin
					// that holds the K-th; once that bin has at most 4 candidates they are fetched and ranked directly.
					int c_lo = 0;
					// the lane's 32 counters (16 bits each) live in LDS, on top of the face records nobody needs any more in this tile:
					// hist[w * GB + gtid], w = bin >> 1.  One fire-and-forget ds_add per candidate instead of sixteen selects.
					unsigned* const hist = reinterpret_cast<unsigned*>(rec[g]) + gtid;
					while (lo < hi) {
						const unsigned span = hi - lo;
						const int shift = span < 32u ? 0 : (27 - __builtin_clz(span));  // (span >> shift) <= 31
#pragma unroll
						for (int w = 0; w < 16; ++w) hist[w * GB] = 0u;
						scan_z([&](unsigned zb) {
							if (zb < lo || zb > hi) return;
							const unsigned bin = (zb - lo) >> shift;
							__hip_atomic_fetch_add(hist + (bin >> 1) * GB, 1u << ((bin & 1u) * 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						});
						// the bin that holds the (K - c_lo)-th candidate of the range
						const int need = K - c_lo;
						int acc = 0, sel = 31, in_sel = 0;
						bool found = false;
#pragma unroll
						for (int w = 0; w < 16; ++w) {
							const unsigned hw = hist[w * GB];
#pragma unroll
							for (int h = 0; h < 2; ++h) {
								const int cb = (int)((hw >> (h * 16)) & 0xFFFFu);
								if (!found) {
									if (acc + cb >= need) { sel = 2 * w + h; in_sel = cb; found = true; }
									else acc += cb;
								}
							}
						}
						c_lo += acc;
						lo = lo + ((unsigned)sel << shift);
						hi = min(hi, lo + ((1u << shift) - 1u));
						if (shift == 0) break;  // a single depth value is left
						if (in_sel <= 4) {
							// fetch the (at most 4) candidates of the bin and take the (K - c_lo)-th smallest of them
							unsigned c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu, c3 = 0xFFFFFFFFu;
							int m = 0;
							scan_z([&](unsigned zb) {
								// (selects, not an if-chain: the compiler turned that into a four-element array in scratch)
								const bool in = zb >= lo && zb <= hi;
								c0 = (in && m == 0) ? zb : c0; c1 = (in && m == 1) ? zb : c1;
								c2 = (in && m == 2) ? zb : c2; c3 = (in && m == 3) ? zb : c3;
								m += in ? 1 : 0;
							});
							// sort the four (absent ones are +max) and index
							unsigned t;
							if (c0 > c1) { t = c0; c0 = c1; c1 = t; }
							if (c2 > c3) { t = c2; c2 = c3; c3 = t; }
							if (c0 > c2) { t = c0; c0 = c2; c2 = t; }
							if (c1 > c3) { t = c1; c1 = c3; c3 = t; }
							if (c1 > c2) { t = c1; c1 = c2; c2 = t; }
							const int rnk = K - c_lo;  // 1-based rank inside the bin
							const unsigned T = rnk == 1 ? c0 : (rnk == 2 ? c1 : (rnk == 3 ? c2 : c3));
							c_lo += (c0 < T) + (c1 < T) + (c2 < T) + (c3 < T);
							lo = hi = T;
							break;
						}
					}
					// lo = bits of the K-th smallest depth, c_lo = candidates strictly in front of it
					int ties = K - c_lo;  // candidates AT the K-th depth that are kept: the earliest ones (PyTorch3D's insertion order)
					float asel = 1.0f;
					unsigned nxt = 0x7F800000u;   // bits of the nearest depth BEHIND the K-th
					bool excess = false;          // more candidates at the K-th depth than are kept
					{
						auto take = [&](float z, float q) {
							const unsigned zb = __float_as_uint(z + 0.0f);
							if (zb < lo) asel *= q;
							else if (zb == lo) {
								if (ties > 0) { asel *= q; --ties; }
								else excess = true;
							} else nxt = min(nxt, zb);
						};
						int i = 0;
						for (; i + KU / 4 <= n4; i += KU / 4) {
							float4 zv[KU / 4], qv[KU / 4];
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) { zv[u] = z4[i + u]; qv[u] = q4[i + u]; }
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) {
								take(zv[u].x, qv[u].x); take(zv[u].y, qv[u].y); take(zv[u].z, qv[u].z); take(zv[u].w, qv[u].w);
							}
						}
						for (; i < n4; ++i) {
							const float4 zv = z4[i], qv = q4[i];
							take(zv.x, qv.x); take(zv.y, qv.y); take(zv.z, qv.z); take(zv.w, qv.w);
						}
					}
					alpha = asel;
					// What the backward compares a candidate's depth with.  It recomputes that depth in another kernel, and the compiler is free
					// to contract the same expression differently there: with the K-th depth itself as the bound, the K-th candidate fell on
					// the wrong side of its own depth in about half of the overflow pixels (2.5e-3 of the gradient's scale on a dense mesh at
					// 64^2).  The bound is the MIDPOINT between the K-th depth and the next one behind it -- any rounding difference smaller
					// than half that gap selects the same K candidates -- and the K-th depth itself only where candidates tied AT it were
					// left out (the backward then takes every tied one: the one case it cannot tell apart without face ids in the lists).
					const float zk = __uint_as_float(lo);
					thr = excess ? zk : 0.5f * (zk + __uint_as_float(nxt));
				}
				if ((a.ablate & 64) && lane == 0) atomicAdd(&a.flags[22], (int)((wall_clock64() - kt0) >> 4));  // K-pass time of the wave, 16-tick units (100 MHz)
			}
		}

		if (a.ablate & 64) {  // diagnostics: [24] (pixel, face) tests issued (64-lane slots, in units of 64), [25] silhouette candidates (units of 64)
			int te = n_eval, tc = in_img ? cnt : 0;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { te += __shfl_xor(te, d, 64); tc += __shfl_xor(tc, d, 64); }
			if (lane == 0 && te) { atomicAdd(&a.flags[24], (te + 32) >> 6); atomicAdd(&a.flags[25], (tc + 32) >> 6); }
			// [26] sum over tiles of the LONGEST of the four quadrants' face loops, [27] sum over tiles of all four: 4 x [26] / [27] is how
			// much longer the tile's waves stay in the fragment phase than their own work needs (the others wait at the barrier)
			if constexpr (NW == 4) {
				if (lane == 0) wit[wave] = n_eval;
				__syncthreads();
				if (tid == 0) {
					atomicAdd(&a.flags[26], max(max(wit[0], wit[1]), max(wit[2], wit[3])));
					atomicAdd(&a.flags[27], wit[0] + wit[1] + wit[2] + wit[3]);
				}
				__syncthreads();
			}
		}
		if (in_img) {
			const int64_t pix = ((int64_t)img * H + yi) * W + xi;
			if (want_sil) {
				a.mask[pix] = 1.0f - alpha;
				a.zthr[pix] = thr;
				a.alpha_ws[pix] = alpha;
			}
			if (want_rgb) {
				if (a.p2f_ws) {
					a.p2f_ws[pix] = bf;
					a.bary_ws[pix * 3 + 0] = bw0; a.bary_ws[pix * 3 + 1] = bw1; a.bary_ws[pix * 3 + 2] = bw2;
				}
				if (a.p2f_out) a.p2f_out[pix] = bf < 0 ? -1 : img * a.F + bf;
				if (a.zbuf_out) a.zbuf_out[pix] = bf < 0 ? -1.0f : bz;
				if (a.image) {
					float* o = a.image + pix * 3;
					if (bf < 0) {
						o[0] = a.rp.background[0]; o[1] = a.rp.background[1]; o[2] = a.rp.background[2];
					} else {
						const int mesh = img / a.n_views, view = img - mesh * a.n_views;
						const int32_t* fp = a.faces + (int64_t)mesh * a.faces_mesh_stride + (int64_t)bf * 3;
						const float bw[3] = {bw0, bw1, bw2};
						float pos[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, tex[3] = {0, 0, 0};
#pragma unroll
						for (int k = 0; k < 3; ++k) {
							const int64_t vo = ((int64_t)mesh * a.V + fp[k]) * 3;
#pragma unroll
							for (int c = 0; c < 3; ++c) {
								pos[c] += bw[k] * a.verts[vo + c];
								nrm[c] += bw[k] * a.normals[vo + c];
								tex[c] += bw[k] * a.colors[vo + c];
							}
						}
						float nx = nrm[0], ny = nrm[1], nz = nrm[2];
						normalize3(nx, ny, nz);
						float lx = a.rp.light_pos[0] - pos[0], ly = a.rp.light_pos[1] - pos[1], lz = a.rp.light_pos[2] - pos[2];
						normalize3(lx, ly, lz);
						const float cosang = nx * lx + ny * ly + nz * lz;
						const float diff = a.rp.diffuse * fmaxf(cosang, 0.f);
						float vx = a.cam[view * 3] - pos[0], vy = a.cam[view * 3 + 1] - pos[1], vz = a.cam[view * 3 + 2] - pos[2];
						normalize3(vx, vy, vz);
						const float rx = -lx + 2.f * cosang * nx, ry = -ly + 2.f * cosang * ny, rz = -lz + 2.f * cosang * nz;
						const float al = fmaxf(vx * rx + vy * ry + vz * rz, 0.f) * (cosang > 0.f ? 1.f : 0.f);
						const float spec = a.rp.specular * powf(al, a.rp.shininess);
						const float eps = 1e-10f;
						const float prob = 1.0f / (1.0f + __expf(bd / a.rp.rgb_sigma));
						const float z_inv = (a.rp.zfar - bz) / (a.rp.zfar - a.rp.znear);
						const float z_inv_max = fmaxf(z_inv, eps);
						const float wnum = prob * __expf((z_inv - z_inv_max) / a.rp.rgb_gamma);
						const float delta = fmaxf(__expf((eps - z_inv_max) / a.rp.rgb_gamma), eps);
						const float den = wnum + delta;
#pragma unroll
						for (int c = 0; c < 3; ++c) {
							const float col = (a.rp.ambient + diff) * tex[c] + spec;
							o[c] = (wnum * col + delta * a.rp.background[c]) / den;
						}
					}
				}
			}
		}
		gsync();  // the next tile reuses list / rec / scratch
	}
}

// ------------------------------------------------------------------------------------------------ 4. silhouette backward
// mask = 1 - prod_k (1 - p_k),  p_k = sigmoid(-d_k / sigma)   =>   d mask / d d_k = -alpha * p_k / sigma, alpha = the product as the forward
// formed it (saved; 1 - mask is zero wherever the mask has rounded to 1, while the true product -- 1e-8, 1e-9 -- times p_k / sigma = 1e4 is
// a gradient autograd through the K fragments does deliver: 2.5e-3 of the gradient's scale on a dense mesh at 64^2).
// LPF lanes per (image, face) (8 for small blurred bboxes, 32 for large ones): they stride over the pixels of the face's blurred bbox (coalesced rows of d_mask / mask / zthr, no
// divergence between faces with different bbox sizes), accumulate the gradients of its three NDC vertices in registers
// (PointTriangleDistanceBackward: nearest edge only, projection parameter treated as constant), butterfly-reduce them, and
// lane 0 issues the six atomics.
template <int LPF>
__global__ __launch_bounds__(256) void sil_bwd_kernel(const find_render_params rp, const float4* __restrict__ frec, const uint32_t* __restrict__ tb,
													   const int32_t* __restrict__ faces, int64_t faces_mesh_stride, int n_views, int V, int F,
													   const float* __restrict__ alpha_ws, const float* __restrict__ d_mask, const float* __restrict__ zthr,
													   float* __restrict__ d_vproj) {
	const int img = blockIdx.y;
	const int sub = threadIdx.x & (LPF - 1);
	const int f = blockIdx.x * (256 / LPF) + threadIdx.x / LPF;
	const int64_t o = (int64_t)img * F + min(f, F - 1);
	const bool act = f < F && tb[o] != TB_EMPTY;
	const int H = rp.image_h, W = rp.image_w;
	const float blur = rp.sil_blur_radius, br = sqrtf(blur);
	FaceRec r;
	make_rec(frec + o * 3, f, br, &r);
	int xlo, xhi, ylo, yhi;
	pix_range(r.xmin, r.xmax, W, &xlo, &xhi);
	pix_range(r.ymin, r.ymax, H, &ylo, &yhi);
	const int bw = xhi - xlo + 1;
	const int npx = act ? max(bw, 0) * max(yhi - ylo + 1, 0) : 0;
	float g0x = 0.f, g0y = 0.f, g1x = 0.f, g1y = 0.f, g2x = 0.f, g2y = 0.f;
	const float inv_sigma = 1.0f / rp.sil_sigma;
	// The three per-pixel values (upstream gradient, K-th depth, mask) are requested together and ONE PIXEL AHEAD of the math: asked for
	// one after the other behind the tests that need them, a lane's ~10 pixels were ~30 memory latencies in a row.  773 -> 730 us at C3;
	// the rest is the fragment math itself (79 M pixel x face evaluations; without the six atomics per face the kernel takes 721 us).
	auto pixel_of = [&](int pi, int* yi, int* xi) -> int64_t {
		const int ry = pi / bw;
		*yi = ylo + ry; *xi = xlo + (pi - ry * bw);
		return ((int64_t)img * H + *yi) * W + *xi;
	};
	int yi = 0, xi = 0;
	float g = 0.f, zt = 0.f, mk = 0.f;
	if (sub < npx) {
		const int64_t pix = pixel_of(sub, &yi, &xi);
		g = d_mask[pix]; zt = zthr[pix]; mk = alpha_ws[pix];
	}
	for (int pi = sub; pi < npx; pi += LPF) {
		const int cy = yi, cx = xi;
		const float cg = g, czt = zt, cmk = mk;
		if (pi + LPF < npx) {
			const int64_t pix = pixel_of(pi + LPF, &yi, &xi);
			g = d_mask[pix]; zt = zthr[pix]; mk = alpha_ws[pix];
		}
		const float py = 1.0f - (2.0f * cy + 1.0f) / (float)H;
		{
			if (cg == 0.f) continue;
			const float px = 1.0f - (2.0f * cx + 1.0f) / (float)W;
			Frag fr;
			if (!eval_frag(r, px, py, &fr)) continue;
			if (!(fr.pz_clip >= 0.f && (fr.inside || fr.dist < blur))) continue;
			if (fr.pz_clip > czt) continue;  // pixel with more than K candidates: this one is not among the K nearest
			const float sd = fr.inside ? -fr.dist : fr.dist;
			const float prob = 1.0f / (1.0f + __expf(sd * inv_sigma));
			const float alpha = cmk;
			// gradient w.r.t. the UNSIGNED distance: sign * dmask/dd
			const float gd = (fr.inside ? -1.0f : 1.0f) * (-cg * alpha * prob * inv_sigma);
			// nearest edge (a,b), q = a + t (b - a):  d = |q - p|^2,  dd/da = 2 (1-t) (q - p),  dd/db = 2 t (q - p)
			float ax, ay, bx, by;
			if (fr.edge == 0) { ax = r.x0; ay = r.y0; bx = r.x1; by = r.y1; }
			else if (fr.edge == 1) { ax = r.x0; ay = r.y0; bx = r.x2; by = r.y2; }
			else { ax = r.x1; ay = r.y1; bx = r.x2; by = r.y2; }
			const float qx = ax + fr.t * (bx - ax) - px, qy = ay + fr.t * (by - ay) - py;
			const float ga = gd * 2.0f * (1.0f - fr.t), gb = gd * 2.0f * fr.t;
			if (fr.edge == 0) { g0x += ga * qx; g0y += ga * qy; g1x += gb * qx; g1y += gb * qy; }
			else if (fr.edge == 1) { g0x += ga * qx; g0y += ga * qy; g2x += gb * qx; g2y += gb * qy; }
			else { g1x += ga * qx; g1y += ga * qy; g2x += gb * qx; g2y += gb * qy; }
		}
	}
#pragma unroll
	for (int d = 1; d < LPF; d <<= 1) {
		g0x += __shfl_xor(g0x, d, 64); g0y += __shfl_xor(g0y, d, 64); g1x += __shfl_xor(g1x, d, 64);
		g1y += __shfl_xor(g1y, d, 64); g2x += __shfl_xor(g2x, d, 64); g2y += __shfl_xor(g2y, d, 64);
	}
	if (!act || sub != 0) return;
	const int mesh = img / n_views;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)f * 3;
	float* dv = d_vproj + (int64_t)img * V * 3;
	if (g0x != 0.f || g0y != 0.f) { atomicAdd(dv + 3 * fp[0], g0x); atomicAdd(dv + 3 * fp[0] + 1, g0y); }
	if (g1x != 0.f || g1y != 0.f) { atomicAdd(dv + 3 * fp[1], g1x); atomicAdd(dv + 3 * fp[1] + 1, g1y); }
	if (g2x != 0.f || g2y != 0.f) { atomicAdd(dv + 3 * fp[2], g2x); atomicAdd(dv + 3 * fp[2] + 1, g2y); }
}

// ------------------------------------------------------------------------------------------------ RGB backward
// Pixel-centric (K = 1): image = (amb + diff) * tex + spec at the nearest inside fragment (the blend weight cancels to
// within 1e-10).  Gradients flow to vertex colours (tex), world vertices (pos), vertex normals (nrm) and,
// through the perspective-correct barycentrics, to the NDC vertices (x, y, z_view).
struct RgbGrad {  // gradients of one face's three vertices: world position, vertex normal, vertex colour, NDC (x, y, z_view)
	float dP[3][3], dN[3][3], dC[3][3], g[9];
};

__device__ __forceinline__ void rgb_grad_zero(RgbGrad& o) {
#pragma unroll
	for (int k = 0; k < 3; ++k)
#pragma unroll
		for (int c = 0; c < 3; ++c) { o.dP[k][c] = 0.f; o.dN[k][c] = 0.f; o.dC[k][c] = 0.f; }
#pragma unroll
	for (int i = 0; i < 9; ++i) o.g[i] = 0.f;
}

__device__ __forceinline__ void rgb_grad_commit(const RgbGrad& o, const int32_t* fp, int64_t mesh, int64_t img, int V, float* d_vproj, float* d_verts,
												float* d_normals, float* d_colors) {
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const int64_t vo = (mesh * V + fp[k]) * 3;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			atomicAdd(d_verts + vo + c, o.dP[k][c]);
			atomicAdd(d_normals + vo + c, o.dN[k][c]);
			if (d_colors) atomicAdd(d_colors + vo + c, o.dC[k][c]);
			atomicAdd(d_vproj + (img * V + fp[k]) * 3 + c, o.g[3 * k + c]);
		}
	}
}

// gradient contribution of ONE pixel whose nearest fragment is face bf (barycentrics bw, upstream gradient g), added into o
__device__ __forceinline__ void rgb_pixel_grad(const find_render_params& rp, const float4* __restrict__ frec, const int32_t* fp, const float* __restrict__ verts,
											   const float* __restrict__ normals, const float* __restrict__ colors, const float* __restrict__ cam, int mesh,
											   int view, int64_t img, int V, int F, int bf, int xi, int yi, const float* g, const float* bw, RgbGrad& o) {
	const int H = rp.image_h, W = rp.image_w;
	float P[3][3], N[3][3], C[3][3];
	float pos[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, tex[3] = {0, 0, 0};
	for (int k = 0; k < 3; ++k) {
		const int64_t vo = ((int64_t)mesh * V + fp[k]) * 3;
		for (int c = 0; c < 3; ++c) {
			P[k][c] = verts[vo + c]; N[k][c] = normals[vo + c]; C[k][c] = colors[vo + c];
			pos[c] += bw[k] * P[k][c]; nrm[c] += bw[k] * N[k][c]; tex[c] += bw[k] * C[k][c];
		}
	}
	// ---- forward recompute
	const float nl = fmaxf(sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]), 1e-6f);
	const float n[3] = {nrm[0] / nl, nrm[1] / nl, nrm[2] / nl};
	float Lv[3] = {rp.light_pos[0] - pos[0], rp.light_pos[1] - pos[1], rp.light_pos[2] - pos[2]};
	const float ll = fmaxf(sqrtf(Lv[0] * Lv[0] + Lv[1] * Lv[1] + Lv[2] * Lv[2]), 1e-6f);
	const float l[3] = {Lv[0] / ll, Lv[1] / ll, Lv[2] / ll};
	const float cosang = n[0] * l[0] + n[1] * l[1] + n[2] * l[2];
	float Vv[3] = {cam[view * 3] - pos[0], cam[view * 3 + 1] - pos[1], cam[view * 3 + 2] - pos[2]};
	const float vl = fmaxf(sqrtf(Vv[0] * Vv[0] + Vv[1] * Vv[1] + Vv[2] * Vv[2]), 1e-6f);
	const float vd[3] = {Vv[0] / vl, Vv[1] / vl, Vv[2] / vl};
	const float r[3] = {-l[0] + 2.f * cosang * n[0], -l[1] + 2.f * cosang * n[1], -l[2] + 2.f * cosang * n[2]};
	const float vr = vd[0] * r[0] + vd[1] * r[1] + vd[2] * r[2];
	const bool lit = cosang > 0.f;
	const float al = (lit && vr > 0.f) ? vr : 0.f;
	const float shade = rp.ambient + rp.diffuse * fmaxf(cosang, 0.f);
	// ---- backward: col_c = shade * tex_c + spec
	float d_tex[3], d_shade = 0.f, d_spec = 0.f;
	for (int c = 0; c < 3; ++c) { d_tex[c] = g[c] * shade; d_shade += g[c] * tex[c]; d_spec += g[c]; }
	float d_cos = lit ? d_shade * rp.diffuse : 0.f;
	const float d_al = (al > 0.f) ? d_spec * rp.specular * rp.shininess * powf(al, rp.shininess - 1.0f) : 0.f;
	// al = vd . r
	float d_vd[3], d_r[3];
	for (int c = 0; c < 3; ++c) { d_vd[c] = d_al * r[c]; d_r[c] = d_al * vd[c]; }
	// r = -l + 2 cos n
	float d_l[3], d_n[3];
	float dot_rn = 0.f;
	for (int c = 0; c < 3; ++c) { d_l[c] = -d_r[c]; d_n[c] = 2.f * cosang * d_r[c]; dot_rn += d_r[c] * n[c]; }
	d_cos += 2.f * dot_rn;
	// cos = n . l
	for (int c = 0; c < 3; ++c) { d_n[c] += d_cos * l[c]; d_l[c] += d_cos * n[c]; }
	// normalisations  x/|x|:  d_x = (d_u - u (u . d_u)) / |x|
	auto unnorm = [](const float* u, const float* du, float len, float* dx) {
		const float dot = u[0] * du[0] + u[1] * du[1] + u[2] * du[2];
		for (int c = 0; c < 3; ++c) dx[c] = (du[c] - u[c] * dot) / len;
	};
	float d_nrm[3], d_Lv[3], d_Vv[3];
	unnorm(n, d_n, nl, d_nrm);
	unnorm(l, d_l, ll, d_Lv);
	unnorm(vd, d_vd, vl, d_Vv);
	float d_pos[3];
	for (int c = 0; c < 3; ++c) d_pos[c] = -d_Lv[c] - d_Vv[c];
	// ---- interpolation: x = sum_k bw_k X_k
	float d_bw[3] = {0, 0, 0};
	for (int k = 0; k < 3; ++k) {
		const int64_t vo = ((int64_t)mesh * V + fp[k]) * 3;
		for (int c = 0; c < 3; ++c) {
			d_bw[k] += d_pos[c] * P[k][c] + d_nrm[c] * N[k][c] + d_tex[c] * C[k][c];
			o.dP[k][c] += bw[k] * d_pos[c];
			o.dN[k][c] += bw[k] * d_nrm[c];
			o.dC[k][c] += bw[k] * d_tex[c];
		}
	}
	// ---- barycentrics -> NDC vertices.  w'_i = t_i / sum t,  t_0 = w0 z1 z2, t_1 = z0 w1 z2, t_2 = z0 z1 w2,
	// w_i = edge_i(p) / area  (BarycentricPerspectiveCorrectionBackward + BarycentricCoordsBackward)
	const float4 fa = frec[((int64_t)img * F + bf) * 3], fb = frec[((int64_t)img * F + bf) * 3 + 1], fc = frec[((int64_t)img * F + bf) * 3 + 2];
	const float x0 = fa.x, y0 = fa.y, x1 = fa.z, y1 = fa.w, x2 = fb.x, y2 = fb.y, z0 = fb.z, z1 = fb.w, z2 = fc.x;
	const float px = 1.0f - (2.0f * xi + 1.0f) / (float)W, py = 1.0f - (2.0f * yi + 1.0f) / (float)H;
	const float area = edge_fn(x2, y2, x0, y0, x1, y1) + KEPS;
	const float e0 = edge_fn(px, py, x1, y1, x2, y2), e1 = edge_fn(px, py, x2, y2, x0, y0), e2 = edge_fn(px, py, x0, y0, x1, y1);
	const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
	const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
	const float den = t0 + t1 + t2;
	if (!(den > KEPS)) return;  // (degenerate fragment: only the interpolation part contributes)
	const float sdb = (d_bw[0] * t0 + d_bw[1] * t1 + d_bw[2] * t2) / den;
	const float d_t0 = (d_bw[0] - sdb) / den, d_t1 = (d_bw[1] - sdb) / den, d_t2 = (d_bw[2] - sdb) / den;
	const float d_w0 = d_t0 * z1 * z2, d_w1 = d_t1 * z0 * z2, d_w2 = d_t2 * z0 * z1;
	const float d_z0 = d_t1 * w1 * z2 + d_t2 * z1 * w2;
	const float d_z1 = d_t0 * w0 * z2 + d_t2 * z0 * w2;
	const float d_z2 = d_t0 * w0 * z1 + d_t1 * z0 * w1;
	// w_i = e_i / area
	const float d_e0 = d_w0 / area, d_e1 = d_w1 / area, d_e2 = d_w2 / area;
	const float d_area = -(d_w0 * w0 + d_w1 * w1 + d_w2 * w2) / area;
	// edge(p,a,b) = (px-ax)(by-ay) - (py-ay)(bx-ax):  d/da = (-(by-ay)+(py-ay), (px-ax)-(bx-ax)) ... written out per vertex
	float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
	auto edge_bwd = [](float qx, float qy, float ax, float ay, float bx, float by, float ge, float& gax, float& gay, float& gbx, float& gby) {
		gax += ge * (-(by - ay) + (qy - ay));
		gay += ge * (-(qx - ax) + (bx - ax));
		gbx += ge * (-(qy - ay));
		gby += ge * (qx - ax);
	};
	edge_bwd(px, py, x1, y1, x2, y2, d_e0, gx1, gy1, gx2, gy2);
	edge_bwd(px, py, x2, y2, x0, y0, d_e1, gx2, gy2, gx0, gy0);
	edge_bwd(px, py, x0, y0, x1, y1, d_e2, gx0, gy0, gx1, gy1);
	// area = edge(v2; v0, v1): the query point is v2 itself
	{
		const float ge = d_area;
		gx2 += ge * (y1 - y0); gy2 += ge * (-(x1 - x0));
		gx0 += ge * (-(y1 - y0) + (y2 - y0)); gy0 += ge * (-(x2 - x0) + (x1 - x0));
		gx1 += ge * (-(y2 - y0)); gy1 += ge * (x2 - x0);
	}
	o.g[0] += gx0; o.g[1] += gy0; o.g[2] += d_z0; o.g[3] += gx1; o.g[4] += gy1; o.g[5] += d_z1; o.g[6] += gx2; o.g[7] += gy2; o.g[8] += d_z2;
}

// Pixel-centric (K = 1): image = (amb + diff) * tex + spec at the nearest inside fragment (the blend weight cancels to
// within 1e-10).  One thread per pixel, 36 atomics per covered pixel: right when a face covers about a pixel.
__global__ __launch_bounds__(256) void rgb_bwd_kernel(const find_render_params rp, const float4* __restrict__ frec,
													   const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
													   const float* __restrict__ verts, const float* __restrict__ normals,
													   const float* __restrict__ colors, const float* __restrict__ cam, int n_views,
													   int V, int F, const int32_t* __restrict__ p2f, const float* __restrict__ bary,
													   const float* __restrict__ d_image, float* __restrict__ d_vproj,
													   float* __restrict__ d_verts, float* __restrict__ d_normals,
													   float* __restrict__ d_colors) {
	const int H = rp.image_h, W = rp.image_w;
	const int img = blockIdx.y;
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= H * W) return;
	const int64_t pix = (int64_t)img * H * W + p;
	const int bf = p2f[pix];
	if (bf < 0) return;
	const float g[3] = {d_image[pix * 3], d_image[pix * 3 + 1], d_image[pix * 3 + 2]};
	if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f) return;
	const int mesh = img / n_views, view = img - mesh * n_views;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)bf * 3;
	const float bw[3] = {bary[pix * 3], bary[pix * 3 + 1], bary[pix * 3 + 2]};
	RgbGrad o;
	rgb_grad_zero(o);
	rgb_pixel_grad(rp, frec, fp, verts, normals, colors, cam, mesh, view, img, V, F, bf, p % W, p / W, g, bw, o);
	rgb_grad_commit(o, fp, mesh, img, V, d_vproj, d_verts, d_normals, d_colors);
}

// Face-centric variant for large images (a visible face covers several pixels): one thread per (image, face) walks the
// face's pixel bbox, sums the contributions of the pixels whose nearest fragment it is, and issues its 36 atomics ONCE.
__global__ __launch_bounds__(256) void rgb_bwd_faces_kernel(const find_render_params rp, const float4* __restrict__ frec, const uint32_t* __restrict__ tb,
															 const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
															 const float* __restrict__ verts, const float* __restrict__ normals,
															 const float* __restrict__ colors, const float* __restrict__ cam, int n_views,
															 int V, int F, const int32_t* __restrict__ p2f, const float* __restrict__ bary,
															 const float* __restrict__ d_image, float* __restrict__ d_vproj,
															 float* __restrict__ d_verts, float* __restrict__ d_normals,
															 float* __restrict__ d_colors) {
	const int H = rp.image_h, W = rp.image_w;
	const int img = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= F) return;
	const int64_t fo = (int64_t)img * F + f;
	if (tb[fo] == TB_EMPTY) return;
	const float4 fa = frec[fo * 3], fb = frec[fo * 3 + 1];
	int xlo, xhi, ylo, yhi;
	pix_range(fminf(fa.x, fminf(fa.z, fb.x)), fmaxf(fa.x, fmaxf(fa.z, fb.x)), W, &xlo, &xhi);
	pix_range(fminf(fa.y, fminf(fa.w, fb.y)), fmaxf(fa.y, fmaxf(fa.w, fb.y)), H, &ylo, &yhi);
	const int mesh = img / n_views, view = img - mesh * n_views;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)f * 3;
	RgbGrad o;
	rgb_grad_zero(o);
	bool any = false;
	for (int yi = ylo; yi <= yhi; ++yi)
		for (int xi = xlo; xi <= xhi; ++xi) {
			const int64_t pix = ((int64_t)img * H + yi) * W + xi;
			if (p2f[pix] != f) continue;
			const float g[3] = {d_image[pix * 3], d_image[pix * 3 + 1], d_image[pix * 3 + 2]};
			if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f) continue;
			const float bw[3] = {bary[pix * 3], bary[pix * 3 + 1], bary[pix * 3 + 2]};
			rgb_pixel_grad(rp, frec, fp, verts, normals, colors, cam, mesh, view, img, V, F, f, xi, yi, g, bw, o);
			any = true;
		}
	if (any) rgb_grad_commit(o, fp, mesh, img, V, d_vproj, d_verts, d_normals, d_colors);
}

// vertex-normal backward: n_v = normalize(sum_f fn_f), fn_f = (v2 - v1) x (v0 - v1).  d_normals holds dL/dn_v.
__global__ void normals_bwd_prepare_kernel(const float* __restrict__ raw, const float* __restrict__ unit, float* __restrict__ d_normals, int64_t n) {
	// in place: dL/d(raw sum) from dL/d(unit normal)
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float rx = raw[i * 3], ry = raw[i * 3 + 1], rz = raw[i * 3 + 2];
	const float len = sqrtf(rx * rx + ry * ry + rz * rz);
	const float ux = unit[i * 3], uy = unit[i * 3 + 1], uz = unit[i * 3 + 2];
	float dx = d_normals[i * 3], dy = d_normals[i * 3 + 1], dz = d_normals[i * 3 + 2];
	if (len > 1e-6f) {
		const float dot = ux * dx + uy * dy + uz * dz;
		dx = (dx - ux * dot) / len; dy = (dy - uy * dot) / len; dz = (dz - uz * dot) / len;
	} else {
		dx /= 1e-6f; dy /= 1e-6f; dz /= 1e-6f;
	}
	d_normals[i * 3] = dx; d_normals[i * 3 + 1] = dy; d_normals[i * 3 + 2] = dz;
}

__global__ void normals_bwd_faces_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
										 int V, int F, const float* __restrict__ d_raw, float* __restrict__ d_verts) {
	const int mesh = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= F) return;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)f * 3;
	if (fp[0] < 0) return;
	const float* vp = verts + (int64_t)mesh * V * 3;
	const float* dr = d_raw + (int64_t)mesh * V * 3;
	// every corner received the same face normal: G = sum of the three vertices' raw-normal gradients
	float G[3];
	for (int c = 0; c < 3; ++c) G[c] = dr[3 * fp[0] + c] + dr[3 * fp[1] + c] + dr[3 * fp[2] + c];
	const float* a = vp + 3 * fp[0]; const float* b = vp + 3 * fp[1]; const float* c3 = vp + 3 * fp[2];
	const float u[3] = {c3[0] - b[0], c3[1] - b[1], c3[2] - b[2]};   // v2 - v1
	const float w[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};      // v0 - v1
	// fn = u x w:  d_u = w x G,  d_w = G x u
	const float du[3] = {w[1] * G[2] - w[2] * G[1], w[2] * G[0] - w[0] * G[2], w[0] * G[1] - w[1] * G[0]};
	const float dw[3] = {G[1] * u[2] - G[2] * u[1], G[2] * u[0] - G[0] * u[2], G[0] * u[1] - G[1] * u[0]};
	float* dv = d_verts + (int64_t)mesh * V * 3;
	for (int c = 0; c < 3; ++c) {
		atomicAdd(dv + 3 * fp[2] + c, du[c]);
		atomicAdd(dv + 3 * fp[0] + c, dw[c]);
		atomicAdd(dv + 3 * fp[1] + c, -du[c] - dw[c]);
	}
}

__global__ void cam_center_kernel(const float* __restrict__ R, const float* __restrict__ T, int n_views, float* __restrict__ cam) {
	const int m = threadIdx.x;
	if (m >= n_views) return;
	const float* Rm = R + m * 9;
	const float* Tm = T + m * 3;
	for (int i = 0; i < 3; ++i) cam[m * 3 + i] = -(Tm[0] * Rm[i * 3 + 0] + Tm[1] * Rm[i * 3 + 1] + Tm[2] * Rm[i * 3 + 2]);
}

}  // namespace render
}  // namespace find

using namespace find;
using namespace find::render;

static int check_params(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t V, int64_t F) {
	FIND_REQUIRE(rp != nullptr, "find_render: params is NULL");
	FIND_REQUIRE(rp->image_h >= 1 && rp->image_w >= 1 && rp->image_h <= 2048 && rp->image_w <= 2048, "find_render: image size out of range (1 .. 2048: tile coordinates are packed in 8 bits)");
	FIND_REQUIRE(n_meshes >= 1 && n_views >= 1 && n_views <= 256 && n_meshes * n_views < 65536, "find_render: bad batch (%lld meshes x %lld views)", (long long)n_meshes, (long long)n_views);
	FIND_REQUIRE(V >= 1 && F >= 1 && V < (1ll << 28) && F < (1ll << 28), "find_render: bad mesh size");
	FIND_REQUIRE(rp->sil_sigma > 0.f && rp->rgb_sigma > 0.f && rp->rgb_gamma > 0.f && rp->zfar > rp->znear, "find_render: bad blend parameters");
	return FIND_OK;
}

extern "C" int64_t find_render_ws_bytes(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t n_verts, int64_t n_faces) {
	if (check_params(rp, n_meshes, n_views, n_verts, n_faces) != FIND_OK) return -1;
	Ws w;
	carve(rp, n_meshes, n_views, n_verts, n_faces, nullptr, &w);
	return w.bytes + 256 * 3 * (int64_t)sizeof(float);  // + camera centres
}

extern "C" int find_render_fwd(const find_render_params* rp, const float* verts, const int32_t* faces, int64_t faces_batch,
							   const float* vert_colors, const float* R, const float* T, int64_t n_meshes, int64_t n_views,
							   int64_t n_verts, int64_t n_faces, float* mask, float* image, int32_t* pix_to_face, float* zbuf,
							   void* ws, int64_t ws_bytes, void* stream) {
	int rc = check_params(rp, n_meshes, n_views, n_verts, n_faces);
	if (rc != FIND_OK) return rc;
	FIND_REQUIRE(verts && faces && R && T && ws, "find_render_fwd: NULL argument");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_render_fwd: faces_batch must be 1 or n_meshes");
	FIND_REQUIRE(mask || image || pix_to_face || zbuf, "find_render_fwd: no output requested");
	FIND_REQUIRE(!image || vert_colors, "find_render_fwd: image requested without vertex colours");
	Ws w;
	carve(rp, n_meshes, n_views, n_verts, n_faces, ws, &w);
	if (ws_bytes < find_render_ws_bytes(rp, n_meshes, n_views, n_verts, n_faces)) { set_error("find_render_fwd: workspace too small"); return FIND_EWORKSPACE; }
	float* cam = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + w.bytes);
	hipStream_t s = (hipStream_t)stream;
	const int64_t n_img = n_meshes * n_views;
	const int V = (int)n_verts, F = (int)n_faces, H = rp->image_h, W = rp->image_w;
	const int64_t fstride = faces_batch == 1 ? 0 : n_faces * 3;
	const float sc = 1.0f / tanf(rp->fov_deg * 3.14159265358979323846f / 180.0f * 0.5f);
	(void)hipMemsetAsync(w.flags, 0, 64 * sizeof(int32_t), s);
	hipLaunchKernelGGL(project_kernel, dim3((unsigned)cdiv(V, 256), (unsigned)n_img), dim3(256), 0, s, verts, R, T, sc, (int)n_views, V, w.vproj);
	// the silhouette's blur margin is a superset of the RGB pass's (blur 0); one scan serves both
	const float blur = mask ? rp->sil_blur_radius : 0.0f;
	const bool per_wg = (find::g_raster_ablate & 128) == 0;   // bit 128: one wave per 8x8 tile (measured slower, see raster_tile_kernel)
	const int TS = per_wg ? TS_WG : TS_WAVE;
	const int tiles_x = (int)cdiv(W, TS), tiles_per_img = tiles_x * (int)cdiv(H, TS);
	(void)hipMemsetAsync(w.tile_any, 0, 2 * n_img * cdiv(W, TS_WAVE) * cdiv(H, TS_WAVE) * sizeof(int32_t), s);   // flags and counts (both arrays, sized for the finer tiling)
	hipLaunchKernelGGL(face_setup_kernel, dim3((unsigned)cdiv(F, 256), (unsigned)n_img), dim3(256), 0, s, w.vproj, faces, fstride, (int)n_views, V, F, H, W,
					   blur, rp->z_clip, w.frec, w.tb, w.tbb, w.flags, w.tile_any, w.tile_cnt, tiles_x, tiles_per_img, TS);
	if (image) {
		(void)hipMemsetAsync(w.normals, 0, n_meshes * n_verts * 3 * sizeof(float), s);
		hipLaunchKernelGGL(normals_scatter_kernel, dim3((unsigned)cdiv(F, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, fstride, V, F, w.normals);
		// keep the raw sums for the backward in d_normals' slot until then? no: recomputed there. Normalise in place.
		hipLaunchKernelGGL(normals_normalize_kernel, dim3((unsigned)cdiv(n_meshes * n_verts, 256)), dim3(256), 0, s, w.normals, n_meshes * n_verts);
		hipLaunchKernelGGL(cam_center_kernel, dim3(1), dim3(256), 0, s, R, T, (int)n_views, cam);
	}
	TileArgs a;
	memset(&a, 0, sizeof(a));
	a.rp = *rp;
	a.frec = w.frec; a.tb = w.tb; a.tbb = w.tbb; a.faces = faces; a.faces_mesh_stride = fstride;
	a.verts = verts; a.normals = w.normals; a.colors = vert_colors; a.cam = cam;
	a.n_views = (int)n_views; a.V = V; a.F = F; a.tiles_x = tiles_x;
	a.mask = mask; a.image = image; a.p2f_out = pix_to_face; a.zbuf_out = zbuf;
	a.p2f_ws = (image || pix_to_face || zbuf) ? w.p2f : nullptr; a.bary_ws = w.bary; a.flags = w.flags;
	a.zthr = w.zthr; a.alpha_ws = w.alpha; a.scratch = w.scratch; a.tile_any = w.tile_any; a.tile_order = w.tile_order;
	a.tiles_per_img = tiles_per_img; a.total_tiles = (int)(a.tiles_per_img * n_img);
	hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, w.tile_any, w.tile_cnt, a.total_tiles, w.tile_order);
	a.ablate = find::g_raster_ablate;
	if (per_wg) hipLaunchKernelGGL(raster_tile_kernel<4>, dim3((unsigned)std::min<int64_t>(a.total_tiles, w.raster_wgs)), dim3(256), 0, s, a);
	else hipLaunchKernelGGL(raster_tile_kernel<1>, dim3((unsigned)std::min<int64_t>(cdiv(a.total_tiles, 4), w.raster_wgs)), dim3(256), 0, s, a);
	FIND_LAUNCH_CHECK("find_render_fwd");
	return FIND_OK;
}

extern "C" int find_render_bwd(const find_render_params* rp, const float* verts, const int32_t* faces, int64_t faces_batch,
							   const float* vert_colors, const float* R, const float* T, int64_t n_meshes, int64_t n_views,
							   int64_t n_verts, int64_t n_faces, const float* mask, const float* d_mask, const float* d_image,
							   float* d_verts, float* d_vert_colors, void* ws, int64_t ws_bytes, void* stream) {
	int rc = check_params(rp, n_meshes, n_views, n_verts, n_faces);
	if (rc != FIND_OK) return rc;
	FIND_REQUIRE(verts && faces && R && T && ws && d_verts, "find_render_bwd: NULL argument");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_meshes, "find_render_bwd: faces_batch must be 1 or n_meshes");
	FIND_REQUIRE(!d_image || vert_colors, "find_render_bwd: d_image given without vertex colours");
	FIND_REQUIRE(!d_mask || mask, "find_render_bwd: d_mask given without the forward mask");
	Ws w;
	carve(rp, n_meshes, n_views, n_verts, n_faces, ws, &w);
	if (ws_bytes < find_render_ws_bytes(rp, n_meshes, n_views, n_verts, n_faces)) { set_error("find_render_bwd: workspace too small"); return FIND_EWORKSPACE; }
	const float* cam = reinterpret_cast<const float*>(reinterpret_cast<char*>(ws) + w.bytes);
	hipStream_t s = (hipStream_t)stream;
	const int64_t n_img = n_meshes * n_views;
	const int V = (int)n_verts, F = (int)n_faces, H = rp->image_h, W = rp->image_w;
	const int64_t fstride = faces_batch == 1 ? 0 : n_faces * 3;
	const float sc = 1.0f / tanf(rp->fov_deg * 3.14159265358979323846f / 180.0f * 0.5f);
	(void)hipMemsetAsync(w.d_vproj, 0, n_img * n_verts * 3 * sizeof(float), s);
	(void)hipMemsetAsync(d_verts, 0, n_meshes * n_verts * 3 * sizeof(float), s);
	if (d_vert_colors) (void)hipMemsetAsync(d_vert_colors, 0, n_meshes * n_verts * 3 * sizeof(float), s);
	if (d_mask) {
		// lanes per face by the size of a typical blurred bbox (a face of ~1 px plus the blur margin on both sides)
		const float side = 2.0f * sqrtf(rp->sil_blur_radius) * 0.5f * (float)std::max(H, W) + 2.0f;
		// (8 lanes per face up to ~12 x 12 pixels, 16 up to ~25 x 25 -- 512^2: 17.5 x 17.5, measured 1 % of the C4 step better than 32 --, 32 above)
		if (side * side > 640.0f)
			hipLaunchKernelGGL(sil_bwd_kernel<32>, dim3((unsigned)cdiv(F, 8), (unsigned)n_img), dim3(256), 0, s, *rp, w.frec, w.tb, faces, fstride, (int)n_views, V, F,
							   w.alpha, d_mask, w.zthr, w.d_vproj);
		else if (side * side > 160.0f)
			hipLaunchKernelGGL(sil_bwd_kernel<16>, dim3((unsigned)cdiv(F, 16), (unsigned)n_img), dim3(256), 0, s, *rp, w.frec, w.tb, faces, fstride, (int)n_views, V, F,
							   w.alpha, d_mask, w.zthr, w.d_vproj);
		else
			hipLaunchKernelGGL(sil_bwd_kernel<8>, dim3((unsigned)cdiv(F, 32), (unsigned)n_img), dim3(256), 0, s, *rp, w.frec, w.tb, faces, fstride, (int)n_views, V, F,
						   w.alpha, d_mask, w.zthr, w.d_vproj);
	}
	if (d_image) {
		(void)hipMemsetAsync(w.d_normals, 0, n_meshes * n_verts * 3 * sizeof(float), s);
		// sum per face before the atomics unless faces vastly outnumber pixels (measured: faster even at about one pixel per visible face)
		if ((int64_t)H * W * 2 >= (int64_t)F)
			hipLaunchKernelGGL(rgb_bwd_faces_kernel, dim3((unsigned)cdiv(F, 256), (unsigned)n_img), dim3(256), 0, s, *rp, w.frec, w.tb, faces, fstride, verts, w.normals,
							   vert_colors, cam, (int)n_views, V, F, w.p2f, w.bary, d_image, w.d_vproj, d_verts, w.d_normals, d_vert_colors);
		else
			hipLaunchKernelGGL(rgb_bwd_kernel, dim3((unsigned)cdiv((int64_t)H * W, 256), (unsigned)n_img), dim3(256), 0, s, *rp, w.frec, faces, fstride, verts,
						   w.normals, vert_colors, cam, (int)n_views, V, F, w.p2f, w.bary, d_image, w.d_vproj, d_verts, w.d_normals, d_vert_colors);
		// unit normals -> raw area-weighted sums -> face cross products -> vertices
		(void)hipMemsetAsync(w.raw_normals, 0, n_meshes * n_verts * 3 * sizeof(float), s);
		hipLaunchKernelGGL(normals_scatter_kernel, dim3((unsigned)cdiv(F, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, fstride, V, F, w.raw_normals);
		hipLaunchKernelGGL(normals_bwd_prepare_kernel, dim3((unsigned)cdiv(n_meshes * n_verts, 256)), dim3(256), 0, s, w.raw_normals, w.normals, w.d_normals, n_meshes * n_verts);
		hipLaunchKernelGGL(normals_bwd_faces_kernel, dim3((unsigned)cdiv(F, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, faces, fstride, V, F, w.d_normals, d_verts);
	}
	hipLaunchKernelGGL(project_bwd_kernel, dim3((unsigned)cdiv(V, 256), (unsigned)n_meshes), dim3(256), 0, s, verts, R, T, sc, (int)n_views, V, w.d_vproj, d_verts, 1);
	FIND_LAUNCH_CHECK("find_render_bwd");
	return FIND_OK;
}

/* Diagnostics of the last forward that used `ws`: flags[0] = faces straddling the z-clip plane (left unclipped),
 * flags[1] = pixels with more than KN_CAP silhouette candidates (the K-nearest rule could not be applied: all of them were
 * blended).  Device->host copy. */
extern "C" int find_render_flags(const void* ws, int32_t* out2, void* stream) {
	FIND_REQUIRE(ws && out2, "find_render_flags: NULL argument");
	hipError_t e = hipMemcpyAsync(out2, ws, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
	if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
	if (e != hipSuccess) { set_error("find_render_flags: %s", hipGetErrorString(e)); return FIND_ELAUNCH; }
	return FIND_OK;
}

// ------------------------------------------------------------------------------------------------ UV textures
// TexturesUV.sample_textures (PyTorch3D, used by the reference for GT scans: src/data/dataset.py:263-271, losses.py:39-43,
// renderer.py:329-346): the UV of a surface point is the barycentric mix of its face's three UV vertices; the map is flipped
// vertically and read with grid_sample(mode='bilinear', align_corners=True, padding_mode='border'), i.e. at
// x = u (W-1), y = (1-v) (H-1), clamped to the border.
namespace find {
namespace render {

__global__ void uv_sample_kernel(const float* __restrict__ maps, int Ht, int Wt, const float* __restrict__ verts_uvs, int Vt,
								 const int32_t* __restrict__ faces_uvs, int64_t faces_mesh_stride, int F, const int32_t* __restrict__ face_idx,
								 const float* __restrict__ bary, int64_t P, int64_t rows_per_map, float* __restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t row = blockIdx.y;
	if (i >= P) return;
	const int64_t o = row * P + i;
	const int f = face_idx[o];
	float* op = out + o * 3;
	if (f < 0 || f >= F) { op[0] = op[1] = op[2] = 0.f; return; }
	const int64_t mesh = row / rows_per_map;
	const int32_t* fu = faces_uvs + mesh * faces_mesh_stride + (int64_t)f * 3;
	const float* uvp = verts_uvs + mesh * (int64_t)Vt * 2;
	float u = 0.f, v = 0.f;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const float w = bary[o * 3 + k];
		u += w * uvp[2 * fu[k]];
		v += w * uvp[2 * fu[k] + 1];
	}
	// grid_sample, align_corners=True, on the vertically flipped map; border padding = clamp of the source coordinate
	float x = u * (float)(Wt - 1), y = (1.0f - v) * (float)(Ht - 1);
	x = fminf(fmaxf(x, 0.f), (float)(Wt - 1));
	y = fminf(fmaxf(y, 0.f), (float)(Ht - 1));
	const int x0 = (int)floorf(x), y0 = (int)floorf(y);
	const int x1 = min(x0 + 1, Wt - 1), y1 = min(y0 + 1, Ht - 1);
	const float tx = x - (float)x0, ty = y - (float)y0;
	const float* mp = maps + mesh * (int64_t)Ht * Wt * 3;
	const float* p00 = mp + ((int64_t)y0 * Wt + x0) * 3;
	const float* p01 = mp + ((int64_t)y0 * Wt + x1) * 3;
	const float* p10 = mp + ((int64_t)y1 * Wt + x0) * 3;
	const float* p11 = mp + ((int64_t)y1 * Wt + x1) * 3;
	const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
#pragma unroll
	for (int c = 0; c < 3; ++c) op[c] = w00 * p00[c] + w01 * p01[c] + w10 * p10[c] + w11 * p11[c];
}

}  // namespace render
}  // namespace find

extern "C" int find_uv_sample(const float* maps, int64_t n_maps, int64_t map_h, int64_t map_w, const float* verts_uvs, int64_t n_uv_verts,
							  const int32_t* faces_uvs, int64_t faces_batch, int64_t n_faces, const int32_t* face_idx, const float* bary,
							  int64_t n_rows, int64_t n_points, float* out, void* stream) {
	FIND_REQUIRE(maps && verts_uvs && faces_uvs && face_idx && bary && out, "find_uv_sample: NULL argument");
	FIND_REQUIRE(n_maps >= 1 && map_h >= 1 && map_w >= 1 && map_h < (1 << 15) && map_w < (1 << 15) && n_uv_verts >= 1 && n_faces >= 1,
				 "find_uv_sample: bad map / topology sizes");
	FIND_REQUIRE(faces_batch == 1 || faces_batch == n_maps, "find_uv_sample: faces_batch must be 1 or n_maps");
	FIND_REQUIRE(n_rows >= 1 && n_rows < 65536 && n_rows % n_maps == 0 && n_points >= 0, "find_uv_sample: n_rows must be a multiple of n_maps (rows of one map are consecutive)");
	if (n_points == 0) return FIND_OK;
	hipLaunchKernelGGL(render::uv_sample_kernel, dim3((unsigned)cdiv(n_points, 256), (unsigned)n_rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), maps,
					   (int)map_h, (int)map_w, verts_uvs, (int)n_uv_verts, faces_uvs, faces_batch == 1 ? 0 : n_faces * 3, (int)n_faces, face_idx, bary, n_points,
					   n_rows / n_maps, out);
	FIND_LAUNCH_CHECK("uv_sample_kernel");
	return FIND_OK;
}

/* Nearest-fragment buffers of the forward that used `ws`: local face id (-1 = background) and perspective-correct
 * barycentrics per pixel -- the inputs of find_uv_sample for a UV-textured render. */
extern "C" int find_render_frags(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t n_verts, int64_t n_faces, const void* ws,
								 int32_t* face_local, float* bary, void* stream) {
	int rc = check_params(rp, n_meshes, n_views, n_verts, n_faces);
	if (rc != FIND_OK) return rc;
	FIND_REQUIRE(ws && face_local && bary, "find_render_frags: NULL argument");
	Ws w;
	carve(rp, n_meshes, n_views, n_verts, n_faces, const_cast<void*>(ws), &w);
	const int64_t px = n_meshes * n_views * rp->image_h * rp->image_w;
	hipStream_t s = reinterpret_cast<hipStream_t>(stream);
	if (hipMemcpyAsync(face_local, w.p2f, px * sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess ||
		hipMemcpyAsync(bary, w.bary, px * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
		set_error("find_render_frags: copy failed");
		return FIND_ELAUNCH;
	}
	return FIND_OK;
}
