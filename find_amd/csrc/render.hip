// Differentiable mesh render for FIND on gfx950: projection, tile rasteriser fused with the soft-silhouette and
// Phong/softmax-blend shaders, and the backward passes.
//
// Replaces FootRenderer.forward / rasterize (reference src/model/renderer.py:208-245, 247-383): PyTorch3D's
// FoVPerspectiveCameras transform, rasterize_meshes (K=100 soft-silhouette pass + K=1 RGB pass), SoftSilhouetteShader,
// SoftPhongShader + softmax_rgb_blend (math restated in-repo at renderer.py:23-72).
//
// Design (HBM/VALU-bound, no MFMA): the K=100 fragment buffers PyTorch3D materialises (2.8 kB/pixel) never exist.
//   1. project_kernel   : world -> (x_ndc, y_ndc, z_view) per (image, vertex)                       [coalesced stream]
//   2. face_setup_kernel: per (image, face) cull + 48-B face record + the 8x8-pixel tiles its blurred bbox touches, packed in one
//      uint32 + the nearest depth it can contribute; per image the depth range and the largest depth extent of a face
//   3. bin_kernel       : one wave per 8x8-pixel tile builds the tile's face list (two-level scan of the packed bboxes: runs of 64
//      consecutive faces first) and sorts it, stably, by DEPTH SLAB (255 slabs over the image's depth range, none thinner than the
//      deepest face); lists go to a pool, tiles to queues by list length (crowded first); empty tiles get their background here
//   4. raster_kernel    : persistent waves take tiles from the queues; a wave stages 64 face records at a time in LDS and every lane
//      evaluates them for its own pixel -- no workgroup barrier anywhere.  Faces arrive front to back, so a pixel that already holds
//      faces_per_pixel candidates in front of everything still to come is FINISHED, and the wave leaves the list when all its pixels
//      are (an interior pixel of a closed surface never looks at the far side).  Candidates (depth, 1 - p) go to the lane's own
//      list in the wave's scratch (32-byte pieces, the 64 lanes' pieces side by side) through an LDS ring; a pixel with more than K of them selects the K nearest by a radix search on
//      its own list (the K-th depth is saved for the backward).  The tail shades the nearest inside fragment, writes mask / image.
//   5. tie_fix_kernel   : PyTorch3D keeps, among candidates of EQUAL depth at the K-th place, those of lower face index (insertion
//      order).  Lists are in depth order, not face order, so pixels where candidates tied at the K-th depth were left out are
//      queued and re-evaluated here with face ids at hand (a wave per pixel): exact K-buffer semantics, and the id of the last
//      face kept goes to the backward.
//   6. backward: silhouette gradient is FACE-centric (lanes walk the blurred bbox of their face and accumulate the six NDC gradients
//      in registers: no per-fragment atomics); projection backward is a deterministic sum over views.
// Every kernel evaluates a (pixel, face) pair with the SAME rounding (eval_frag is compiled without fp contraction, its fused
// multiply-adds are written out): the depth a candidate had in the forward is the depth the fix-up and the backward recompute.
// Conventions: SURVEY.md Appendix A.2-A.4 (row-vector transforms, NDC +x left / +y up, image = mesh*n_views + view).
#include <type_traits>

#include "common.h"

namespace find {
int g_raster_ablate = 0;
namespace render {

constexpr int T8 = 8;             // tile edge in pixels: one wave per tile
constexpr float KEPS = 1e-8f;
constexpr uint32_t TB_EMPTY = 0x000000FFu;  // tx0 = 255 > tx1 = 0
constexpr int KU = 16;            // list entries in flight per lane in the K-nearest passes (they are L2-latency-bound)
constexpr int KN_CAP = 4096;      // silhouette candidates kept per pixel for the K-nearest rule (more: unresolved, flags[1]); the pole of a 10 002-vertex lat-long scan @256^2 collects ~2000
constexpr int RING = 8;           // candidates a lane collects in LDS before it writes them out: 2 x 32 contiguous bytes per flush
constexpr int RASTER_WGS = 1024;  // persistent rasteriser workgroups (4 independent waves each); each owns 256 x KN_CAP x 8 B of scratch
constexpr int64_t BAND_MIN_PIXELS = 384 * 384;   // images of at least this many pixels go through raster_band_kernel (find_render_fwd)
constexpr int BIN_CAP = 2048;     // list entries a binning wave keeps (and sorts) in LDS; a longer list goes out in face order, without the early exit
constexpr int N_SLABS = 255;      // depth slabs per image (8 bits of a list entry; the other 24 are the face index)
constexpr int N_QUEUES = 24;      // tile queues by log2(list length), the longest lists first
constexpr int CURSOR_STRIDE = 32;  // ints between two counters of the cursor array
constexpr uint32_t LIST_UNSORTED = 1u << 30;
constexpr uint32_t FACE_MASK = 0x00FFFFFFu;
constexpr int REC_F4 = 8;         // 16-byte pieces of a face record (FaceRec)
constexpr int REC_DW = 32;        // its dwords
constexpr int PAIR_STRIDE = 68;   // floats between two pair blocks of the rasteriser's LDS staging: 2 x REC_DW + 4 (a stride of 64 dwords put every block on the same two banks: 32-way conflicts on the staging writes)

struct Fix {      // a pixel whose K-th depth is shared by more candidates than fit: resolved by tie_fix_kernel
	int32_t pix;      // (img * H + y) * W + x
	uint32_t zk;      // bits of the K-th depth
	int32_t keep;     // how many of the candidates at that depth belong to the K nearest
	float a_lt;       // product of (1 - p) over the candidates in front of it
};

struct FaceRec;
struct Ws {
	float* vproj;     // (n_img, V, 3)
	float4* frec;     // (n_img, F, 3) float4: [x0 y0 x1 y1][x2 y2 z0 z1][z2 - - -]   (backward kernels)
	struct FaceRec* recs;  // (n_img, F) the same face as the affine forms of the fragment math (128 B), staged through LDS by the rasteriser
	uint32_t* rz;     // (n_img, ceil(F / 64), 4) per run of 64 faces: bits of the smallest / largest fzmin, of the largest depth extent
	uint32_t* tb;     // (n_img, F) packed bbox in 8-pixel tiles: x0 | x1 << 8 | y0 << 16 | y1 << 24 (TB_EMPTY: culled)
	uint32_t* tbb;    // (n_img, ceil(F / 64)) the same for every run of 64 consecutive faces
	float* fzmin;     // (n_img, F) nearest depth a face can contribute (min vertex depth, not below 0)
	int32_t* zinfo;   // (n_img, 8): [0] bits of the smallest fzmin, [1] of the largest, [2] of the largest depth extent of a face, [4..7] tile bbox of all faces
	float* normals;   // (n_meshes, V, 3) area-weighted vertex normals (world)
	int32_t* p2f;     // (n_img, H, W) nearest inside face (local id) or -1   [saved for backward]
	float* bary;      // (n_img, H, W, 3) its perspective-correct barycentrics
	float* frag;      // (n_img, H, W, 8) [w0 w1 w2 z | d - - -] of that fragment, between the rasteriser and shade_kernel
	float* d_vproj;   // (n_img, V, 3) backward accumulator
	float* d_normals; // (n_meshes, V, 3) backward accumulator
	float* raw_normals; // (n_meshes, V, 3) un-normalised vertex-normal sums (backward)
	int32_t* flags;   // [0] straddling faces seen, [1] overflow pixels left unresolved (> KN_CAP candidates), [3] tiles taken from the queues, [6] pool cursor, [7] fix-up entries, [8..] diagnostics
	int32_t* qn;      // [0..N_QUEUES) tiles per list-length class, [32..32+N_QUEUES) fill cursors of the classes, [62] tiles with a list
	int32_t* cursor;  // (n_img) entries handed out of each image's part of the pool
	float* zthr;      // (n_img, H, W) depth bound of the K nearest silhouette candidates (+inf: every candidate counts; negative: -depth, ties at it resolved by tie_face)  [backward]
	float* alpha;     // (n_img, H, W) prod (1 - p_k) over the blended candidates  [backward: 1 - mask has lost it wherever the mask rounds to 1]
	int32_t* tie_face; // (n_img, H, W) last face kept among those tied at the K-th depth (valid where zthr < 0)  [backward]
	float2* scratch;  // (raster workgroups, 2, 256, KN_CAP) per-pixel candidate lists of the tile in flight: depths, then 1 - p; per wave, 32-byte piece k of lane l at piece index 64 k + l
	int2* tinfo;      // (n_img, tiles) list of a tile: x = offset into pool, y = length | LIST_UNSORTED, or -1: no room in the pool (the rasteriser scans the faces itself)
	uint32_t* pool;   // list entries: slab << 24 | face
	int32_t* order;   // (n_img * tiles) ids of the tiles that have a list, the longest lists first
	Fix* fix;         // (n_img * H * W) fix-up queue
	int64_t pool_cap, n_tiles, raster_wgs;
	int64_t bytes;
};

static void carve(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t V, int64_t F, void* ws, Ws* o) {
	Carver c(ws);
	const int64_t n_img = n_meshes * n_views;
	const int64_t px = n_img * rp->image_h * rp->image_w;
	o->flags = c.take<int32_t>(64);
	o->qn = c.take<int32_t>(64);
	// per image: entries handed out of its part of the pool -- one counter per 128-byte line (atomics on one line are served one after
	// the other, ~50 ns each: 64 counters packed into two lines were the slowest thing in the binning kernel, 260 us)
	o->cursor = c.take<int32_t>(n_img * CURSOR_STRIDE);
	o->vproj = c.take<float>(n_img * V * 3);
	o->frec = c.take<float4>(n_img * F * 3);
	o->recs = reinterpret_cast<FaceRec*>(c.take<float4>(n_img * F * REC_F4));
	o->rz = c.take<uint32_t>(n_img * cdiv(F, 64) * 4);
	o->tb = c.take<uint32_t>(n_img * F);
	o->tbb = c.take<uint32_t>(n_img * cdiv(F, 64));
	o->fzmin = c.take<float>(n_img * F);
	o->zinfo = c.take<int32_t>(n_img * 8);
	o->normals = c.take<float>(n_meshes * V * 3);
	o->p2f = c.take<int32_t>(px);
	o->bary = c.take<float>(px * 3);
	o->frag = c.take<float>(px * 8);
	o->d_vproj = c.take<float>(n_img * V * 3);
	o->d_normals = c.take<float>(n_meshes * V * 3);
	o->raw_normals = c.take<float>(n_meshes * V * 3);
	o->zthr = c.take<float>(px);
	o->alpha = c.take<float>(px);
	o->tie_face = c.take<int32_t>(px);
	o->n_tiles = n_img * cdiv(rp->image_w, T8) * cdiv(rp->image_h, T8);
	o->raster_wgs = std::min<int64_t>(cdiv(o->n_tiles, 4), RASTER_WGS);
	// (raster_kernel: 256 x KN_CAP x 8 B per workgroup; raster_band_kernel, render_band.h: 4 waves x 3 x KN_CAP x 64 x 4 B -- the larger)
	o->scratch = c.take<float2>(o->raster_wgs * KN_CAP * 256 * 3 / 2);
	o->tinfo = c.take<int2>(o->n_tiles);
	// a face of ~1 pixel with the silhouette's blur margin touches ~4.5 tiles at 256^2 and ~10 at 512^2; a tile that finds the pool
	// full is rasterised from the face arrays directly (slower, never wrong)
	// (every image has its own part of the pool and its own cursor: one cursor for all was the hottest address of the launch)
	o->pool_cap = std::min<int64_t>(F * 16 + (1 << 14), ((int64_t)1 << 30) / n_img);
	o->pool = c.take<uint32_t>(o->pool_cap * n_img);
	o->order = c.take<int32_t>(o->n_tiles);
	o->fix = c.take<Fix>(px);
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
// (pixel, face) arithmetic below is compiled WITHOUT floating-point contraction and its fused multiply-adds are explicit: the
// compiler may otherwise fuse the same expression differently in the rasteriser, the tie fix-up and the backward, and a candidate's
// depth is compared bit for bit between them.
__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
#pragma clang fp contract(off)
	return __builtin_fmaf(px - ax, by - ay, -((py - ay) * (bx - ax)));
}

// pixel index range [lo, hi] whose centres 1-(2i+1)/S may fall in NDC [cmin, cmax] (one pixel of slack; exact test per pixel)
__device__ __forceinline__ void pix_range(float cmin, float cmax, int S, int* lo, int* hi) {
	const float a = ((1.0f - cmax) * S - 1.0f) * 0.5f;
	const float b = ((1.0f - cmin) * S - 1.0f) * 0.5f;
	*lo = max((int)floorf(a) - 1, 0);
	*hi = min((int)ceilf(b) + 1, S - 1);
}

// the same range without the pixel of slack (a hundredth of a pixel instead: far above the rounding of the two expressions): which
// tiles a face is listed in
__device__ __forceinline__ void pix_range_tight(float cmin, float cmax, int S, int* lo, int* hi) {
	const float a = ((1.0f - cmax) * S - 1.0f) * 0.5f;
	const float b = ((1.0f - cmin) * S - 1.0f) * 0.5f;
	*lo = max((int)ceilf(a - 0.01f), 0);
	*hi = min((int)floorf(b + 0.01f), S - 1);
}


// ------------------------------------------------------------------------------------------------ shared fragment math
// Everything of the (pixel, face) arithmetic that does not depend on the pixel is computed ONCE per (image, face), in double, by
// face_setup_kernel, as AFFINE FORMS in d = p - v0 (round 5; rounds 1-4 kept vertices, 1 / area and 1 / |edge|^2 and rebuilt the edge
// functions per pixel: ~100 multiply-add-class operations per pixel and face, ~60 now):
//   perspective-weighted barycentric numerators   t_i(p) = e_i(p) / area * z_j z_k = Tix dx + Tiy dy (+ T0c: e_1 and e_2 vanish at v0)
//   projection parameters on the three edges       u_k(p) = (p - a_k) . (b_k - a_k) / |b_k - a_k|^2 = Ukx dx + Uky dy + Ukc
//                                                  (a degenerate edge: U = 0, Uc = 1 -- the parameter is then 1, the far end, as PyTorch3D)
// with the origin at the face's own first vertex the forms lose nothing against the difference form (|d| is the blurred bbox's size, not
// the NDC coordinate: tools-free check in DESIGN 4.2).  PyTorch3D's order of operations -- w_i = e_i / area, then t_i = w_i z_j z_k,
// w'_i = t_i / max(sum t, eps), clipped c_i = max(w'_i, 0) / max(sum, 1e-5), depth = sum c_i z_i, squared distance to the nearest of the
// three segments with the parameter clamped to [0, 1] -- is unchanged from t_i on.
struct FaceRec {  // 128 bytes = eight 16-byte pieces; written once per (image, face) by face_setup_kernel
	float ox, oy, x1, y1;     // v0 (NDC); v1 - v0
	float x2, y2, ex, ey;     // v2 - v0; v2 - v1
	float z0, z1, z2, T0c;
	float T0x, T0y, T1x, T1y;
	float T2x, T2y, U0x, U0y; // edge 0: v0 v1
	float U0c, U1x, U1y, U1c; // edge 1: v0 v2
	float U2x, U2y, U2c; int f;   // edge 2: v1 v2;  f: the face index (the rasteriser's staging puts the list entry here: slab << 24 | face)
	float xmin, xmax, ymin, ymax;     // blurred NDC bbox
};
static_assert(sizeof(FaceRec) == 128, "FaceRec is read as 32 dwords");

// (pixel, face) arithmetic below is compiled WITHOUT floating-point contraction and its fused multiply-adds are explicit: the
// compiler may otherwise fuse the same expression differently in the rasteriser's two sweeps and the backward, and a candidate's
// depth is compared bit for bit between them.
__device__ __forceinline__ float sqlen_f32(float dx, float dy) {
#pragma clang fp contract(off)
	const float a = dx * dx, b = dy * dy;
	return a + b;
}
__device__ __forceinline__ void make_rec(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2, int f, float br, FaceRec* r) {
	// double: differences of fp32 coordinates are exact, products and quotients correctly rounded once
	const double X1 = (double)x1 - x0, Y1 = (double)y1 - y0, X2 = (double)x2 - x0, Y2 = (double)y2 - y0, EX = (double)x2 - x1, EY = (double)y2 - y1;
	const double area = X2 * Y1 - Y2 * X1;                 // edge(v2; v0, v1)
	const double ia = 1.0 / (area + (double)KEPS);
	const double k0 = ia * (double)z1 * (double)z2, k1 = ia * (double)z0 * (double)z2, k2 = ia * (double)z0 * (double)z1;
	r->ox = x0; r->oy = y0; r->x1 = (float)X1; r->y1 = (float)Y1; r->x2 = (float)X2; r->y2 = (float)Y2; r->ex = (float)EX; r->ey = (float)EY;
	r->z0 = z0; r->z1 = z1; r->z2 = z2;
	// e_0(p) = edge(p; v1, v2) = EY dx - EX dy + area;  e_1(p) = edge(p; v2, v0) = -Y2 dx + X2 dy;  e_2(p) = edge(p; v0, v1) = Y1 dx - X1 dy
	r->T0x = (float)(EY * k0); r->T0y = (float)(-EX * k0); r->T0c = (float)(area * k0);
	r->T1x = (float)(-Y2 * k1); r->T1y = (float)(X2 * k1);
	r->T2x = (float)(Y1 * k2); r->T2y = (float)(-X1 * k2);
	const double l01 = X1 * X1 + Y1 * Y1, l02 = X2 * X2 + Y2 * Y2, l12 = EX * EX + EY * EY;
	// "degenerate" is a threshold decision (PyTorch3D: squared length <= 1e-8 in fp32) and not a continuous quantity: the ring edges at the
	// poles of a 50 002-vertex lat-long template @64^2 are 1e-4 long, right AT it.  Decided in the reference's own arithmetic -- fp32
	// differences, fp32 products, fp32 sum, no contraction -- so that the same edges take the far-end branch on both sides.
	const float f01 = sqlen_f32(x1 - x0, y1 - y0), f02 = sqlen_f32(x2 - x0, y2 - y0), f12 = sqlen_f32(x2 - x1, y2 - y1);
	const bool d01 = f01 <= KEPS, d02 = f02 <= KEPS, d12 = f12 <= KEPS;
	r->U0x = d01 ? 0.f : (float)(X1 / l01); r->U0y = d01 ? 0.f : (float)(Y1 / l01); r->U0c = d01 ? 1.f : 0.f;
	r->U1x = d02 ? 0.f : (float)(X2 / l02); r->U1y = d02 ? 0.f : (float)(Y2 / l02); r->U1c = d02 ? 1.f : 0.f;
	r->U2x = d12 ? 0.f : (float)(EX / l12); r->U2y = d12 ? 0.f : (float)(EY / l12); r->U2c = d12 ? 1.f : (float)(-(X1 * EX + Y1 * EY) / l12);
	r->f = f;
	r->xmin = fminf(x0, fminf(x1, x2)) - br; r->xmax = fmaxf(x0, fmaxf(x1, x2)) + br;
	r->ymin = fminf(y0, fminf(y1, y2)) - br; r->ymax = fmaxf(y0, fmaxf(y1, y2)) + br;
}
__device__ __forceinline__ void store_rec(FaceRec* dst, const FaceRec& r) {
	float4* d = reinterpret_cast<float4*>(dst);
	d[0] = make_float4(r.ox, r.oy, r.x1, r.y1); d[1] = make_float4(r.x2, r.y2, r.ex, r.ey); d[2] = make_float4(r.z0, r.z1, r.z2, r.T0c);
	d[3] = make_float4(r.T0x, r.T0y, r.T1x, r.T1y); d[4] = make_float4(r.T2x, r.T2y, r.U0x, r.U0y); d[5] = make_float4(r.U0c, r.U1x, r.U1y, r.U1c);
	d[6] = make_float4(r.U2x, r.U2y, r.U2c, __int_as_float(r.f)); d[7] = make_float4(r.xmin, r.xmax, r.ymin, r.ymax);
}
__device__ __forceinline__ FaceRec load_rec(const float4* q, int swz = 0) {   // swz: piece k sits at k ^ swz (the second sweep's LDS layout)
	const float4 q0 = q[0 ^ swz], q1 = q[1 ^ swz], q2 = q[2 ^ swz], q3 = q[3 ^ swz], q4 = q[4 ^ swz], q5 = q[5 ^ swz], q6 = q[6 ^ swz], q7 = q[7 ^ swz];
	FaceRec r;
	r.ox = q0.x; r.oy = q0.y; r.x1 = q0.z; r.y1 = q0.w; r.x2 = q1.x; r.y2 = q1.y; r.ex = q1.z; r.ey = q1.w; r.z0 = q2.x; r.z1 = q2.y; r.z2 = q2.z; r.T0c = q2.w;
	r.T0x = q3.x; r.T0y = q3.y; r.T1x = q3.z; r.T1y = q3.w; r.T2x = q4.x; r.T2y = q4.y; r.U0x = q4.z; r.U0y = q4.w; r.U0c = q5.x; r.U1x = q5.y; r.U1y = q5.z; r.U1c = q5.w;
	r.U2x = q6.x; r.U2y = q6.y; r.U2c = q6.z; r.f = __float_as_int(q6.w); r.xmin = q7.x; r.xmax = q7.y; r.ymin = q7.z; r.ymax = q7.w;
	return r;
}

struct Frag {
	float w0, w1, w2;     // perspective-correct barycentrics, unclipped
	float pz_clip;        // depth from clipped barycentrics (silhouette pass)
	float pz;             // depth from unclipped barycentrics (RGB pass)
	float dist;           // unsigned squared distance to the triangle outline
	bool inside;
	int edge;             // nearest edge: 0 = v0v1, 1 = v0v2, 2 = v1v2
	float t;              // its clamped projection parameter
	float qx, qy;         // nearest point of that edge minus the pixel (dist = qx^2 + qy^2)
};

// geometry_utils: barycentric, perspective correction, clip, depth, point-triangle distance.  Returns false when the
// pixel is outside the blurred bbox.
__device__ __forceinline__ void eval_core(const FaceRec& r, float px, float py, Frag* o);
__device__ __forceinline__ bool eval_frag(const FaceRec& r, float px, float py, Frag* o) {
	if (px > r.xmax || px < r.xmin || py > r.ymax || py < r.ymin) return false;
	eval_core(r, px, py, o);
	return true;
}
// the same for a pixel already known to lie inside the blurred bbox (the rasteriser tests 64 pixels against it first)
__device__ __forceinline__ void eval_core(const FaceRec& r, float px, float py, Frag* o) {
#pragma clang fp contract(off)
	const float dx = px - r.ox, dy = py - r.oy;
	const float t0 = __builtin_fmaf(r.T0x, dx, __builtin_fmaf(r.T0y, dy, r.T0c));
	const float t1 = __builtin_fmaf(r.T1x, dx, r.T1y * dy);
	const float t2 = __builtin_fmaf(r.T2x, dx, r.T2y * dy);
	const float iden = __builtin_amdgcn_rcpf(fmaxf(t0 + t1 + t2, KEPS));
	const float w0 = t0 * iden, w1 = t1 * iden, w2 = t2 * iden;
	o->w0 = w0; o->w1 = w1; o->w2 = w2;
	o->inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
	float c0 = fmaxf(w0, 0.f), c1 = fmaxf(w1, 0.f), c2 = fmaxf(w2, 0.f);
	const float isum = __builtin_amdgcn_rcpf(fmaxf(c0 + c1 + c2, 1e-5f));
	c0 *= isum; c1 *= isum; c2 *= isum;
	o->pz_clip = __builtin_fmaf(c2, r.z2, __builtin_fmaf(c1, r.z1, c0 * r.z0));
	o->pz = __builtin_fmaf(w2, r.z2, __builtin_fmaf(w1, r.z1, w0 * r.z0));
	const float ta = __builtin_amdgcn_fmed3f(__builtin_fmaf(r.U0x, dx, __builtin_fmaf(r.U0y, dy, r.U0c)), 0.f, 1.f);
	const float tb = __builtin_amdgcn_fmed3f(__builtin_fmaf(r.U1x, dx, __builtin_fmaf(r.U1y, dy, r.U1c)), 0.f, 1.f);
	const float tc = __builtin_amdgcn_fmed3f(__builtin_fmaf(r.U2x, dx, __builtin_fmaf(r.U2y, dy, r.U2c)), 0.f, 1.f);
	const float ax = __builtin_fmaf(ta, r.x1, -dx), ay = __builtin_fmaf(ta, r.y1, -dy);
	const float bx = __builtin_fmaf(tb, r.x2, -dx), by = __builtin_fmaf(tb, r.y2, -dy);
	const float cx = __builtin_fmaf(tc, r.ex, r.x1 - dx), cy = __builtin_fmaf(tc, r.ey, r.y1 - dy);
	const float e01 = __builtin_fmaf(ax, ax, ay * ay), e02 = __builtin_fmaf(bx, bx, by * by), e12 = __builtin_fmaf(cx, cx, cy * cy);
	if (e01 <= e02 && e01 <= e12) { o->dist = e01; o->edge = 0; o->t = ta; o->qx = ax; o->qy = ay; }
	else if (e02 <= e01 && e02 <= e12) { o->dist = e02; o->edge = 1; o->t = tb; o->qx = bx; o->qy = by; }
	else { o->dist = e12; o->edge = 2; o->t = tc; o->qx = cx; o->qy = cy; }
}

// ---- the same fragment math for TWO faces at once (the rasteriser's inner loop): every quantity is a pair (face a, face b) in the two
// halves of a 64-bit register pair, so the multiplies, adds and fused multiply-adds are packed instructions (v_pk_*_f32: one issue slot for
// both faces).  Operation for operation the expressions of eval_core above, in the same order, without contraction: each half is
// bit-identical to eval_core on that face (the second sweep and the backward recompute depths with eval_core and compare them to the bit;
// tests/test_gpu_render.py).  Maxima, medians, compares, selects and reciprocals have no packed form.
typedef float v2f __attribute__((ext_vector_type(2)));
struct Frag2 {
	v2f w0, w1, w2, pz_clip, pz, dist;
	bool inside_a, inside_b;
};
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f max2(v2f a, float b) { return v2f{fmaxf(a.x, b), fmaxf(a.y, b)}; }
__device__ __forceinline__ v2f rcp2(v2f a) { return v2f{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ v2f med2(v2f a) { return v2f{__builtin_amdgcn_fmed3f(a.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(a.y, 0.f, 1.f)}; }
// blk: the pair's 64 floats in LDS, field j of face a at [2 j], of face b at [2 j + 1] (FaceRec's dword order)
__device__ __forceinline__ void eval_pair(const float* blk, float pxs, float pys, Frag2* o) {
#pragma clang fp contract(off)
	const float4* b4 = reinterpret_cast<const float4*>(blk);
	const float4 r0 = b4[0], r1 = b4[1], r2 = b4[2], r3 = b4[3], r4 = b4[4], r5 = b4[5], r6 = b4[6], r7 = b4[7], r8 = b4[8], r9 = b4[9], r10 = b4[10], r11 = b4[11];
	const float4 r12 = b4[12], r13 = b4[13];
	const v2f ox = {r0.x, r0.y}, oy = {r0.z, r0.w}, x1 = {r1.x, r1.y}, y1 = {r1.z, r1.w}, x2 = {r2.x, r2.y}, y2 = {r2.z, r2.w}, ex = {r3.x, r3.y}, ey = {r3.z, r3.w};
	const v2f z0 = {r4.x, r4.y}, z1 = {r4.z, r4.w}, z2 = {r5.x, r5.y}, T0c = {r5.z, r5.w}, T0x = {r6.x, r6.y}, T0y = {r6.z, r6.w}, T1x = {r7.x, r7.y}, T1y = {r7.z, r7.w};
	const v2f T2x = {r8.x, r8.y}, T2y = {r8.z, r8.w}, U0x = {r9.x, r9.y}, U0y = {r9.z, r9.w}, U0c = {r10.x, r10.y}, U1x = {r10.z, r10.w}, U1y = {r11.x, r11.y}, U1c = {r11.z, r11.w};
	const v2f U2x = {r12.x, r12.y}, U2y = {r12.z, r12.w}, U2c = {r13.x, r13.y};
	const v2f dx = v2f{pxs, pxs} - ox, dy = v2f{pys, pys} - oy;
	const v2f t0 = fma2(T0x, dx, fma2(T0y, dy, T0c));
	const v2f t1 = fma2(T1x, dx, T1y * dy);
	const v2f t2 = fma2(T2x, dx, T2y * dy);
	const v2f iden = rcp2(max2(t0 + t1 + t2, KEPS));
	const v2f w0 = t0 * iden, w1 = t1 * iden, w2 = t2 * iden;
	o->w0 = w0; o->w1 = w1; o->w2 = w2;
	o->inside_a = w0.x > 0.f && w1.x > 0.f && w2.x > 0.f;
	o->inside_b = w0.y > 0.f && w1.y > 0.f && w2.y > 0.f;
	v2f c0 = max2(w0, 0.f), c1 = max2(w1, 0.f), c2 = max2(w2, 0.f);
	const v2f isum = rcp2(max2(c0 + c1 + c2, 1e-5f));
	c0 *= isum; c1 *= isum; c2 *= isum;
	o->pz_clip = fma2(c2, z2, fma2(c1, z1, c0 * z0));
	o->pz = fma2(w2, z2, fma2(w1, z1, w0 * z0));
	const v2f ta = med2(fma2(U0x, dx, fma2(U0y, dy, U0c)));
	const v2f tb = med2(fma2(U1x, dx, fma2(U1y, dy, U1c)));
	const v2f tc = med2(fma2(U2x, dx, fma2(U2y, dy, U2c)));
	const v2f ax = fma2(ta, x1, -dx), ay = fma2(ta, y1, -dy);
	const v2f bx = fma2(tb, x2, -dx), by = fma2(tb, y2, -dy);
	const v2f cx = fma2(tc, ex, x1 - dx), cy = fma2(tc, ey, y1 - dy);
	const v2f e01 = fma2(ax, ax, ay * ay), e02 = fma2(bx, bx, by * by), e12 = fma2(cx, cx, cy * cy);
	// (the value eval_core's three-way choice ends up with is the smallest of the three)
	o->dist = v2f{fminf(fminf(e01.x, e02.x), e12.x), fminf(fminf(e01.y, e02.y), e12.y)};
}

// sigmoid(-d / sigma).  The reciprocal is the hardware's (1 ulp) rather than an IEEE division (ten instructions in the innermost loop of the
// rasteriser and of its backward); forward and backward call this one function.
__device__ __forceinline__ float silhouette_prob(float signed_dist, float inv_sigma) { return __builtin_amdgcn_rcpf(1.0f + __expf(signed_dist * inv_sigma)); }

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
	const float l = fmaxf(sqrtf(x * x + y * y + z * z), 1e-6f);
	x /= l; y /= l; z /= l;
}

// per (image, face): cull, the records (backward: 48 B; forward: the full 80 B), the tiles its blurred bbox touches, its nearest depth;
// per run of 64 faces: their common tile bbox and depth statistics (zinfo_kernel folds those into the image's)
__global__ void face_setup_kernel(const float* __restrict__ vproj, const int32_t* __restrict__ faces, int64_t faces_mesh_stride,
								  int n_views, int V, int F, int H, int W, float blur_radius, float z_clip,
								  float4* __restrict__ frec, FaceRec* __restrict__ recs, uint32_t* __restrict__ tb, uint32_t* __restrict__ tbb,
								  float* __restrict__ fzmin, uint32_t* __restrict__ rz, int32_t* __restrict__ flags) {
	const int img = blockIdx.y;
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	const int mesh = img / n_views;
	const int32_t* fp = faces + (int64_t)mesh * faces_mesh_stride + (int64_t)min(f, F - 1) * 3;
	uint32_t packed = TB_EMPTY;
	const int64_t o = (int64_t)img * F + f;
	float zmn = 0.f, zext = 0.f;
	if (f < F && fp[0] >= 0) {
		const float* vp = vproj + (int64_t)img * V * 3;
		const float x0 = vp[3 * fp[0]], y0 = vp[3 * fp[0] + 1], z0 = vp[3 * fp[0] + 2];
		const float x1 = vp[3 * fp[1]], y1 = vp[3 * fp[1] + 1], z1 = vp[3 * fp[1] + 2];
		const float x2 = vp[3 * fp[2]], y2 = vp[3 * fp[2] + 1], z2 = vp[3 * fp[2] + 2];
		const float4 ra = make_float4(x0, y0, x1, y1), rb = make_float4(x2, y2, z0, z1), rc = make_float4(z2, 0.f, 0.f, 0.f);
		frec[o * 3 + 0] = ra; frec[o * 3 + 1] = rb; frec[o * 3 + 2] = rc;
		const float br = sqrtf(blur_radius);
		FaceRec r;
		make_rec(x0, y0, z0, x1, y1, z1, x2, y2, z2, f, br, &r);
		store_rec(recs + o, r);
		const bool all_behind = z0 < z_clip && z1 < z_clip && z2 < z_clip;
		const bool any_behind = z0 < z_clip || z1 < z_clip || z2 < z_clip;
		if (any_behind && !all_behind) atomicAdd(&flags[0], 1);  // straddling the clip plane: PyTorch3D would clip it (clip.py)
		const float zmax = fmaxf(z0, fmaxf(z1, z2));
		const float area = edge_fn(x0, y0, x1, y1, x2, y2);
		const bool degenerate = area <= KEPS && area >= -KEPS;
		if (!all_behind && !(zmax < 0.f) && !degenerate) {
			int xlo, xhi, ylo, yhi;
			pix_range_tight(r.xmin, r.xmax, W, &xlo, &xhi);
			pix_range_tight(r.ymin, r.ymax, H, &ylo, &yhi);
			if (xlo <= xhi && ylo <= yhi) {
				packed = (uint32_t)(xlo / T8) | ((uint32_t)(xhi / T8) << 8) | ((uint32_t)(ylo / T8) << 16) | ((uint32_t)(yhi / T8) << 24);
				// a fragment's depth is a convex mix of the three vertex depths (clipped or inside barycentrics) and is dropped below 0
				zmn = fmaxf(fminf(z0, fminf(z1, z2)), 0.f);
				zext = zmax - zmn;
			}
		}
	}
	if (f < F) { tb[o] = packed; fzmin[o] = zmn; }
	// per wave: the tile bbox of its 64 consecutive faces (the binning waves test these first and only open the runs that reach their
	// tile: faces that are neighbours in the index are neighbours on the surface in any mesh that was not shuffled on purpose), and
	// their depth statistics
	const bool live = packed != TB_EMPTY;
	int bx0 = packed & 255, bx1 = (packed >> 8) & 255, by0 = (packed >> 16) & 255, by1 = packed >> 24;
	if (!live) { bx0 = 255; bx1 = 0; by0 = 255; by1 = 0; }
	uint32_t zlo = live ? __float_as_uint(zmn) : 0xFFFFFFFFu, zhi = live ? __float_as_uint(zmn) : 0u, zex = live ? __float_as_uint(zext) : 0u;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		bx0 = min(bx0, __shfl_xor(bx0, d, 64)); bx1 = max(bx1, __shfl_xor(bx1, d, 64));
		by0 = min(by0, __shfl_xor(by0, d, 64)); by1 = max(by1, __shfl_xor(by1, d, 64));
		zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, d, 64)); zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, d, 64));
		zex = max(zex, (uint32_t)__shfl_xor((int)zex, d, 64));
	}
	if ((threadIdx.x & 63) == 0 && f < F) {
		const int64_t ro = (int64_t)img * ((F + 63) / 64) + f / 64;
		tbb[ro] = bx0 > bx1 ? TB_EMPTY : ((uint32_t)bx0 | ((uint32_t)bx1 << 8) | ((uint32_t)by0 << 16) | ((uint32_t)by1 << 24));
		rz[ro * 4] = zlo; rz[ro * 4 + 1] = zhi; rz[ro * 4 + 2] = zex; rz[ro * 4 + 3] = 0u;
	}
}

// one workgroup per image: depth range of its faces, their largest depth extent, the tile bbox of all of them
__global__ __launch_bounds__(256) void zinfo_kernel(const uint32_t* __restrict__ tbb, const uint32_t* __restrict__ rz, int n_runs, int32_t* __restrict__ zinfo) {
	__shared__ uint32_t red[4][8];
	const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t zlo = 0xFFFFFFFFu, zhi = 0u, zex = 0u;
	int bx0 = 0x7FFFFFFF, bx1 = -1, by0 = 0x7FFFFFFF, by1 = -1;
	for (int r = threadIdx.x; r < n_runs; r += 256) {
		const int64_t ro = (int64_t)img * n_runs + r;
		const uint32_t t = tbb[ro];
		if (t == TB_EMPTY) continue;
		zlo = min(zlo, rz[ro * 4]); zhi = max(zhi, rz[ro * 4 + 1]); zex = max(zex, rz[ro * 4 + 2]);
		bx0 = min(bx0, (int)(t & 255)); bx1 = max(bx1, (int)((t >> 8) & 255)); by0 = min(by0, (int)((t >> 16) & 255)); by1 = max(by1, (int)(t >> 24));
	}
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, d, 64)); zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, d, 64)); zex = max(zex, (uint32_t)__shfl_xor((int)zex, d, 64));
		bx0 = min(bx0, __shfl_xor(bx0, d, 64)); bx1 = max(bx1, __shfl_xor(bx1, d, 64)); by0 = min(by0, __shfl_xor(by0, d, 64)); by1 = max(by1, __shfl_xor(by1, d, 64));
	}
	if (lane == 0) { red[wave][0] = zlo; red[wave][1] = zhi; red[wave][2] = zex; red[wave][4] = (uint32_t)bx0; red[wave][5] = (uint32_t)bx1; red[wave][6] = (uint32_t)by0; red[wave][7] = (uint32_t)by1; }
	__syncthreads();
	if (threadIdx.x == 0) {
		int32_t* z = zinfo + img * 8;
		z[0] = (int32_t)min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
		z[1] = (int32_t)max(max(red[0][1], red[1][1]), max(red[2][1], red[3][1]));
		z[2] = (int32_t)max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
		z[3] = 0;
		z[4] = min(min((int)red[0][4], (int)red[1][4]), min((int)red[2][4], (int)red[3][4]));
		z[5] = max(max((int)red[0][5], (int)red[1][5]), max((int)red[2][5], (int)red[3][5]));
		z[6] = min(min((int)red[0][6], (int)red[1][6]), min((int)red[2][6], (int)red[3][6]));
		z[7] = max(max((int)red[0][7], (int)red[1][7]), max((int)red[2][7], (int)red[3][7]));
	}
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

// ------------------------------------------------------------------------------------------------ 3. binning
struct TileArgs {
	find_render_params rp;
	const uint32_t* tb;
	const uint32_t* tbb;     // per run of 64 faces
	const float* fzmin;
	const int32_t* zinfo;
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
	float* frag_ws;          // (n_img,H,W,8) nearest inside fragment: barycentrics, depth, signed distance (rasteriser -> shade_kernel)
	int32_t* flags;
	int32_t* qn;
	int32_t* cursor;
	float* zthr;
	float* alpha_ws;
	int32_t* tie_face;
	float2* scratch;
	int2* tinfo;
	uint32_t* pool;
	int32_t* order;
	Fix* fix;
	int64_t pool_cap;
	int tiles_per_img, total_tiles;
	int ablate;              // profiling only (find_debug_raster_ablate): 1 no candidate lists, 2 no K-nearest pass, 4 no fragment math, 8 no early exit, 16 lists unsorted, 32 no scan, 64 counters, 256 tiny list pool
};

__device__ __forceinline__ bool tile_hit(uint32_t t, int tile_x, int tile_y) {
	const int tx0 = t & 255, tx1 = (t >> 8) & 255, ty0 = (t >> 16) & 255, ty1 = t >> 24;
	return tile_x >= tx0 && tile_x <= tx1 && tile_y >= ty0 && tile_y <= ty1;  // (TB_EMPTY: tx0 = 255 > tx1 = 0)
}

// depth slabs of an image: lower edge of the first slab and the slab width.  No slab is thinner than the deepest face plus a margin far
// above the rounding of a fragment's depth: a fragment of a face whose nearest depth lies in slab s lies in front of slab s + 2.
__device__ __forceinline__ void slab_layout(const int32_t* zi, float* zlo, float* w) {
	const float lo = __uint_as_float((uint32_t)zi[0]), hi = __uint_as_float((uint32_t)zi[1]), ext = __uint_as_float((uint32_t)zi[2]);
	*zlo = lo;
	*w = fmaxf((hi - lo) * (1.0f / (N_SLABS - 1)), ext + 1e-4f * hi + 1e-12f);
}
__device__ __forceinline__ int slab_of(float zmin, float zlo, float inv_w) { return min(N_SLABS - 1, max(0, (int)((zmin - zlo) * inv_w))); }
// every face of slab s (and later) has its fragments behind this depth; a little short of the slab's lower edge, for the rounding of both sides
__device__ __forceinline__ float slab_front(int s, float zlo, float w) { return (zlo + (float)s * w) * (1.0f - 2e-6f); }

// One wave: its LDS operations execute in order, so all a "barrier" has to do is keep the compiler from moving LDS accesses across it and
// wait for the LDS counter.  (NOT an acquire / release fence: those also wait for the wave's outstanding GLOBAL stores -- the list
// entries on their way to the pool, the candidate pieces on their way to the scratch -- microseconds each, twice per 64 faces.)
__device__ __forceinline__ void wave_lds_sync() {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

// One wave per 8x8-pixel tile, four independent waves per workgroup (no barrier).  The wave scans the packed bboxes on two levels --
// the runs of 64 consecutive faces, then the faces of the runs that reach its tile, four runs' loads in flight together: the scan is
// bound by the latency of these loads, not by their 4 B per face -- and appends the hits, in face order, to its list in LDS; sorts the
// list by depth slab on its way to the pool (a stable counting sort: faces keep their index order inside a slab, so the list -- and with
// it every sum the rasteriser forms -- is the same in every run); and queues the tile by the length of its list.  A tile no face reaches gets its background right here.
__device__ __forceinline__ void write_background(const TileArgs& a, int img, int tile_x, int tile_y, int lane) {
	const int H = a.rp.image_h, W = a.rp.image_w;
	const int xi = tile_x * T8 + (lane & 7), yi = tile_y * T8 + (lane >> 3);
	if (xi < W && yi < H) {
		const int64_t pix = ((int64_t)img * H + yi) * W + xi;
		if (a.mask) { a.mask[pix] = 0.0f; a.zthr[pix] = INFINITY; a.alpha_ws[pix] = 1.0f; }
		if (a.p2f_ws) a.p2f_ws[pix] = -1;
		if (a.p2f_out) a.p2f_out[pix] = -1;
		if (a.zbuf_out) a.zbuf_out[pix] = -1.0f;
		if (a.image) { float* o = a.image + pix * 3; o[0] = a.rp.background[0]; o[1] = a.rp.background[1]; o[2] = a.rp.background[2]; }
	}
}

__device__ __forceinline__ void bin_tile(const TileArgs& a, uint32_t* src, int* hist, int* runs, int img, int t, int tile_x, int tile_y, int lane,
										  unsigned long long lt, int F, int n_runs) {
	const int32_t* zi = a.zinfo + img * 8;
	const int t_id = img * a.tiles_per_img + t;
	int n = 0;
	float zlo = 0.f, w = 1.f;
	slab_layout(zi, &zlo, &w);
	const float inv_w = 1.0f / w;
	const uint32_t* tbbp = a.tbb + (int64_t)img * n_runs;
	const uint32_t* tbp = a.tb + (int64_t)img * F;
	const float* fzp = a.fzmin + (int64_t)img * F;
	// scan: fn(entry, position) for every face whose tile bbox holds this tile, in face order; returns the count.  Two levels, both with
	// their loads in flight together (the scan is a chain of memory latencies, not of bytes): the bboxes of 256 runs of 64 faces, four
	// per lane; then the faces of the runs that reach the tile, eight runs per turn.
	auto scan = [&](auto&& fn) __attribute__((always_inline)) {
		int cnt = 0;
		for (int rb0 = 0; rb0 < n_runs; rb0 += 256) {
			uint32_t t4[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) { const int r = rb0 + u * 64 + lane; t4[u] = r < n_runs ? tbbp[r] : TB_EMPTY; }
			int nr = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const bool h = tile_hit(t4[u], tile_x, tile_y);
				const unsigned long long hm = __ballot(h);
				if (h) runs[nr + (int)__popcll(hm & lt)] = rb0 + u * 64 + lane;
				nr += (int)__popcll(hm);
			}
			wave_lds_sync();
			for (int j0 = 0; j0 < nr; j0 += 8) {
				uint32_t tv[8]; float zv[8]; int fi[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					fi[u] = j0 + u < nr ? runs[j0 + u] * 64 + lane : -1;
					const bool ok = fi[u] >= 0 && fi[u] < F;
					tv[u] = ok ? tbp[fi[u]] : TB_EMPTY; zv[u] = ok ? fzp[fi[u]] : 0.f;
				}
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					if (j0 + u >= nr) break;   // uniform
					const bool h = tile_hit(tv[u], tile_x, tile_y);
					const unsigned long long hm = __ballot(h);
					if (h) fn((uint32_t)fi[u] | ((uint32_t)slab_of(zv[u], zlo, inv_w) << 24), cnt + (int)__popcll(hm & lt));
					cnt += (int)__popcll(hm);
				}
			}
			wave_lds_sync();
		}
		return cnt;
	};
	if (!FIND_ABL(a.ablate, 32)) n = scan([&](uint32_t e, int pos) { if (pos < BIN_CAP) src[pos] = e; });
	if (n == 0) {
		// nothing can touch this tile: its outputs are the background
		if (lane == 0) a.tinfo[t_id] = make_int2(0, 0);
		write_background(a, img, tile_x, tile_y, lane);
		return;
	}
	int off = 0;
	if (lane == 0) off = atomicAdd(&a.cursor[img * CURSOR_STRIDE], n);
	off = __shfl(off, 0, 64);
	// (bit 256 of the profiling switch shrinks the pool to 512 entries per image: nearly every tile then takes the no-room path -- the rasteriser
	//  scans the faces itself --, which real scenes reach only with faces tens of tiles wide: tests/test_gpu_render.py compares the two)
	const bool room = (int64_t)off + n <= ((a.ablate & 256) ? (int64_t)512 : a.pool_cap) && off >= 0;
	uint32_t* const pool = a.pool + (int64_t)img * a.pool_cap;
	uint32_t mode = 0;
	if (!room) {
		if (lane == 0) a.tinfo[t_id] = make_int2(0, -1);
	} else if (n > BIN_CAP || (a.ablate & 16)) {
		// (a pole of a dense mesh on a small image: thousands of faces in one tile) the list goes out as the scan finds it
		scan([&](uint32_t e, int pos) { pool[off + pos] = e; });
		mode = LIST_UNSORTED;
	} else {
		// stable counting sort by slab, LDS -> pool: slab histogram, running offsets, then 64 entries at a time -- an entry's place is its
		// slab's offset plus the entries of the same slab among the lanes below it (the lanes of one slab find each other with a ballot
		// per slab bit), and the lowest lane of every slab moves the offset on for the next 64
		for (int i = lane; i < 256; i += 64) hist[i] = 0;
		wave_lds_sync();
		for (int i = lane; i < n; i += 64) __hip_atomic_fetch_add(&hist[src[i] >> 24], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		wave_lds_sync();
		{
			// exclusive prefix over the 256 bins: four consecutive bins per lane, then across the lanes
			int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
			const int own = h0 + h1 + h2 + h3;
			int inc = own;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { const int t2 = __shfl_up(inc, d, 64); if (lane >= d) inc += t2; }
			const int ex = inc - own;
			hist[4 * lane] = ex; hist[4 * lane + 1] = ex + h0; hist[4 * lane + 2] = ex + h0 + h1; hist[4 * lane + 3] = ex + h0 + h1 + h2;
		}
		wave_lds_sync();
		for (int i0 = 0; i0 < n; i0 += 64) {
			const int i = i0 + lane;
			const bool live = i < n;
			const uint32_t e = src[min(i, n - 1)];
			const uint32_t sl = e >> 24;
			unsigned long long same = __ballot(live);
#pragma unroll
			for (int bit = 0; bit < 8; ++bit) {
				const unsigned long long bm = __ballot((sl >> bit) & 1u);
				same &= ((sl >> bit) & 1u) ? bm : ~bm;
			}
			if (live) {
				const int base = hist[sl];
				pool[off + base + (int)__popcll(same & lt)] = e;
			}
			wave_lds_sync();
			if (live && (same & lt) == 0ull) hist[sl] += (int)__popcll(same);
			wave_lds_sync();
		}
	}
	if (lane == 0 && room) a.tinfo[t_id] = make_int2(off, (int)((uint32_t)n | mode));
}

// background of every tile outside its image's tile bbox (one wave per tile; the tiles inside are bin_kernel's)
__global__ __launch_bounds__(256) void outside_kernel(const TileArgs a) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int img = blockIdx.y, t = blockIdx.x * 4 + wave;
	if (t >= a.tiles_per_img) return;
	const int tile_x = t % a.tiles_x, tile_y = t / a.tiles_x;
	const int32_t* zi = a.zinfo + img * 8;
	if (tile_x >= zi[4] && tile_x <= zi[5] && tile_y >= zi[6] && tile_y <= zi[7]) return;
	if (lane == 0) a.tinfo[img * a.tiles_per_img + t] = make_int2(0, 0);
	write_background(a, img, tile_x, tile_y, lane);
}

// Persistent waves: a plain one-wave-per-tile launch leaves three quarters of the wave slots to tiles that have nothing to do (most of an
// image is empty, and a workgroup's slots stay taken until its busiest wave is through: 260 us against 120).  The waves are dealt out to
// the images (wave w: image w mod n_img), and the waves of an image stride over the tiles of its bbox -- no counter to ask (a counter per
// image cost 90 us in atomics alone).
__global__ __launch_bounds__(256) void bin_kernel(const TileArgs a, int n_img) {
	__shared__ uint32_t la[4][BIN_CAP];
	__shared__ int lh[4][256];
	__shared__ int lr[4][256];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long lt = (1ull << lane) - 1ull;
	const int F = a.F, n_runs = (F + 63) >> 6;
	const int gw = blockIdx.x * 4 + wave, n_waves = gridDim.x * 4;
	// images handled by this wave: gw mod n_img, and -- more images than waves -- every n_waves-th after it
	for (int img = gw % n_img; img < n_img; img += max(n_waves, n_img)) {
		const int32_t* zi = a.zinfo + img * 8;
		const int bx0 = zi[4], bx1 = zi[5], by0 = zi[6], by1 = zi[7];
		const int nbx = bx1 - bx0 + 1, n_in = bx1 >= bx0 ? nbx * (by1 - by0 + 1) : 0;
		const int per_img = max(n_waves / n_img, 1);   // waves that share this image
		for (int tq = gw / n_img; tq < n_in; tq += per_img) {
			const int tile_x = bx0 + tq % nbx, tile_y = by0 + tq / nbx;
			bin_tile(a, la[wave], lh[wave], lr[wave], img, tile_y * a.tiles_x + tile_x, tile_x, tile_y, lane, lt, F, n_runs);
		}
	}
}

// Tiles with a list, the longest lists first (a tile's cost grows with its list; taken in image order, the expensive tiles of the last
// images arrived when nothing was left to overlap them with): a counting sort by log2(length) in two launches.  Workgroups count their
// 1024 tiles per class in LDS and add the counts to the global ones; then every workgroup reserves, per class, a range for its own
// tiles behind the ranges reserved before it and writes them there (the order inside a class does not matter: tiles are independent).
__device__ __forceinline__ int list_class(int2 ti) {   // 0 = longest; -1: no list
	if (ti.y == 0) return -1;
	if (ti.y < 0) return 0;   // no room in the pool: the rasteriser walks every face of the image
	const int n = (int)((uint32_t)ti.y & ~LIST_UNSORTED);
	return max(0, N_QUEUES - 1 - (31 - __builtin_clz(n)));
}
__global__ __launch_bounds__(1024) void order_count_kernel(const int2* __restrict__ tinfo, int n_tiles, int32_t* __restrict__ qn) {
	__shared__ int h[N_QUEUES];
	if (threadIdx.x < N_QUEUES) h[threadIdx.x] = 0;
	__syncthreads();
	const int t = blockIdx.x * 1024 + threadIdx.x;
	const int c = t < n_tiles ? list_class(tinfo[t]) : -1;
	if (c >= 0) atomicAdd(&h[c], 1);
	__syncthreads();
	if (threadIdx.x < N_QUEUES && h[threadIdx.x]) atomicAdd(&qn[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(1024) void order_fill_kernel(const int2* __restrict__ tinfo, int n_tiles, int32_t* __restrict__ qn, int32_t* __restrict__ order) {
	__shared__ int h[N_QUEUES];
	__shared__ int base[N_QUEUES];
	if (threadIdx.x < N_QUEUES) h[threadIdx.x] = 0;
	__syncthreads();
	const int t = blockIdx.x * 1024 + threadIdx.x;
	const int c = t < n_tiles ? list_class(tinfo[t]) : -1;
	int rank = 0;
	if (c >= 0) rank = atomicAdd(&h[c], 1);
	__syncthreads();
	if (threadIdx.x < N_QUEUES) {
		int start = 0;
		for (int q = 0; q < threadIdx.x; ++q) start += qn[q];
		base[threadIdx.x] = start + (h[threadIdx.x] ? atomicAdd(&qn[32 + threadIdx.x], h[threadIdx.x]) : 0);
		if (blockIdx.x == 0 && threadIdx.x == N_QUEUES - 1) qn[62] = start + qn[N_QUEUES - 1];
	}
	__syncthreads();
	if (c >= 0) order[base[c] + rank] = t;
}

// ------------------------------------------------------------------------------------------------ 4. rasteriser
// Persistent waves, each a worker of its own: tile from the queues -> list from the pool, 64 faces at a time (the next 64 records are on
// their way from memory while these are evaluated) -> every lane evaluates the staged records for its own pixel.  Silhouette candidates
// multiply into the pixel's alpha and are appended (depth, 1 - p) to the lane's own list in the wave's scratch through an
// LDS ring (whole 32-byte pieces, written once; piece k of the 64 lanes side by side).  The list is in depth-slab order: after a batch, every face still to come lies behind
// `front`, so a pixel that holds K candidates in front of `front` has its K nearest, and a pixel whose nearest inside fragment lies in
// front of it has its colour -- the wave leaves the list when every pixel is finished.  A pixel that ends with more than K candidates
// finds the K-th smallest depth of its list by a lane-parallel radix search and blends the K nearest; where candidates tied at that depth
// were left out the pixel is queued for tie_fix_kernel (PyTorch3D's K-buffer keeps the lower face indices: needs ids, which lists omit).
struct RasterArgs {   // (what the rasteriser's loop needs and no more: the shading tail and its dozen pointers live in shade_kernel)
	float sil_blur_radius, sil_sigma;
	int sil_faces_per_pixel, image_h, image_w;
	const uint32_t* tb;
	const int32_t* zinfo;
	const int2* tinfo;
	int F, tiles_x, tiles_per_img;
	int64_t pool_cap;
	float* mask;
	int32_t* p2f_ws;      // null: no colour pass
	float* frag_ws;
	int32_t* flags;
	const int32_t* qn;
	float* zthr;
	float* alpha_ws;
	float2* scratch;
	Fix* fix;
	int ablate;
};

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool want_sil, bool want_rgb>
__global__ __launch_bounds__(256) void raster_kernel(const RasterArgs a, const FaceRec* __restrict__ recs, const uint32_t* __restrict__ pool_all,
													  const int32_t* __restrict__ order) {
	// the batch in flight, as PAIRS of faces: 32 blocks of 64 (+ 4: PAIR_STRIDE) floats per wave, field j of the pair's faces at [2 j], [2 j + 1]
	// (eval_pair); the K-nearest search's counters live on top of it afterwards
	__shared__ __attribute__((aligned(16))) float rec[4][32 * PAIR_STRIDE];
	// write-combining rings of the candidate lists: [slot][thread], so that a wave's appends (different slots per lane) never conflict
	__shared__ float ring_z[RING][256];
	__shared__ float ring_q[RING][256];

	const int H = a.image_h, W = a.image_w;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const float blur = a.sil_blur_radius;
	const float inv_sigma = 1.0f / a.sil_sigma;
	const int K = a.sil_faces_per_pixel;
	const bool early = !(a.ablate & 8);
	const int n_order = a.qn[62];
	// Candidate lists: every pixel (thread) owns up to KN_CAP depths and KN_CAP (1 - p) values.  A lane collects RING candidates in LDS and
	// writes them out as whole 32-byte pieces (two 16-byte stores per array): the lists reach memory as full sectors, once.
	// Layout of a wave's lists: 32-byte piece k of lane l at piece index k * 64 + l -- the pieces of the 64 lanes side by side.  Lanes of
	// a tile fill their pieces at about the same pace, so L2 completes 128-byte lines out of four lanes' pieces before it evicts them, and
	// the K-nearest pass reads "16 bytes of every lane's piece k" as one coalesced 2-KB access (a lane-contiguous run of KN_CAP entries
	// per lane put every 32-byte piece into a DRAM row of its own and every list read into 64 different lines).
	float4* const wz4 = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.scratch + (int64_t)blockIdx.x * KN_CAP * 256) + wave * (KN_CAP * 64));
	float4* const wq4 = wz4 + KN_CAP * 256 / 4;
	auto lidx = [&](int j) __attribute__((always_inline)) { return ((j >> 1) * 64 + lane) * 2 + (j & 1); };   // 16-byte word j of this lane's list
	auto flush_ring = [&](int base, int valid) __attribute__((always_inline)) {   // ring -> list entries [base, base + RING); slots >= valid become (+inf, 1): never selected
		float zv[RING], qv[RING];
#pragma unroll
		for (int u = 0; u < RING; ++u) {
			zv[u] = u < valid ? ring_z[u][tid] : INFINITY;
			qv[u] = u < valid ? ring_q[u][tid] : 1.0f;
		}
#pragma unroll
		for (int u = 0; u < RING; u += 4) {
			wz4[lidx((base + u) >> 2)] = make_float4(zv[u], zv[u + 1], zv[u + 2], zv[u + 3]);
			wq4[lidx((base + u) >> 2)] = make_float4(qv[u], qv[u + 1], qv[u + 2], qv[u + 3]);
		}
	};

	// (Measured and dropped: every wave's first tile by its own number instead of from the counter -- the 4096 first requests queue up
	// for ~48 us on that one address -- made the kernel 0.4 ms SLOWER: the staggered start spreads the longest lists, which all sit at
	// the head of the order, over time and over the CUs.)
	for (;;) {
		int t_q = 0;
		if (lane == 0) t_q = atomicAdd(&a.flags[3], 1);
		t_q = __builtin_amdgcn_readfirstlane(t_q);
		if (t_q >= n_order) break;
		const int t_id = __builtin_amdgcn_readfirstlane(order[t_q]);
		const int img = t_id / a.tiles_per_img, tile = t_id - img * a.tiles_per_img;
		const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
		const int xi = tile_x * T8 + (lane & 7), yi = tile_y * T8 + (lane >> 3);
		const bool in_img = xi < W && yi < H;
		const float px = 1.0f - (2.0f * xi + 1.0f) / (float)W;
		const float py = 1.0f - (2.0f * yi + 1.0f) / (float)H;
		int2 ti = a.tinfo[t_id];
		ti.x = __builtin_amdgcn_readfirstlane(ti.x); ti.y = __builtin_amdgcn_readfirstlane(ti.y);
		const bool binned = ti.y >= 0;
		const int n_list = binned ? (int)((uint32_t)ti.y & ~LIST_UNSORTED) : a.F;
		const bool sorted = binned && !((uint32_t)ti.y & LIST_UNSORTED);
		float zlo = 0.f, sw = 1.f;
		slab_layout(a.zinfo + img * 8, &zlo, &sw);
		const uint32_t* tbp = a.tb + (int64_t)img * a.F;
		const FaceRec* rp_img = recs + (int64_t)img * a.F;
		const uint32_t* lp = pool_all + (int64_t)img * a.pool_cap + ti.x;

		float alpha = 1.0f, z_lo = INFINITY, z_hi = 0.0f;
		int cnt = 0, c_lt = 0, c_a = 0, c_b = 0;   // candidates in front of `front`, within one slab behind it, further back (never beyond two)
		int n_eval = 0;
		float bz = INFINITY, bd = 0.f, bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
		int bf = -1;
		float front = 0.0f, front1 = 0.0f;   // every face not yet evaluated has its fragments behind `front`; front1: one slab further
		int s_front = -2;                   // slab whose lower edge `front` is
		bool stopped = false;               // the wave left the list early

		// entry of this lane in batch b: face index (or -1); *slab0: the slab of the batch's first entry
		auto entry = [&](int b, int* slab0) __attribute__((always_inline)) -> int {
			const int i = b * 64 + lane;
			int f = -1;
			uint32_t e0 = 0u;
			if (binned) {
				if (i < n_list) f = (int)(lp[i] & FACE_MASK);
				if (b * 64 < n_list) e0 = lp[b * 64];
			} else if (i < n_list && tile_hit(tbp[i], tile_x, tile_y)) {
				f = i;   // no room in the pool for this tile's list: every face of the image, tested here
			}
			*slab0 = (int)(e0 >> 24);
			return f;
		};
		const int n_batches = (n_list + 63) >> 6;
		int slab_next = 0, slab_cur = 0;
		int f_cur = entry(0, &slab_cur);
		int f_next = n_batches > 1 ? entry(1, &slab_next) : -1;
		for (int b = 0; b < n_batches; ++b) {
			const bool more = b + 1 < n_batches;
			const int slab_after = slab_next;
			const int f_lane = f_cur;
			if (more) {
				f_cur = f_next;
				f_next = b + 2 < n_batches ? entry(b + 2, &slab_next) : -1;   // (the entries two batches ahead are on their way while this one is evaluated)
			}
			// a pixel that holds its K nearest (and its colour) in front of everything from this batch on needs nothing more
			const bool fin = !in_img || ((!want_sil || c_lt >= K) && (!want_rgb || bz < front));
			const bool need = (early ? !fin : in_img) && !FIND_ABL(a.ablate, 4);
			// What is still to come AFTER this batch lies behind the lower edge of the next batch's first slab.  The candidates seen so far
			// come from faces of slabs up to that one, so none lies two slabs behind the old edge: when the edge moves by one slab, those
			// within a slab of the old one are now in front, the rest are within a slab of the new one; by more, all are in front.
			if (!more) { c_lt += c_a + c_b; c_a = c_b = 0; front = front1 = INFINITY; }
			else if (sorted && slab_after > s_front) {
				if (slab_after == s_front + 1) { c_lt += c_a; c_a = c_b; } else { c_lt += c_a + c_b; c_a = 0; }
				c_b = 0;
				s_front = slab_after;
				front = slab_front(slab_after, zlo, sw); front1 = slab_front(slab_after + 1, zlo, sw);
			}
			// stage the batch: the faces present are packed to the front (the no-room path tests every face of the image: most lanes hold
			// none) and every lane scatters its face's 32 dwords into its half of a pair block; then every lane reads all the blocks
			const unsigned long long m_have = __ballot(f_lane >= 0);
			const int nb = (int)__popcll(m_have);
			if (f_lane >= 0) {
				const int pos = (int)__popcll(m_have & ((1ull << lane) - 1ull));
				const float4* src = reinterpret_cast<const float4*>(rp_img + f_lane);
				float4 q[REC_F4];
#pragma unroll
				for (int k = 0; k < REC_F4; ++k) q[k] = src[k];
				float* d = &rec[wave][(pos >> 1) * PAIR_STRIDE + (pos & 1)];
#pragma unroll
				for (int k = 0; k < REC_F4; ++k) { d[8 * k] = q[k].x; d[8 * k + 2] = q[k].y; d[8 * k + 4] = q[k].z; d[8 * k + 6] = q[k].w; }
			}
			wave_lds_sync();
			for (int t = 0; 2 * t < nb; ++t) {   // two faces per turn, in packed arithmetic
				const float* blk = &rec[wave][t * PAIR_STRIDE];
				const bool two = 2 * t + 1 < nb;   // (an odd batch: the last block's second half is stale, and masked out)
				// nobody who still needs faces lies inside either bbox: next (the far side of a closed surface goes by like this)
				const float4 bx4 = *reinterpret_cast<const float4*>(blk + 56), by4 = *reinterpret_cast<const float4*>(blk + 60);
				const bool inb_a = need & (px <= bx4.z) & (px >= bx4.x) & (py <= by4.z) & (py >= by4.x);
				const bool inb_b = two & need & (px <= bx4.w) & (px >= bx4.y) & (py <= by4.w) & (py >= by4.y);
				if (__ballot(inb_a | inb_b) == 0ull) continue;
				if (FIND_ABL(a.ablate, 64)) n_eval += two ? 2 : 1;
				Frag2 fr2;
				eval_pair(blk, px, py, &fr2);   // (every lane: the ones outside the bboxes compute along and are masked out below)
				const float2 fid = *reinterpret_cast<const float2*>(blk + 54);   // the two face indices
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const bool inb = u ? inb_b : inb_a;
					const bool f_inside = u ? fr2.inside_b : fr2.inside_a;
					const float f_pzc = u ? fr2.pz_clip.y : fr2.pz_clip.x, f_pz = u ? fr2.pz.y : fr2.pz.x, f_dist = u ? fr2.dist.y : fr2.dist.x;
					if (want_sil) {
						const bool cand = inb & (f_pzc >= 0.f) & (f_inside | (f_dist < blur));
						if (cand) {
							const float prob = silhouette_prob(f_inside ? -f_dist : f_dist, inv_sigma);
							alpha *= (1.0f - prob);
							if (cnt < KN_CAP && !FIND_ABL(a.ablate, 1)) {
								const int slot = cnt & (RING - 1);
								ring_z[slot][tid] = f_pzc; ring_q[slot][tid] = 1.0f - prob;
								if (slot == RING - 1) flush_ring(cnt - (RING - 1), RING);
							}
							++cnt;
							c_lt += f_pzc < front ? 1 : 0;
							c_a += ((f_pzc >= front) & (f_pzc < front1)) ? 1 : 0;
							c_b += f_pzc >= front1 ? 1 : 0;
							z_lo = fminf(z_lo, f_pzc); z_hi = fmaxf(z_hi, f_pzc);  // depth range of the candidates (bounds of the radix search)
						}
					}
					if (want_rgb) {
						// nearest inside fragment; equal depths: the lower face index (PyTorch3D's insertion order)
						const int f_id = __float_as_int(u ? fid.y : fid.x);
						if (inb & f_inside & (f_pz >= 0.f) & ((f_pz < bz) | ((f_pz == bz) & (f_id < bf)))) {
							bz = f_pz; bf = f_id; bd = -f_dist;
							bw0 = u ? fr2.w0.y : fr2.w0.x; bw1 = u ? fr2.w1.y : fr2.w1.x; bw2 = u ? fr2.w2.y : fr2.w2.x;
						}
					}
				}
			}
			wave_lds_sync();   // the records are overwritten by the next batch
			if (early && more) {
				const bool fin2 = !in_img || ((!want_sil || c_lt >= K) && (!want_rgb || bz < front));
				if (__ballot(!fin2) == 0ull) { stopped = true; break; }
			}
		}

		// ---- K-nearest rule for the pixels that collected more than K candidates.  Lane-parallel and exact: every such
		// lane finds the K-th smallest depth of its OWN list (16-byte reads of its pieces) by a radix search
		// on the integer image of the depth (non-negative floats order like their bit patterns) with one counting pass per
		// step, then blends the candidates in front of it and as many of those AT it as still fit.
		float thr = INFINITY;
		if (want_sil) {
			// a wave that left early has candidates it never looked at behind `front`: its pixels hold >= K in front of it, and the bound
			// handed to the backward must keep those out even when the pixel holds exactly K
			if (stopped && in_img) thr = front;
			const bool over = in_img && cnt > K && !FIND_ABL(a.ablate, 2);
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
				if (over && cnt <= KN_CAP) {
					// the candidates still in the ring join the list, padded to a whole piece with (+inf, 1) entries that no pass selects
					if (cnt & (RING - 1)) flush_ring(cnt & ~(RING - 1), cnt & (RING - 1));
					const float4* z4 = wz4;
					const float4* q4 = wq4;
					const int n4 = (cnt + 3) >> 2;   // 16-byte pieces to read (the padding of the last one is inert)
					// one pass over the lane's depths, KU entries in flight, fn(bits of the depth)
					auto scan_z = [&](auto&& fn) {
						int i = 0;
						for (; i + KU / 4 <= n4; i += KU / 4) {
							float4 v[KU / 4];
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) v[u] = z4[lidx(i + u)];
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) {
								fn(__float_as_uint(v[u].x + 0.0f)); fn(__float_as_uint(v[u].y + 0.0f));
								fn(__float_as_uint(v[u].z + 0.0f)); fn(__float_as_uint(v[u].w + 0.0f));
							}
						}
						for (; i < n4; ++i) {
							const float4 v = z4[lidx(i)];
							fn(__float_as_uint(v.x + 0.0f)); fn(__float_as_uint(v.y + 0.0f)); fn(__float_as_uint(v.z + 0.0f)); fn(__float_as_uint(v.w + 0.0f));
						}
					};
					unsigned lo = __float_as_uint(z_lo + 0.0f), hi = __float_as_uint(z_hi + 0.0f);
					// Radix search for the K-th smallest depth.  Invariant: every candidate lies in [lo, hi] or was counted in c_lo
					// (candidates in front of lo) or lies behind hi; the K-th smallest is inside [lo, hi].  A level histograms
					// (z - lo) >> shift into 32 bins in ONE read of the list and keeps the bin
					// that holds the K-th; once that bin has at most 4 candidates they are fetched and ranked directly.
					int c_lo = 0;
					// the lane's 32 counters (16 bits each) live in LDS:
					// hist[w * 64 + lane], w = bin >> 1.  One fire-and-forget ds_add per candidate instead of sixteen selects.
					unsigned* const hist = reinterpret_cast<unsigned*>(&rec[wave][0]) + lane;
					while (lo < hi) {
						const unsigned span = hi - lo;
						const int shift = span < 32u ? 0 : (27 - __builtin_clz(span));  // (span >> shift) <= 31
#pragma unroll
						for (int w = 0; w < 16; ++w) hist[w * 64] = 0u;
						scan_z([&](unsigned zb) {
							if (zb < lo || zb > hi) return;
							const unsigned bin = (zb - lo) >> shift;
							__hip_atomic_fetch_add(hist + (bin >> 1) * 64, 1u << ((bin & 1u) * 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						});
						// the bin that holds the (K - c_lo)-th candidate of the range
						const int need = K - c_lo;
						int acc = 0, sel = 31, in_sel = 0;
						bool found = false;
#pragma unroll
						for (int w = 0; w < 16; ++w) {
							const unsigned hw = hist[w * 64];
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
							int mm = 0;
							scan_z([&](unsigned zb) {
								// (selects, not an if-chain: the compiler turned that into a four-element array in scratch)
								const bool in = zb >= lo && zb <= hi;
								c0 = (in && mm == 0) ? zb : c0; c1 = (in && mm == 1) ? zb : c1;
								c2 = (in && mm == 2) ? zb : c2; c3 = (in && mm == 3) ? zb : c3;
								mm += in ? 1 : 0;
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
					const int keep = K - c_lo;   // how many of the candidates AT the K-th depth belong to the K nearest
					int ties = keep;
					float a_lt = 1.0f, a_tie = 1.0f;
					unsigned nxt = 0x7F800000u;   // bits of the nearest depth BEHIND the K-th
					bool excess = false;          // more candidates at the K-th depth than are kept
					{
						auto take = [&](float z, float q) {
							const unsigned zb = __float_as_uint(z + 0.0f);
							if (zb < lo) a_lt *= q;
							else if (zb == lo) {
								if (ties > 0) { a_tie *= q; --ties; }
								else excess = true;
							} else nxt = min(nxt, zb);
						};
						int i = 0;
						for (; i + KU / 4 <= n4; i += KU / 4) {
							float4 zv[KU / 4], qv[KU / 4];
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) { zv[u] = z4[lidx(i + u)]; qv[u] = q4[lidx(i + u)]; }
#pragma unroll
							for (int u = 0; u < KU / 4; ++u) {
								take(zv[u].x, qv[u].x); take(zv[u].y, qv[u].y); take(zv[u].z, qv[u].z); take(zv[u].w, qv[u].w);
							}
						}
						for (; i < n4; ++i) {
							const float4 zv = z4[lidx(i)], qv = q4[lidx(i)];
							take(zv.x, qv.x); take(zv.y, qv.y); take(zv.z, qv.z); take(zv.w, qv.w);
						}
					}
					alpha = a_lt * a_tie;
					// The bound the backward compares a candidate's depth with: the MIDPOINT between the K-th depth and the next one behind it
					// (never beyond `front` where the wave left early: what it did not look at lies behind that).  Where candidates tied AT
					// the K-th depth were left out, which of them stay is decided by face index: tie_fix_kernel rewrites this pixel.
					const float zk = __uint_as_float(lo);
					thr = fminf(0.5f * (zk + __uint_as_float(nxt)), stopped ? front : INFINITY);
					if (excess) {
						thr = zk;   // provisional (every tied candidate): replaced by -zk and the face id of the last one kept
						Fix fx;
						fx.pix = (int32_t)(((int64_t)img * H + yi) * W + xi); fx.zk = lo; fx.keep = keep; fx.a_lt = a_lt;
						a.fix[atomicAdd(&a.flags[7], 1)] = fx;
					}
				}
			}
		}

		if (FIND_ABL(a.ablate, 64)) {  // diagnostics: [24] (pixel, face) tests issued (64-lane slots, in units of 64), [25] silhouette candidates (units of 64)
			int te = n_eval, tc = in_img ? cnt : 0;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { te += __shfl_xor(te, d, 64); tc += __shfl_xor(tc, d, 64); }
			if (lane == 0 && te) { atomicAdd(&a.flags[24], (te + 32) >> 6); atomicAdd(&a.flags[25], (tc + 32) >> 6); }
			if (lane == 0 && stopped) atomicAdd(&a.flags[26], 1);   // [26] tiles left early
		}
		if (in_img) {
			const int64_t pix = ((int64_t)img * H + yi) * W + xi;
			if (want_sil) {
				a.mask[pix] = 1.0f - alpha;
				a.zthr[pix] = thr;
				a.alpha_ws[pix] = alpha;
			}
			if (want_rgb) {   // the nearest inside fragment: shade_kernel turns it into the pixel's colour
				a.p2f_ws[pix] = bf;
				*reinterpret_cast<float4*>(a.frag_ws + pix * 8) = make_float4(bw0, bw1, bw2, bz);
				a.frag_ws[pix * 8 + 4] = bd;
			}
		}
		wave_lds_sync();  // the next tile reuses the rings
	}
}

#include "render_band.h"


// Phong shading + softmax blend (K = 1) of every pixel's nearest inside fragment, the user-visible pix_to_face / zbuf, and the
// barycentrics the RGB backward reads: one thread per pixel of the tiles that have a list (the others got their background from bin_kernel).
__global__ __launch_bounds__(256) void shade_kernel(const TileArgs a) {
	const int H = a.rp.image_h, W = a.rp.image_w;
	const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (pix >= (int64_t)a.total_tiles / a.tiles_per_img * H * W) return;
	const int xi = (int)(pix % W), yi = (int)((pix / W) % H), img = (int)(pix / ((int64_t)W * H));
	if (a.tinfo[img * a.tiles_per_img + (yi / T8) * a.tiles_x + xi / T8].y == 0) return;
	const int bf = a.p2f_ws[pix];
	const float4 fw = *reinterpret_cast<const float4*>(a.frag_ws + pix * 8);
	const float bw0 = fw.x, bw1 = fw.y, bw2 = fw.z, bz = fw.w, bd = a.frag_ws[pix * 8 + 4];
	a.bary_ws[pix * 3 + 0] = bw0; a.bary_ws[pix * 3 + 1] = bw1; a.bary_ws[pix * 3 + 2] = bw2;
	if (a.p2f_out) a.p2f_out[pix] = bf < 0 ? -1 : img * a.F + bf;
	if (a.zbuf_out) a.zbuf_out[pix] = bf < 0 ? -1.0f : bz;
	if (!a.image) return;
	float* o = a.image + pix * 3;
	if (bf < 0) {
		o[0] = a.rp.background[0]; o[1] = a.rp.background[1]; o[2] = a.rp.background[2];
		return;
	}
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

// ------------------------------------------------------------------------------------------------ 5. ties at the K-th depth
// PyTorch3D inserts fragments into a pixel's K-buffer in face order and a later fragment of EQUAL depth does not displace an earlier
// one: of the candidates tied at the K-th depth, the `keep` lowest face indices stay.  (Ties are not exotic: a pixel outside a fan of
// faces that share their nearest vertex gets that vertex's depth from every one of them.)  One wave per queued pixel walks the tile's
// list with the face ids at hand, collects the candidates whose depth equals the K-th to the bit -- the same eval_frag, the same
// rounding --, keeps the `keep` lowest ids and rewrites mask, alpha, the bound (negated: "ties at this depth are decided by
// tie_face") and the id of the last face kept.
__global__ __launch_bounds__(256) void tie_fix_kernel(const TileArgs a, const FaceRec* __restrict__ recs) {
	constexpr int TIE_CAP = 128;
	__shared__ int tf[4][TIE_CAP];
	__shared__ float tq[4][TIE_CAP];
	__shared__ float ts[4][TIE_CAP];
	const int H = a.rp.image_h, W = a.rp.image_w;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const float blur = a.rp.sil_blur_radius;
	const float inv_sigma = 1.0f / a.rp.sil_sigma;
	const int n_fix = a.flags[7];
	const unsigned long long lt = (1ull << lane) - 1ull;
	for (int i = blockIdx.x * 4 + wave; i < n_fix; i += gridDim.x * 4) {
		const Fix fx = a.fix[i];
		const int xi = fx.pix % W, yi = (fx.pix / W) % H, img = fx.pix / (W * H);
		const float px = 1.0f - (2.0f * xi + 1.0f) / (float)W;
		const float py = 1.0f - (2.0f * yi + 1.0f) / (float)H;
		const int tile_x = xi / T8, tile_y = yi / T8;
		const int2 ti = a.tinfo[img * a.tiles_per_img + tile_y * a.tiles_x + tile_x];
		const bool binned = ti.y >= 0;
		const int n_list = binned ? (int)((uint32_t)ti.y & ~LIST_UNSORTED) : a.F;
		const uint32_t* tbp = a.tb + (int64_t)img * a.F;
		const uint32_t* lp = a.pool + (int64_t)img * a.pool_cap + ti.x;
		int n_t = 0;
		for (int i0 = 0; i0 < n_list; i0 += 64) {
			const int j = i0 + lane;
			int f = -1;
			if (j < n_list) {
				if (binned) f = (int)(lp[j] & FACE_MASK);
				else if (tile_hit(tbp[j], tile_x, tile_y)) f = j;
			}
			bool tied = false;
			float q = 1.0f;
			if (f >= 0) {
				const FaceRec r = recs[(int64_t)img * a.F + f];
				Frag fr;
				if (eval_frag(r, px, py, &fr) && fr.pz_clip >= 0.f && (fr.inside || fr.dist < blur) && __float_as_uint(fr.pz_clip + 0.0f) == fx.zk) {
					tied = true;
					q = 1.0f - silhouette_prob(fr.inside ? -fr.dist : fr.dist, inv_sigma);
				}
			}
			const unsigned long long m = __ballot(tied);
			const int pos = n_t + (int)__popcll(m & lt);
			if (tied && pos < TIE_CAP) { tf[wave][pos] = f; tq[wave][pos] = q; }
			n_t += (int)__popcll(m);
		}
		wave_lds_sync();
		n_t = min(n_t, TIE_CAP);   // (more than TIE_CAP faces of one depth at one pixel: the first TIE_CAP in list order stand for all)
		// rank by face id; the `keep` lowest, multiplied in id order
		int last = -1;
		for (int j = lane; j < n_t; j += 64) {
			const int fj = tf[wave][j];
			int rank = 0;
			for (int k = 0; k < n_t; ++k) rank += tf[wave][k] < fj;
			ts[wave][rank] = tq[wave][j];
			if (rank == fx.keep - 1) last = fj;
		}
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) last = max(last, __shfl_xor(last, d, 64));
		wave_lds_sync();
		if (lane == 0) {
			float al = fx.a_lt;
			const int nk = min(fx.keep, n_t);
			for (int k = 0; k < nk; ++k) al *= ts[wave][k];
			a.mask[fx.pix] = 1.0f - al;
			a.alpha_ws[fx.pix] = al;
			a.zthr[fx.pix] = -__uint_as_float(fx.zk);
			a.tie_face[fx.pix] = last;
		}
		wave_lds_sync();
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
__global__ __launch_bounds__(256) void sil_bwd_kernel(const find_render_params rp, const FaceRec* __restrict__ recs, const uint32_t* __restrict__ tb,
													   const int32_t* __restrict__ faces, int64_t faces_mesh_stride, int n_views, int V, int F,
													   const float* __restrict__ alpha_ws, const float* __restrict__ d_mask, const float* __restrict__ zthr,
													   const int32_t* __restrict__ tie_face, float* __restrict__ d_vproj) {
	const int img = blockIdx.y;
	const int sub = threadIdx.x & (LPF - 1);
	const int f = blockIdx.x * (256 / LPF) + threadIdx.x / LPF;
	const int64_t o = (int64_t)img * F + min(f, F - 1);
	const bool act = f < F && tb[o] != TB_EMPTY;
	const int H = rp.image_h, W = rp.image_w;
	const float blur = rp.sil_blur_radius;
	const FaceRec r = recs[o];
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
	// A lane walks pixels sub, sub + LPF, ... of the bbox in row-major order: the step is (LPF / bw rows, LPF % bw columns) -- ONE integer
	// division per face instead of one per pixel --, per-pixel addresses are 32-bit offsets from the image's base (H, W <= 2048), and the NDC
	// coordinate of a pixel centre, 1 - (2 i + 1) / S, is a multiplication where S is a power of two (the same value to the bit; the
	// rasteriser divides): the address and coordinate arithmetic was ~70 of the ~330 instructions of an iteration.
	const int bws = max(bw, 1);
	const int step_y = LPF / bws, step_x = LPF - step_y * bws;
	const int64_t base = (int64_t)img * H * W;
	const float* const dm = d_mask + base;
	const float* const zth = zthr + base;
	const float* const alw = alpha_ws + base;
	const int32_t* const tf = tie_face + base;
	const bool w_p2 = (W & (W - 1)) == 0, h_p2 = (H & (H - 1)) == 0;
	const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
	int yi = ylo + sub / bws, xi = xlo + sub % bws;
	float g = 0.f, zt = 0.f, mk = 0.f;
	int off_cur = 0;
	if (sub < npx) {
		off_cur = yi * W + xi;
		g = dm[off_cur]; zt = zth[off_cur]; mk = alw[off_cur];
	}
	for (int pi = sub; pi < npx; pi += LPF) {
		const int cy = yi, cx = xi;
		const float cg = g, cmk = mk;
		float czt = zt;
		const int coff = off_cur;
		if (pi + LPF < npx) {
			xi += step_x; yi += step_y;
			if (xi > xhi) { xi -= bws; ++yi; }
			off_cur = yi * W + xi;
			g = dm[off_cur]; zt = zth[off_cur]; mk = alw[off_cur];
		}
		const float py = h_p2 ? 1.0f - (2.0f * cy + 1.0f) * inv_h : 1.0f - (2.0f * cy + 1.0f) / (float)H;
		{
			if (cg == 0.f) continue;
			const float px = w_p2 ? 1.0f - (2.0f * cx + 1.0f) * inv_w : 1.0f - (2.0f * cx + 1.0f) / (float)W;
			Frag fr;
			if (!eval_frag(r, px, py, &fr)) continue;
			if (!(fr.pz_clip >= 0.f && (fr.inside || fr.dist < blur))) continue;
			// pixel with more than K candidates: is this one among the K nearest?  (a negative bound: candidates tied AT that depth were
			// sorted out by face index in the forward, tie_fix_kernel: those up to tie_face stay)
			if (czt < 0.f) {
				czt = -czt;
				if (fr.pz_clip == czt && f > tf[coff]) continue;
			}
			if (fr.pz_clip > czt) continue;
			const float sd = fr.inside ? -fr.dist : fr.dist;
			const float prob = silhouette_prob(sd, inv_sigma);
			const float alpha = cmk;
			// gradient w.r.t. the UNSIGNED distance: sign * dmask/dd
			const float gd = (fr.inside ? -1.0f : 1.0f) * (-cg * alpha * prob * inv_sigma);
			// nearest edge (a,b), q = a + t (b - a):  d = |q - p|^2,  dd/da = 2 (1-t) (q - p),  dd/db = 2 t (q - p)
			// (selects, no branches: edge 0: a = v0, b = v1;  edge 1: a = v0, b = v2;  edge 2: a = v1, b = v2)
			const bool e0 = fr.edge == 0, e2 = fr.edge == 2;
			const float qx = fr.qx, qy = fr.qy;   // (q - p comes with the fragment)
			const float ga = gd * 2.0f * (1.0f - fr.t), gb = gd * 2.0f * fr.t;
			const float wax = ga * qx, way = ga * qy, wbx = gb * qx, wby = gb * qy;
			g0x += e2 ? 0.f : wax; g0y += e2 ? 0.f : way;
			g1x += e0 ? wbx : (e2 ? wax : 0.f); g1y += e0 ? wby : (e2 ? way : 0.f);
			g2x += e0 ? 0.f : wbx; g2y += e0 ? 0.f : wby;
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
	// (one division per denominator, then multiplications: an IEEE division is ten instructions, and there were thirty of them per pixel)
	const float inv_nl = 1.0f / nl;
	const float n[3] = {nrm[0] * inv_nl, nrm[1] * inv_nl, nrm[2] * inv_nl};
	float Lv[3] = {rp.light_pos[0] - pos[0], rp.light_pos[1] - pos[1], rp.light_pos[2] - pos[2]};
	const float ll = fmaxf(sqrtf(Lv[0] * Lv[0] + Lv[1] * Lv[1] + Lv[2] * Lv[2]), 1e-6f);
	const float inv_ll = 1.0f / ll;
	const float l[3] = {Lv[0] * inv_ll, Lv[1] * inv_ll, Lv[2] * inv_ll};
	const float cosang = n[0] * l[0] + n[1] * l[1] + n[2] * l[2];
	float Vv[3] = {cam[view * 3] - pos[0], cam[view * 3 + 1] - pos[1], cam[view * 3 + 2] - pos[2]};
	const float vl = fmaxf(sqrtf(Vv[0] * Vv[0] + Vv[1] * Vv[1] + Vv[2] * Vv[2]), 1e-6f);
	const float inv_vl = 1.0f / vl;
	const float vd[3] = {Vv[0] * inv_vl, Vv[1] * inv_vl, Vv[2] * inv_vl};
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
	auto unnorm = [](const float* u, const float* du, float inv_len, float* dx) {
		const float dot = u[0] * du[0] + u[1] * du[1] + u[2] * du[2];
		for (int c = 0; c < 3; ++c) dx[c] = (du[c] - u[c] * dot) * inv_len;
	};
	float d_nrm[3], d_Lv[3], d_Vv[3];
	unnorm(n, d_n, inv_nl, d_nrm);
	unnorm(l, d_l, inv_ll, d_Lv);
	unnorm(vd, d_vd, inv_vl, d_Vv);
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
	const float inv_area = 1.0f / area;
	const float w0 = e0 * inv_area, w1 = e1 * inv_area, w2 = e2 * inv_area;
	const float t0 = w0 * z1 * z2, t1 = z0 * w1 * z2, t2 = z0 * z1 * w2;
	const float den = t0 + t1 + t2;
	if (!(den > KEPS)) return;  // (degenerate fragment: only the interpolation part contributes)
	const float inv_den = 1.0f / den;
	const float sdb = (d_bw[0] * t0 + d_bw[1] * t1 + d_bw[2] * t2) * inv_den;
	const float d_t0 = (d_bw[0] - sdb) * inv_den, d_t1 = (d_bw[1] - sdb) * inv_den, d_t2 = (d_bw[2] - sdb) * inv_den;
	const float d_w0 = d_t0 * z1 * z2, d_w1 = d_t1 * z0 * z2, d_w2 = d_t2 * z0 * z1;
	const float d_z0 = d_t1 * w1 * z2 + d_t2 * z1 * w2;
	const float d_z1 = d_t0 * w0 * z2 + d_t2 * z0 * w2;
	const float d_z2 = d_t0 * w0 * z1 + d_t1 * z0 * w1;
	// w_i = e_i / area
	const float d_e0 = d_w0 * inv_area, d_e1 = d_w1 * inv_area, d_e2 = d_w2 * inv_area;
	const float d_area = -(d_w0 * w0 + d_w1 * w1 + d_w2 * w2) * inv_area;
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
	FIND_REQUIRE(V >= 1 && F >= 1 && V < (1ll << 28) && F < (1ll << 24), "find_render: bad mesh size (a tile list entry holds the face index in 24 bits)");
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
	(void)hipMemsetAsync(w.flags, 0, (64 + 64) * sizeof(int32_t) + n_img * CURSOR_STRIDE * sizeof(int32_t), s);   // flags, the class counts behind them, the pool cursors behind those
	hipLaunchKernelGGL(project_kernel, dim3((unsigned)cdiv(V, 256), (unsigned)n_img), dim3(256), 0, s, verts, R, T, sc, (int)n_views, V, w.vproj);
	// the silhouette's blur margin is a superset of the RGB pass's (blur 0); one set of lists serves both
	const float blur = mask ? rp->sil_blur_radius : 0.0f;
	const int tiles_x = (int)cdiv(W, T8), tiles_per_img = tiles_x * (int)cdiv(H, T8);
	const int n_runs = (int)cdiv(F, 64);
	hipLaunchKernelGGL(face_setup_kernel, dim3((unsigned)cdiv(F, 256), (unsigned)n_img), dim3(256), 0, s, w.vproj, faces, fstride, (int)n_views, V, F, H, W,
					   blur, rp->z_clip, w.frec, w.recs, w.tb, w.tbb, w.fzmin, w.rz, w.flags);
	hipLaunchKernelGGL(zinfo_kernel, dim3((unsigned)n_img), dim3(256), 0, s, w.tbb, w.rz, n_runs, w.zinfo);
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
	a.tb = w.tb; a.tbb = w.tbb; a.fzmin = w.fzmin; a.zinfo = w.zinfo; a.faces = faces; a.faces_mesh_stride = fstride;
	a.verts = verts; a.normals = w.normals; a.colors = vert_colors; a.cam = cam;
	a.n_views = (int)n_views; a.V = V; a.F = F; a.tiles_x = tiles_x;
	a.mask = mask; a.image = image; a.p2f_out = pix_to_face; a.zbuf_out = zbuf;
	a.p2f_ws = (image || pix_to_face || zbuf) ? w.p2f : nullptr; a.bary_ws = w.bary; a.frag_ws = w.frag; a.flags = w.flags; a.qn = w.qn; a.cursor = w.cursor;
	a.zthr = w.zthr; a.alpha_ws = w.alpha; a.tie_face = w.tie_face; a.scratch = w.scratch;
	a.tinfo = w.tinfo; a.pool = w.pool; a.order = w.order; a.fix = w.fix; a.pool_cap = w.pool_cap;
	a.tiles_per_img = tiles_per_img; a.total_tiles = (int)(a.tiles_per_img * n_img);
	a.ablate = find::g_raster_ablate;
	hipLaunchKernelGGL(outside_kernel, dim3((unsigned)cdiv(tiles_per_img, 4), (unsigned)n_img), dim3(256), 0, s, a);
	hipLaunchKernelGGL(bin_kernel, dim3((unsigned)std::max<int64_t>(std::min<int64_t>(cdiv(a.total_tiles, 4), 1024), cdiv(n_img, 4))), dim3(256), 0, s, a, (int)n_img);
	hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)cdiv(a.total_tiles, 1024)), dim3(1024), 0, s, w.tinfo, a.total_tiles, w.qn);
	hipLaunchKernelGGL(order_fill_kernel, dim3((unsigned)cdiv(a.total_tiles, 1024)), dim3(1024), 0, s, w.tinfo, a.total_tiles, w.qn, w.order);
	RasterArgs ra;
	memset(&ra, 0, sizeof(ra));
	ra.sil_blur_radius = rp->sil_blur_radius; ra.sil_sigma = rp->sil_sigma; ra.sil_faces_per_pixel = rp->sil_faces_per_pixel; ra.image_h = H; ra.image_w = W;
	ra.tb = w.tb; ra.zinfo = w.zinfo; ra.tinfo = w.tinfo; ra.F = F; ra.tiles_x = tiles_x; ra.tiles_per_img = tiles_per_img; ra.pool_cap = w.pool_cap;
	ra.mask = mask; ra.p2f_ws = a.p2f_ws; ra.frag_ws = w.frag; ra.flags = w.flags; ra.qn = w.qn; ra.zthr = w.zthr; ra.alpha_ws = w.alpha;
	ra.scratch = w.scratch; ra.fix = w.fix; ra.ablate = a.ablate;
	// Which rasteriser: the list-free band kernel (render_band.h) from BAND_MIN_PIXELS pixels per image on, the candidate-list kernel below
	// (switch bits 2048 / 4096 of find_render_switches force the band / the list kernel at every size: tests and A/B runs)
	const bool band = (a.ablate & 2048) || (!(a.ablate & 4096) && (int64_t)H * W >= BAND_MIN_PIXELS);
	if (band) {
		RasterBandArgs rb;
		memset(&rb, 0, sizeof(rb));
		rb.sil_blur_radius = ra.sil_blur_radius; rb.sil_sigma = ra.sil_sigma; rb.sil_faces_per_pixel = ra.sil_faces_per_pixel; rb.image_h = H; rb.image_w = W;
		rb.tb = ra.tb; rb.zinfo = ra.zinfo; rb.tinfo = ra.tinfo; rb.F = ra.F; rb.tiles_x = ra.tiles_x; rb.tiles_per_img = ra.tiles_per_img; rb.pool_cap = ra.pool_cap;
		rb.mask = ra.mask; rb.p2f_ws = ra.p2f_ws; rb.frag_ws = ra.frag_ws; rb.flags = ra.flags; rb.qn = ra.qn; rb.zthr = ra.zthr; rb.alpha_ws = ra.alpha_ws;
		rb.tie_face = w.tie_face; rb.scratch = reinterpret_cast<float*>(w.scratch); rb.ablate = ra.ablate;
		if (mask && a.p2f_ws) hipLaunchKernelGGL((raster_band_kernel<true, true>), dim3((unsigned)w.raster_wgs), dim3(256), 0, s, rb, w.recs, w.pool, w.order);
		else if (mask) hipLaunchKernelGGL((raster_band_kernel<true, false>), dim3((unsigned)w.raster_wgs), dim3(256), 0, s, rb, w.recs, w.pool, w.order);
		else hipLaunchKernelGGL((raster_band_kernel<false, true>), dim3((unsigned)w.raster_wgs), dim3(256), 0, s, rb, w.recs, w.pool, w.order);
	} else if (mask && a.p2f_ws) hipLaunchKernelGGL((raster_kernel<true, true>), dim3((unsigned)w.raster_wgs), dim3(256), 0, s, ra, w.recs, w.pool, w.order);
	else if (mask) hipLaunchKernelGGL((raster_kernel<true, false>), dim3((unsigned)w.raster_wgs), dim3(256), 0, s, ra, w.recs, w.pool, w.order);
	else hipLaunchKernelGGL((raster_kernel<false, true>), dim3((unsigned)w.raster_wgs), dim3(256), 0, s, ra, w.recs, w.pool, w.order);
	if (a.p2f_ws) hipLaunchKernelGGL(shade_kernel, dim3((unsigned)cdiv(n_img * H * W, 256)), dim3(256), 0, s, a);
	if (mask) hipLaunchKernelGGL(tie_fix_kernel, dim3(256), dim3(256), 0, s, a, w.recs);
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
			hipLaunchKernelGGL(sil_bwd_kernel<32>, dim3((unsigned)cdiv(F, 8), (unsigned)n_img), dim3(256), 0, s, *rp, w.recs, w.tb, faces, fstride, (int)n_views, V, F,
							   w.alpha, d_mask, w.zthr, w.tie_face, w.d_vproj);
		else if (side * side > 160.0f)
			hipLaunchKernelGGL(sil_bwd_kernel<16>, dim3((unsigned)cdiv(F, 16), (unsigned)n_img), dim3(256), 0, s, *rp, w.recs, w.tb, faces, fstride, (int)n_views, V, F,
							   w.alpha, d_mask, w.zthr, w.tie_face, w.d_vproj);
		else
			hipLaunchKernelGGL(sil_bwd_kernel<8>, dim3((unsigned)cdiv(F, 32), (unsigned)n_img), dim3(256), 0, s, *rp, w.recs, w.tb, faces, fstride, (int)n_views, V, F,
						   w.alpha, d_mask, w.zthr, w.tie_face, w.d_vproj);
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
