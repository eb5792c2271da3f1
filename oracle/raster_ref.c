/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the mesh rasteriser + shaders the reference reaches through
 * PyTorch3D (rows a7-a10 of SURVEY.md §8): FoVPerspectiveCameras projection, rasterize_meshes NAIVE path (what
 * PyTorch3D executes on CPU: every pixel tests every face), SoftSilhouetteShader and SoftPhongShader + softmax_rgb_blend.
 *
 * PARITY UNPINNED: PyTorch3D @ 1706eb8216248e54f68cad86f7ea4125c79a3ca4 (requirements_mac_linux.txt:31) is not
 * vendored under /root/reference and cannot be installed, and the reference has no tests / golden images.  The code
 * follows PyTorch3D's published algorithm (csrc/rasterize_meshes/rasterize_meshes_cpu.cpp, csrc/utils/geometry_utils.h,
 * renderer/mesh/shading.py, renderer/lighting.py, renderer/blending.py) as recalled, anchored on the reference call
 * sites (src/model/renderer.py:113-142, 208-245, 269-311; in-repo blend math renderer.py:23-72) and on analytic
 * known-answer tests (tests/test_oracle_raster.py).
 *
 * Conventions: row-vector transforms p_view = p_world @ R + T; NDC +x left, +y up;
 * pixel (yi, xi) centre = (1 - (2*xi+1)/W, 1 - (2*yi+1)/H); z handed to the rasteriser is view-space depth.
 * Image index = mesh * n_views + view.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).  All arrays are C-contiguous float32 / int32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define K_EPS 1e-8f

typedef struct {
	int32_t image_h, image_w;
	float fov_deg, znear, zfar;
	float sil_blur_radius, sil_sigma;
	int32_t sil_faces_per_pixel;
	float rgb_sigma, rgb_gamma;
	float background[3];
	float light_pos[3];
	float ambient, diffuse, specular, shininess;
	float z_clip;
} render_params;

/* ---------------------------------------------------------------- projection (A.2) */
/* verts (n_meshes,n_verts,3) -> vproj (n_meshes*n_views, n_verts, 3) = (x_ndc, y_ndc, z_view) */
void ref_project(const render_params* rp, const float* verts, const float* R, const float* T, int n_meshes, int n_views,
				 int n_verts, float* vproj) {
	const float s = 1.0f / tanf(rp->fov_deg * (float)M_PI / 180.0f * 0.5f);
#pragma omp parallel for collapse(2)
	for (int n = 0; n < n_meshes; ++n)
		for (int m = 0; m < n_views; ++m) {
			const float* Rm = R + m * 9;
			const float* Tm = T + m * 3;
			for (int v = 0; v < n_verts; ++v) {
				const float* p = verts + ((size_t)n * n_verts + v) * 3;
				const float x = p[0] * Rm[0] + p[1] * Rm[3] + p[2] * Rm[6] + Tm[0];
				const float y = p[0] * Rm[1] + p[1] * Rm[4] + p[2] * Rm[7] + Tm[1];
				const float z = p[0] * Rm[2] + p[1] * Rm[5] + p[2] * Rm[8] + Tm[2];
				float* o = vproj + (((size_t)n * n_views + m) * n_verts + v) * 3;
				o[0] = s * x / z;
				o[1] = s * y / z;
				o[2] = z;
			}
		}
}

/* ---------------------------------------------------------------- geometry_utils.h */
static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
	return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float point_line_dist(float px, float py, float ax, float ay, float bx, float by) {
	const float bax = bx - ax, bay = by - ay;
	const float l2 = bax * bax + bay * bay;
	if (l2 <= K_EPS) return (px - bx) * (px - bx) + (py - by) * (py - by);
	float t = (bax * (px - ax) + bay * (py - ay)) / l2;
	t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
	const float qx = ax + t * bax, qy = ay + t * bay;
	return (qx - px) * (qx - px) + (qy - py) * (qy - py);
}

typedef struct {
	float z;
	int32_t f;
	float dist;
	float b0, b1, b2;
} frag_t;

/* Evaluate one (pixel, face).  Returns 1 and fills *o if the face contributes a fragment. */
static inline int pixel_face(float xf, float yf, const float* v0, const float* v1, const float* v2, float blur_radius,
							 int perspective_correct, int clip_bary, int cull_backfaces, float z_clip, frag_t* o) {
	/* z-clipping, cull-only form: faces entirely behind z_clip are dropped (clip.py); straddling faces are left
	 * whole -- FIND's cameras sit 0.3 m from a <=0.15 m object so none exists on its path. */
	if (v0[2] < z_clip && v1[2] < z_clip && v2[2] < z_clip) return 0;
	const float zmax = fmaxf(v0[2], fmaxf(v1[2], v2[2]));
	if (zmax < 0.f) return 0;
	const float br = sqrtf(blur_radius);
	const float xmin = fminf(v0[0], fminf(v1[0], v2[0])) - br, xmax = fmaxf(v0[0], fmaxf(v1[0], v2[0])) + br;
	const float ymin = fminf(v0[1], fminf(v1[1], v2[1])) - br, ymax = fmaxf(v0[1], fmaxf(v1[1], v2[1])) + br;
	if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) return 0;
	const float face_area = edge_fn(v0[0], v0[1], v1[0], v1[1], v2[0], v2[1]);
	if (cull_backfaces && face_area < 0.f) return 0;
	if (face_area <= K_EPS && face_area >= -K_EPS) return 0;
	/* barycentric coordinates */
	const float area = edge_fn(v2[0], v2[1], v0[0], v0[1], v1[0], v1[1]) + K_EPS;
	float w0 = edge_fn(xf, yf, v1[0], v1[1], v2[0], v2[1]) / area;
	float w1 = edge_fn(xf, yf, v2[0], v2[1], v0[0], v0[1]) / area;
	float w2 = edge_fn(xf, yf, v0[0], v0[1], v1[0], v1[1]) / area;
	if (perspective_correct) {
		const float t0 = w0 * v1[2] * v2[2], t1 = v0[2] * w1 * v2[2], t2 = v0[2] * v1[2] * w2;
		const float den = fmaxf(t0 + t1 + t2, K_EPS);
		w0 = t0 / den; w1 = t1 / den; w2 = t2 / den;
	}
	float c0 = w0, c1 = w1, c2 = w2;
	if (clip_bary) {
		c0 = fmaxf(w0, 0.f); c1 = fmaxf(w1, 0.f); c2 = fmaxf(w2, 0.f);
		const float sum = fmaxf(c0 + c1 + c2, 1e-5f);
		c0 /= sum; c1 /= sum; c2 /= sum;
	}
	const float pz = c0 * v0[2] + c1 * v1[2] + c2 * v2[2];
	if (pz < 0.f) return 0;
	const float e01 = point_line_dist(xf, yf, v0[0], v0[1], v1[0], v1[1]);
	const float e02 = point_line_dist(xf, yf, v0[0], v0[1], v2[0], v2[1]);
	const float e12 = point_line_dist(xf, yf, v1[0], v1[1], v2[0], v2[1]);
	const float dist = fminf(fminf(e01, e02), e12);
	const int inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
	if (!inside && dist >= blur_radius) return 0;
	o->z = pz; o->dist = inside ? -dist : dist;
	o->b0 = c0; o->b1 = c1; o->b2 = c2;
	return 1;
}

/* ---------------------------------------------------------------- rasterize_meshes, naive (A.3)
 * vproj (n_img, n_verts, 3); faces (n_faces,3) shared if faces_batch==1 else (n_meshes,n_faces,3), image i uses mesh i/n_views.
 * Outputs (n_img,H,W,K): pix_to_face (packed id img*n_faces + f, -1 empty), zbuf (-1), bary (…,3) (-1), dists (-1),
 * sorted by ascending z (ties: lower face id first). */
void ref_rasterize(const float* vproj, const int32_t* faces, int faces_batch, int n_img, int n_views, int n_verts, int n_faces,
				   int H, int W, int K, float blur_radius, int perspective_correct, int clip_bary, int cull_backfaces,
				   float z_clip, int32_t* pix_to_face, float* zbuf, float* bary, float* dists) {
#pragma omp parallel
	{
		frag_t* q = (frag_t*)malloc(sizeof(frag_t) * (size_t)(K + 1));
#pragma omp for collapse(2) schedule(dynamic, 4)
		for (int im = 0; im < n_img; ++im)
			for (int yi = 0; yi < H; ++yi) {
				const float* vp = vproj + (size_t)im * n_verts * 3;
				const int32_t* fp = faces + (faces_batch == 1 ? 0 : (size_t)(im / n_views) * n_faces * 3);
				const float yf = 1.0f - (2.0f * yi + 1.0f) / (float)H;
				for (int xi = 0; xi < W; ++xi) {
					const float xf = 1.0f - (2.0f * xi + 1.0f) / (float)W;
					int cnt = 0;
					for (int f = 0; f < n_faces; ++f) {
						if (fp[f * 3] < 0) continue; /* -1 padding of ragged batches */
						frag_t fr;
						if (!pixel_face(xf, yf, vp + 3 * fp[f * 3], vp + 3 * fp[f * 3 + 1], vp + 3 * fp[f * 3 + 2], blur_radius,
										perspective_correct, clip_bary, cull_backfaces, z_clip, &fr))
							continue;
						fr.f = f;
						/* insertion into the K nearest (ascending z; earlier face wins ties) */
						int pos = cnt;
						while (pos > 0 && q[pos - 1].z > fr.z) { if (pos < K) q[pos] = q[pos - 1]; --pos; }
						if (pos < K) { q[pos] = fr; if (cnt < K) ++cnt; }
					}
					const size_t o = (((size_t)im * H + yi) * W + xi) * K;
					for (int k = 0; k < K; ++k) {
						if (k < cnt) {
							pix_to_face[o + k] = im * n_faces + q[k].f;
							zbuf[o + k] = q[k].z; dists[o + k] = q[k].dist;
							bary[(o + k) * 3] = q[k].b0; bary[(o + k) * 3 + 1] = q[k].b1; bary[(o + k) * 3 + 2] = q[k].b2;
						} else {
							pix_to_face[o + k] = -1; zbuf[o + k] = -1.f; dists[o + k] = -1.f;
							bary[(o + k) * 3] = bary[(o + k) * 3 + 1] = bary[(o + k) * 3 + 2] = -1.f;
						}
					}
				}
			}
		free(q);
	}
}

/* ---------------------------------------------------------------- soft silhouette (A.4; renderer.py:50-54,310) */
void ref_silhouette(const int32_t* pix_to_face, const float* dists, int64_t n_pix, int K, float sigma, float* mask) {
#pragma omp parallel for
	for (int64_t p = 0; p < n_pix; ++p) {
		float alpha = 1.0f;
		for (int k = 0; k < K; ++k) {
			if (pix_to_face[p * K + k] < 0) continue;
			const float prob = 1.0f / (1.0f + expf(dists[p * K + k] / sigma)); /* sigmoid(-d/sigma) */
			alpha *= (1.0f - prob);
		}
		mask[p] = 1.0f - alpha;
	}
}

/* ---------------------------------------------------------------- vertex normals (Meshes.verts_normals_packed) */
void ref_vertex_normals(const float* verts, const int32_t* faces, int faces_batch, int n_meshes, int n_verts, int n_faces, float* normals) {
#pragma omp parallel for
	for (int n = 0; n < n_meshes; ++n) {
		const float* vp = verts + (size_t)n * n_verts * 3;
		const int32_t* fp = faces + (faces_batch == 1 ? 0 : (size_t)n * n_faces * 3);
		float* np_ = normals + (size_t)n * n_verts * 3;
		memset(np_, 0, sizeof(float) * (size_t)n_verts * 3);
		for (int f = 0; f < n_faces; ++f) {
			if (fp[f * 3] < 0) continue;
			const float* a = vp + 3 * fp[f * 3]; const float* b = vp + 3 * fp[f * 3 + 1]; const float* c = vp + 3 * fp[f * 3 + 2];
			/* cross(v2 - v1, v0 - v1): magnitude 2*area => area weighting */
			const float ux = c[0] - b[0], uy = c[1] - b[1], uz = c[2] - b[2];
			const float wx = a[0] - b[0], wy = a[1] - b[1], wz = a[2] - b[2];
			const float nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
			for (int k = 0; k < 3; ++k) { float* o = np_ + 3 * fp[f * 3 + k]; o[0] += nx; o[1] += ny; o[2] += nz; }
		}
		for (int v = 0; v < n_verts; ++v) {
			float* o = np_ + 3 * v;
			const float l = fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-6f);
			o[0] /= l; o[1] /= l; o[2] /= l;
		}
	}
}

static inline void normalize3(float* v) {
	const float l = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-6f);
	v[0] /= l; v[1] /= l; v[2] /= l;
}

/* ---------------------------------------------------------------- softmax blend of K fragments (A.4)
 * The in-repo statement of the formula is FootRenderer's own softmax_blend (src/model/renderer.py:23-72, the C-channel version of
 * PyTorch3D's softmax_rgb_blend): p_k = sigmoid(-d_k / sigma) on valid slots, z_inv_k = (zfar - z_k) / (zfar - znear) on valid slots,
 * weights p_k exp((z_inv_k - max z_inv) / gamma), background weight delta = max(exp((eps - max z_inv) / gamma), eps), eps = 1e-10,
 * max z_inv clamped from below by eps.  PINNED: tests/test_oracle_pins.py runs this function on fragments the imported reference
 * function blended (tests/golden/blend.npz).  colors: K x C per pixel; out: C. */
static inline void softmax_blend_pixel(const int32_t* pf, const float* d, const float* z, const float* colors, int K, int C, float sigma, float gamma,
									   float znear, float zfar, const float* background, float* out) {
	const float eps = 1e-10f;
	float z_inv_max = 0.f;   /* z_inv * mask: empty slots contribute 0 to the maximum, as in the reference */
	for (int k = 0; k < K; ++k)
		if (pf[k] >= 0) z_inv_max = fmaxf(z_inv_max, (zfar - z[k]) / (zfar - znear));
	z_inv_max = fmaxf(z_inv_max, eps);
	const float delta = fmaxf(expf((eps - z_inv_max) / gamma), eps);
	float den = delta;
	for (int c = 0; c < C; ++c) out[c] = delta * background[c];
	for (int k = 0; k < K; ++k) {
		if (pf[k] < 0) continue;
		const float prob = 1.0f / (1.0f + expf(d[k] / sigma));
		const float z_inv = (zfar - z[k]) / (zfar - znear);
		const float wnum = prob * expf((z_inv - z_inv_max) / gamma);
		den += wnum;
		for (int c = 0; c < C; ++c) out[c] += wnum * colors[(size_t)k * C + c];
	}
	for (int c = 0; c < C; ++c) out[c] /= den;
}

/* colors (n_pix, K, C) -> out (n_pix, C): softmax_blend over whole fragment buffers (test entry point) */
void ref_softmax_blend(const int32_t* pix_to_face, const float* dists, const float* zbuf, const float* colors, int64_t n_pix, int K, int C, float sigma,
					   float gamma, float znear, float zfar, const float* background, float* out) {
#pragma omp parallel for
	for (int64_t p = 0; p < n_pix; ++p)
		softmax_blend_pixel(pix_to_face + p * K, dists + p * K, zbuf + p * K, colors + (size_t)p * K * C, K, C, sigma, gamma, znear, zfar, background,
							out + p * C);
}

/* ---------------------------------------------------------------- Phong + softmax_rgb_blend with K = 1 (A.4)
 * Fragments come from ref_rasterize(K=1, blur 0, no clip).  verts/normals/colors are WORLD-space per mesh.
 * cam_center (n_views,3) = -T @ R^T.  image (n_img,H,W,3). */
void ref_phong_blend(const render_params* rp, const int32_t* pix_to_face, const float* zbuf, const float* bary, const float* dists,
					 const float* verts, const float* normals, const float* colors, const int32_t* faces, int faces_batch,
					 const float* cam_center, int n_img, int n_views, int n_verts, int n_faces, float* image) {
	const int H = rp->image_h, W = rp->image_w;
#pragma omp parallel for
	for (int64_t p = 0; p < (int64_t)n_img * H * W; ++p) {
		const int im = (int)(p / ((int64_t)H * W));
		const int mesh = im / n_views, view = im % n_views;
		float* o = image + p * 3;
		const int32_t pf = pix_to_face[p];
		if (pf < 0) {
			/* no face: weights are zero, pixel = delta*bg/delta */
			o[0] = rp->background[0]; o[1] = rp->background[1]; o[2] = rp->background[2];
			continue;
		}
		const int f = pf - im * n_faces;
		const int32_t* fp = faces + (faces_batch == 1 ? 0 : (size_t)mesh * n_faces * 3) + (size_t)f * 3;
		const float* b = bary + p * 3;
		float pos[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, tex[3] = {0, 0, 0};
		for (int k = 0; k < 3; ++k) {
			const size_t vo = ((size_t)mesh * n_verts + fp[k]) * 3;
			for (int c = 0; c < 3; ++c) { pos[c] += b[k] * verts[vo + c]; nrm[c] += b[k] * normals[vo + c]; tex[c] += b[k] * colors[vo + c]; }
		}
		/* lighting.py: diffuse / specular of a point light */
		float n[3] = {nrm[0], nrm[1], nrm[2]};
		normalize3(n);
		float l[3] = {rp->light_pos[0] - pos[0], rp->light_pos[1] - pos[1], rp->light_pos[2] - pos[2]};
		normalize3(l);
		const float cosang = n[0] * l[0] + n[1] * l[1] + n[2] * l[2];
		const float diff = rp->diffuse * fmaxf(cosang, 0.f);
		float vdir[3] = {cam_center[view * 3] - pos[0], cam_center[view * 3 + 1] - pos[1], cam_center[view * 3 + 2] - pos[2]};
		normalize3(vdir);
		const float r[3] = {-l[0] + 2.f * cosang * n[0], -l[1] + 2.f * cosang * n[1], -l[2] + 2.f * cosang * n[2]};
		float al = fmaxf(vdir[0] * r[0] + vdir[1] * r[1] + vdir[2] * r[2], 0.f) * (cosang > 0.f ? 1.f : 0.f);
		const float spec = rp->specular * powf(al, rp->shininess);
		/* softmax blend of the one fragment (K = 1, renderer.py:119-128 raster_settings) */
		float col[3];
		for (int c = 0; c < 3; ++c) col[c] = (rp->ambient + diff) * tex[c] + spec;
		softmax_blend_pixel(&pf, dists + p, zbuf + p, col, 1, 3, rp->rgb_sigma, rp->rgb_gamma, rp->znear, rp->zfar, rp->background, o);
	}
}
