// TEMPORARY: entry points declared in include/find_hip.h whose kernels are not written yet.
// Each returns FIND_EINVAL with an explicit message; this file shrinks as kernels land and is then deleted.
#include "common.h"
#define PENDING(name) do { find::set_error(name ": not implemented yet"); return FIND_EINVAL; } while (0)
extern "C" {
int find_sample_points_fwd(const float*, const int32_t*, int64_t, const int32_t*, const float*, int64_t, int64_t, int64_t, int64_t, float*, const float*, float*, void*) { PENDING("find_sample_points_fwd"); }
int find_sample_points_bwd(const int32_t*, int64_t, const int32_t*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, float*, void*) { PENDING("find_sample_points_bwd"); }
int find_face_areas(const float*, const int32_t*, int64_t, int64_t, int64_t, int64_t, float*, void*) { PENDING("find_face_areas"); }
int find_nn_fwd(const float*, const int32_t*, const float*, const int32_t*, int64_t, int64_t, int64_t, float*, int32_t*, void*) { PENDING("find_nn_fwd"); }
int find_nn_bwd(const float*, const int32_t*, const float*, const int32_t*, const float*, int64_t, int64_t, int64_t, float*, float*, void*) { PENDING("find_nn_bwd"); }
int64_t find_smooth_ws_bytes(int64_t, int64_t, int64_t, int64_t) { return -1; }
int find_smooth_fwd(const float*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t, float*, float*, void*, int64_t, void*) { PENDING("find_smooth_fwd"); }
int find_smooth_bwd(const float*, const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t, const float*, const float*, const void*, int64_t, float*, void*) { PENDING("find_smooth_bwd"); }
int64_t find_render_ws_bytes(const find_render_params*, int64_t, int64_t, int64_t, int64_t) { return -1; }
int find_render_fwd(const find_render_params*, const float*, const int32_t*, int64_t, const float*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, float*, float*, int32_t*, float*, void*, int64_t, void*) { PENDING("find_render_fwd"); }
int find_render_bwd(const find_render_params*, const float*, const int32_t*, int64_t, const float*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, const float*, const float*, float*, float*, void*, int64_t, void*) { PENDING("find_render_bwd"); }
}
