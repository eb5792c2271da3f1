// TEMPORARY: entry points declared in include/find_hip.h whose kernels are not written yet.
// Each returns FIND_EINVAL with an explicit message; this file shrinks as kernels land and is then deleted.
#include "common.h"
#define PENDING(name) do { find::set_error(name ": not implemented yet"); return FIND_EINVAL; } while (0)
extern "C" {
int64_t find_render_ws_bytes(const find_render_params*, int64_t, int64_t, int64_t, int64_t) { return -1; }
int find_render_fwd(const find_render_params*, const float*, const int32_t*, int64_t, const float*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, float*, float*, int32_t*, float*, void*, int64_t, void*) { PENDING("find_render_fwd"); }
int find_render_bwd(const find_render_params*, const float*, const int32_t*, int64_t, const float*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, const float*, const float*, float*, float*, void*, int64_t, void*) { PENDING("find_render_bwd"); }
}
