// Latent-table row lookup and its gradient (reference src/model/model.py:131-152, LatentVector.__getitem__ with a tensor
// of indices; autograd there scatters the row gradients back with index_put).
//   fwd: out[i, :] = table[idx[i], :]
//   bwd: d_table[r, :] = sum_{i : idx[i] == r} d_out[i, :]   -- every table element owned by one thread, summed in index
//        order: deterministic, duplicates included, rows that were not looked up come out zero (no separate fill).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "find_hip.h"
#include "common.h"

namespace find {
namespace latent {

__global__ void gather_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, int64_t n_rows, int dim, int64_t n_idx,
								  float* __restrict__ out) {
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_idx * dim) return;
	const int64_t i = e / dim;
	const int j = (int)(e - i * dim);
	int64_t r = idx[i];
	if (r < 0) r += n_rows;  // python-style negative index
	if (r < 0 || r >= n_rows) { out[e] = __builtin_nanf(""); return; }  // loud downstream, no host sync here
	out[e] = table[r * dim + j];
}

constexpr int IDX_TILE = 256;
__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ d_out, const int64_t* __restrict__ idx, int64_t n_rows, int dim,
														  int64_t n_idx, float* __restrict__ d_table) {
	__shared__ int64_t sidx[IDX_TILE];
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = e < n_rows * dim;
	const int64_t r = live ? e / dim : -1;
	const int j = live ? (int)(e - r * dim) : 0;
	float s = 0.f;
	for (int64_t i0 = 0; i0 < n_idx; i0 += IDX_TILE) {
		const int nt = (int)min((int64_t)IDX_TILE, n_idx - i0);
		__syncthreads();
		if ((int)threadIdx.x < nt) {
			int64_t v = idx[i0 + threadIdx.x];
			sidx[threadIdx.x] = v < 0 ? v + n_rows : v;
		}
		__syncthreads();
		if (live)
			for (int t = 0; t < nt; ++t)
				if (sidx[t] == r) s += d_out[(i0 + t) * dim + j];
	}
	if (live) d_table[e] = s;
}

}  // namespace latent
}  // namespace find

using namespace find;

extern "C" int find_latent_gather_fwd(const float* table, int64_t n_rows, int64_t dim, const int64_t* idx, int64_t n_idx, float* out, void* stream) {
	FIND_REQUIRE(table && idx && out, "find_latent_gather_fwd: NULL argument");
	FIND_REQUIRE(n_rows >= 1 && dim >= 1 && dim < (1 << 24) && n_idx >= 0, "find_latent_gather_fwd: bad sizes");
	if (n_idx == 0) return FIND_OK;
	hipStream_t s = reinterpret_cast<hipStream_t>(stream);
	const int64_t total = n_idx * dim;
	hipLaunchKernelGGL(latent::gather_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, table, idx, n_rows, (int)dim, n_idx, out);
	FIND_LAUNCH_CHECK("gather_fwd_kernel");
	return FIND_OK;
}

extern "C" int find_latent_gather_bwd(const float* d_out, const int64_t* idx, int64_t n_idx, int64_t n_rows, int64_t dim, float* d_table,
									  void* stream) {
	FIND_REQUIRE(d_out && idx && d_table, "find_latent_gather_bwd: NULL argument");
	FIND_REQUIRE(n_rows >= 1 && dim >= 1 && dim < (1 << 24) && n_idx >= 0, "find_latent_gather_bwd: bad sizes");
	hipStream_t s = reinterpret_cast<hipStream_t>(stream);
	const int64_t total = n_rows * dim;
	hipLaunchKernelGGL(latent::gather_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_out, idx, n_rows, (int)dim, n_idx, d_table);
	FIND_LAUNCH_CHECK("gather_bwd_kernel");
	return FIND_OK;
}
