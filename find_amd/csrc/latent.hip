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


// ---- several tables in one launch (a training step looks up four: shape, pose, texture code and registration rows)
constexpr int MANY_MAX = 8;
struct ManyArgs {
	const float* table[MANY_MAX];   // fwd: tables.  bwd: d_out of every lookup (NULL: no gradient arrived, the table's gradient is zero)
	const int64_t* idx[MANY_MAX];
	float* out[MANY_MAX];           // fwd: gathered rows.  bwd: d_table
	int64_t n_rows[MANY_MAX];
	int dim[MANY_MAX];
	int64_t n_idx;
};

__global__ void gather_many_fwd_kernel(const ManyArgs a) {
	const int t = blockIdx.y;
	const int dim = a.dim[t];
	const int64_t n_rows = a.n_rows[t];
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.n_idx * dim) return;
	const int64_t i = e / dim;
	const int j = (int)(e - i * dim);
	int64_t r = a.idx[t][i];
	if (r < 0) r += n_rows;
	a.out[t][e] = (r < 0 || r >= n_rows) ? __builtin_nanf("") : a.table[t][r * dim + j];
}

__global__ __launch_bounds__(256) void gather_many_bwd_kernel(const ManyArgs a) {
	__shared__ int64_t sidx[IDX_TILE];
	const int t = blockIdx.y;
	const int dim = a.dim[t];
	const int64_t n_rows = a.n_rows[t];
	const float* d_out = a.table[t];
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if ((int64_t)blockIdx.x * blockDim.x >= n_rows * dim) return;   // (whole block beyond this table: uniform)
	const bool live = e < n_rows * dim;
	const int64_t r = live ? e / dim : -1;
	const int j = live ? (int)(e - r * dim) : 0;
	float s = 0.f;
	if (d_out)
		for (int64_t i0 = 0; i0 < a.n_idx; i0 += IDX_TILE) {
			const int nt = (int)min((int64_t)IDX_TILE, a.n_idx - i0);
			__syncthreads();
			if ((int)threadIdx.x < nt) {
				int64_t v = a.idx[t][i0 + threadIdx.x];
				sidx[threadIdx.x] = v < 0 ? v + n_rows : v;
			}
			__syncthreads();
			if (live)
				for (int k = 0; k < nt; ++k)
					if (sidx[k] == r) s += d_out[(i0 + k) * dim + j];
		}
	if (live) a.out[t][e] = s;
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

static int fill_many(const char* who, int64_t n_tables, const float* const* src, const int64_t* n_rows, const int64_t* dims, const int64_t* const* idx,
					 int64_t n_idx, float* const* dst, bool src_optional, latent::ManyArgs* a, int64_t* max_elems, bool bwd) {
	FIND_REQUIRE(n_tables >= 1 && n_tables <= latent::MANY_MAX, "%s: 1 .. %d tables (got %lld)", who, latent::MANY_MAX, (long long)n_tables);
	FIND_REQUIRE(src && n_rows && dims && idx && dst && n_idx >= 0, "%s: NULL argument", who);
	memset(a, 0, sizeof(*a));
	*max_elems = 0;
	for (int64_t t = 0; t < n_tables; ++t) {
		FIND_REQUIRE((src[t] || src_optional) && idx[t] && dst[t], "%s: NULL pointer for table %lld", who, (long long)t);
		FIND_REQUIRE(n_rows[t] >= 1 && dims[t] >= 1 && dims[t] < (1 << 24), "%s: bad sizes for table %lld", who, (long long)t);
		a->table[t] = src[t]; a->idx[t] = idx[t]; a->out[t] = dst[t]; a->n_rows[t] = n_rows[t]; a->dim[t] = (int)dims[t];
		*max_elems = std::max<int64_t>(*max_elems, (bwd ? n_rows[t] : n_idx) * dims[t]);
	}
	a->n_idx = n_idx;
	return FIND_OK;
}

extern "C" int find_latent_gather_many_fwd(int64_t n_tables, const float* const* tables, const int64_t* n_rows, const int64_t* dims,
										   const int64_t* const* idx, int64_t n_idx, float* const* outs, void* stream) {
	latent::ManyArgs a;
	int64_t m;
	int rc = fill_many("find_latent_gather_many_fwd", n_tables, tables, n_rows, dims, idx, n_idx, outs, false, &a, &m, false);
	if (rc != FIND_OK) return rc;
	if (n_idx == 0) return FIND_OK;
	hipLaunchKernelGGL(latent::gather_many_fwd_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)n_tables), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
	FIND_LAUNCH_CHECK("gather_many_fwd_kernel");
	return FIND_OK;
}

extern "C" int find_latent_gather_many_bwd(int64_t n_tables, const float* const* d_outs, const int64_t* n_rows, const int64_t* dims,
										   const int64_t* const* idx, int64_t n_idx, float* const* d_tables, void* stream) {
	latent::ManyArgs a;
	int64_t m;
	int rc = fill_many("find_latent_gather_many_bwd", n_tables, d_outs, n_rows, dims, idx, n_idx, d_tables, true, &a, &m, true);
	if (rc != FIND_OK) return rc;
	hipLaunchKernelGGL(latent::gather_many_bwd_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)n_tables), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
	FIND_LAUNCH_CHECK("gather_many_bwd_kernel");
	return FIND_OK;
}

// ------------------------------------------------------------------------------------------------ weighted loss terms
// losses[k] = raw_k * opts.weight_k, loss = sum_k losses[k]   (reference src/model/model.py:1157-1163): one launch for all terms.
namespace find {
namespace latent {
struct TermArgs {
	const float* term[MANY_MAX];
	float w[MANY_MAX];
	int n;
};
__global__ void weighted_terms_fwd_kernel(const TermArgs a, float* __restrict__ scaled, float* __restrict__ total) {
	if (threadIdx.x != 0) return;
	float s = 0.f;
	for (int i = 0; i < a.n; ++i) {
		const float v = *a.term[i] * a.w[i];
		scaled[i] = v;
		s += v;   // in term order, as Python's sum() over the dict
	}
	*total = s;
}
__global__ void weighted_terms_bwd_kernel(const TermArgs a, const float* __restrict__ g_total, const float* __restrict__ g_scaled, float* __restrict__ d_terms) {
	const int i = threadIdx.x;
	if (i >= a.n) return;
	d_terms[i] = a.w[i] * ((g_total ? *g_total : 0.f) + (g_scaled ? g_scaled[i] : 0.f));
}
}  // namespace latent
}  // namespace find

extern "C" int find_weighted_terms_fwd(int64_t n, const float* const* terms, const float* weights, float* scaled, float* total, void* stream) {
	FIND_REQUIRE(n >= 1 && n <= latent::MANY_MAX && terms && weights && scaled && total, "find_weighted_terms_fwd: 1 .. %d terms, no NULL argument", latent::MANY_MAX);
	latent::TermArgs a;
	memset(&a, 0, sizeof(a));
	for (int64_t i = 0; i < n; ++i) {
		FIND_REQUIRE(terms[i], "find_weighted_terms_fwd: term %lld is NULL", (long long)i);
		a.term[i] = terms[i]; a.w[i] = weights[i];
	}
	a.n = (int)n;
	hipLaunchKernelGGL(latent::weighted_terms_fwd_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a, scaled, total);
	FIND_LAUNCH_CHECK("weighted_terms_fwd_kernel");
	return FIND_OK;
}

extern "C" int find_weighted_terms_bwd(int64_t n, const float* weights, const float* g_total, const float* g_scaled, float* d_terms, void* stream) {
	FIND_REQUIRE(n >= 1 && n <= latent::MANY_MAX && weights && d_terms, "find_weighted_terms_bwd: 1 .. %d terms, no NULL argument", latent::MANY_MAX);
	FIND_REQUIRE(g_total || g_scaled, "find_weighted_terms_bwd: both upstream gradients NULL");
	latent::TermArgs a;
	memset(&a, 0, sizeof(a));
	for (int64_t i = 0; i < n; ++i) a.w[i] = weights[i];
	a.n = (int)n;
	hipLaunchKernelGGL(latent::weighted_terms_bwd_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a, g_total, g_scaled, d_terms);
	FIND_LAUNCH_CHECK("weighted_terms_bwd_kernel");
	return FIND_OK;
}
