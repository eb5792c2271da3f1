// Fused multi-tensor optimiser steps for FIND's three optimisers (reference src/train/train.py:161-168: Adam over the
// network parameters and over the latent tables, SGD with momentum 0.9 over the registration parameters; stepped once per
// batch, src/train/trainer.py:121-123).  One launch updates up to 48 tensors: their device pointers travel in the kernel
// arguments (no device-side tables, nothing allocated), a workgroup finds its tensor by scanning the block prefix.
// Arithmetic follows torch.optim's single-tensor reference implementations operation for operation (dense updates: a row
// of a latent table with a zero gradient still moves with its moments, exactly as torch's dense Adam).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "find_hip.h"
#include "common.h"

namespace find {
namespace optim {

constexpr int MAXT = 48;          // tensors per launch
constexpr int CHUNK = 256 * 8;    // elements per workgroup

struct Pack {
	float* p[MAXT];
	const float* g[MAXT];
	float* a[MAXT];       // exp_avg / momentum buffer
	float* b[MAXT];       // exp_avg_sq (Adam)
	int64_t numel[MAXT];
	int blk_end[MAXT];    // exclusive prefix of workgroups
	int n;
};

__device__ __forceinline__ int find_tensor(const Pack& k, int blk, int* first_blk) {
	int t = 0, lo = 0;
	while (t < k.n - 1 && blk >= k.blk_end[t]) { lo = k.blk_end[t]; ++t; }
	*first_blk = lo;
	return t;
}

// step_dev == NULL: step_size = lr / bias_correction1 and bias_c2_sqrt come from the host (torch's default path, Python-float
// arithmetic).  step_dev != NULL ("capturable", for HIP-graph replay): the fp32 step count lives on the device, `lr` carries the
// learning rate and both corrections are formed here in fp32, as torch.optim.Adam(capturable=True) does with its device step tensor.
__global__ __launch_bounds__(256) void adam_kernel(const Pack k, float step_size, float beta1, float beta2, float eps, float weight_decay,
												   float lr, float bias_c2_sqrt, const float* __restrict__ step_dev) {
	if (step_dev) {
		const float st = *step_dev;
		step_size = lr / (1.0f - powf(beta1, st));
		bias_c2_sqrt = sqrtf(1.0f - powf(beta2, st));
	}
	int first;
	const int t = find_tensor(k, blockIdx.x, &first);
	const int64_t base = (int64_t)(blockIdx.x - first) * CHUNK;
	float* p = k.p[t];
	const float* g = k.g[t];
	float* m = k.a[t];
	float* v = k.b[t];
	const int64_t n = k.numel[t];
#pragma unroll
	for (int u = 0; u < CHUNK / 256; ++u) {
		const int64_t i = base + u * 256 + threadIdx.x;
		if (i >= n) break;
		float gi = g[i];
		const float pi = p[i];
		if (weight_decay != 0.f) gi = gi + weight_decay * pi;              // grad.add(param, alpha=weight_decay)
		const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);              // exp_avg.lerp_(grad, 1 - beta1)
		const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
		const float denom = sqrtf(vi) / bias_c2_sqrt + eps;                // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
		m[i] = mi;
		v[i] = vi;
		p[i] = pi - step_size * (mi / denom);                              // param.addcdiv_(exp_avg, denom, value=-step_size)
	}
}

__global__ __launch_bounds__(256) void sgd_kernel(const Pack k, float lr, float momentum, float dampening, float weight_decay, int nesterov,
												  int first_step) {
	int first;
	const int t = find_tensor(k, blockIdx.x, &first);
	const int64_t base = (int64_t)(blockIdx.x - first) * CHUNK;
	float* p = k.p[t];
	const float* g = k.g[t];
	float* buf = k.a[t];
	const int64_t n = k.numel[t];
#pragma unroll
	for (int u = 0; u < CHUNK / 256; ++u) {
		const int64_t i = base + u * 256 + threadIdx.x;
		if (i >= n) break;
		float gi = g[i];
		const float pi = p[i];
		if (weight_decay != 0.f) gi = gi + weight_decay * pi;
		if (momentum != 0.f) {
			const float bi = first_step ? gi : momentum * buf[i] + (1.0f - dampening) * gi;  // clone on the first step, then mul_.add_
			buf[i] = bi;
			gi = nesterov ? gi + momentum * bi : bi;
		}
		p[i] = pi - lr * gi;
	}
}

static int pack_and_launch(bool adam, int64_t n, float* const* param, const float* const* grad, float* const* a, float* const* b,
						   const int64_t* numel, hipStream_t s, float f0, float f1, float f2, float f3, float f4, float f5, float f6, int i0, int i1,
						   const float* step_dev = nullptr) {
	for (int64_t t0 = 0; t0 < n; t0 += MAXT) {
		Pack k;
		memset(&k, 0, sizeof(k));
		k.n = (int)std::min<int64_t>(MAXT, n - t0);
		int blocks = 0;
		for (int t = 0; t < k.n; ++t) {
			const int64_t ne = numel[t0 + t];
			FIND_REQUIRE(param[t0 + t] && grad[t0 + t] && ne >= 0 && ne < (1ll << 40), "find optimiser step: bad tensor %lld", (long long)(t0 + t));
			k.p[t] = param[t0 + t]; k.g[t] = grad[t0 + t]; k.a[t] = a ? a[t0 + t] : nullptr; k.b[t] = b ? b[t0 + t] : nullptr;
			FIND_REQUIRE(!adam || (k.a[t] && k.b[t]), "find_adam_step: NULL moment buffer");
			k.numel[t] = ne;
			blocks += (int)((ne + CHUNK - 1) / CHUNK);
			k.blk_end[t] = blocks;
		}
		if (blocks == 0) continue;
		if (adam) hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, k, f0, f1, f2, f3, f4, f5, f6, step_dev);
		else hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, k, f0, f1, f2, f3, i0, i1);
		FIND_LAUNCH_CHECK(adam ? "adam_kernel" : "sgd_kernel");
	}
	return FIND_OK;
}

}  // namespace optim
}  // namespace find

using namespace find;

extern "C" int find_adam_step(int64_t n_tensors, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
							  const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
	FIND_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (param && grad && exp_avg && exp_avg_sq && numel)), "find_adam_step: NULL argument");
	FIND_REQUIRE(step >= 1, "find_adam_step: step counts from 1 (it is the value AFTER this update, as torch.optim.Adam's state['step'])");
	FIND_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "find_adam_step: bad hyper-parameters");
	// bias corrections in double, then rounded once: what torch computes on the host in Python floats
	const double bc1 = 1.0 - pow((double)beta1, (double)step);
	const double bc2 = 1.0 - pow((double)beta2, (double)step);
	return optim::pack_and_launch(true, n_tensors, param, grad, exp_avg, exp_avg_sq, numel, reinterpret_cast<hipStream_t>(stream), (float)((double)lr / bc1), beta1,
								  beta2, eps, weight_decay, 0.f, (float)sqrt(bc2), 0, 0);
}

extern "C" int find_adam_step_dev(int64_t n_tensors, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
								  const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay, const float* step_dev, void* stream) {
	FIND_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (param && grad && exp_avg && exp_avg_sq && numel)), "find_adam_step_dev: NULL argument");
	FIND_REQUIRE(step_dev != nullptr, "find_adam_step_dev: step_dev is NULL");
	FIND_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "find_adam_step_dev: bad hyper-parameters");
	return optim::pack_and_launch(true, n_tensors, param, grad, exp_avg, exp_avg_sq, numel, reinterpret_cast<hipStream_t>(stream), 0.f, beta1,
								  beta2, eps, weight_decay, lr, 0.f, 0, 0, step_dev);
}

extern "C" int find_sgd_step(int64_t n_tensors, float* const* param, const float* const* grad, float* const* momentum_buf, const int64_t* numel,
							 float lr, float momentum, float dampening, float weight_decay, int nesterov, int first_step, void* stream) {
	FIND_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (param && grad && numel)), "find_sgd_step: NULL argument");
	FIND_REQUIRE(momentum == 0.f || momentum_buf, "find_sgd_step: momentum needs its buffers");
	FIND_REQUIRE(!nesterov || (momentum > 0.f && dampening == 0.f), "find_sgd_step: Nesterov momentum requires a momentum and zero dampening");
	return optim::pack_and_launch(false, n_tensors, param, grad, momentum_buf, nullptr, numel, reinterpret_cast<hipStream_t>(stream), lr, momentum, dampening,
								  weight_decay, 0.f, 0.f, 0.f, nesterov, first_step);
}
