// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 from one wave per SIMD (4 waves/CU), 4 independent accumulators,
// (a) registers only, (b) with 5 ds_read_b128 per 16 MFMAs like the GEMM's fragment traffic.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, unsigned long long* cyc) {
	__shared__ float4 sm[4096];
	for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = make_float4(i * 1e-3f, 1.f, 2.f, 3.f);
	__syncthreads();
	f32x16 acc[4];
	for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
	float4 a = sm[threadIdx.x], b[4] = {sm[threadIdx.x + 256], sm[threadIdx.x + 512], sm[threadIdx.x + 768], sm[threadIdx.x + 1024]};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
		float4 na = a, nb[4] = {b[0], b[1], b[2], b[3]};
		if (LDS) {
			const int o = (threadIdx.x + it * 64) & 2047;
			na = sm[o]; nb[0] = sm[o + 256]; nb[1] = sm[o + 512]; nb[2] = sm[o + 768]; nb[3] = sm[o + 1024];
		}
#pragma unroll
		for (int kk = 0; kk < 4; ++kk)
#pragma unroll
			for (int n = 0; n < 4; ++n) {
				const float av = kk == 0 ? a.x : kk == 1 ? a.y : kk == 2 ? a.z : a.w;
				const float bv = kk == 0 ? b[n].x : kk == 1 ? b[n].y : kk == 2 ? b[n].z : b[n].w;
				acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
			}
		a = na; b[0] = nb[0]; b[1] = nb[1]; b[2] = nb[2]; b[3] = nb[3];
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0.f;
	for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int LDS>
void run(const char* name) {
	float* out; unsigned long long* cyc;
	hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
	const int iters = 20000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<LDS><<<256, 256>>>(out, 1000, cyc);
	hipEventRecord(e0);
	k<LDS><<<256, 256>>>(out, iters, cyc);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
	const double mf = (double)iters * 16;
	const double flops = 256.0 * 4 * mf * 32 * 32 * 2 * 2;
	printf("%s: %.3f ms  %.1f TF/s  %.1f cycles/MFMA  clock %.2f GHz\n", name, ms, flops / ms / 1e9, c / mf, c / (ms * 1e6));
}

int main() { run<0>("regs only"); run<1>("with ds_read_b128 x5 / 16 MFMA"); return 0; }
