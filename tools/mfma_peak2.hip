// Variants of feeding MFMA fragments from LDS: V0 burst of 5 ds_read_b128 / 16 MFMA; V1 same reads spread 1 per 3 MFMAs
// (sched_group_barrier); V2 10 x ds_read_b64; V3 20 x ds_read_b32; V4 3 reads / 16 MFMA (what a 64x128 wave tile needs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, unsigned long long* cyc) {
	__shared__ float4 sm[4096];
	for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = make_float4(i * 1e-3f, 1.f, 2.f, 3.f);
	__syncthreads();
	f32x16 acc[4];
	for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
	float4 a = sm[threadIdx.x], b[4] = {sm[threadIdx.x + 256], sm[threadIdx.x + 512], sm[threadIdx.x + 768], sm[threadIdx.x + 1024]};
	const float* sf = reinterpret_cast<const float*>(sm);
	const float2* s2 = reinterpret_cast<const float2*>(sm);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
		float4 na = a, nb[4] = {b[0], b[1], b[2], b[3]};
		const int o = (threadIdx.x + it * 64) & 1023;
		if (V == 0 || V == 1) { na = sm[o]; nb[0] = sm[o + 256]; nb[1] = sm[o + 512]; nb[2] = sm[o + 768]; nb[3] = sm[o + 1024]; }
		if (V == 2) {
			float2 t[10];
			for (int q = 0; q < 10; ++q) t[q] = s2[o + q * 512];
			na = make_float4(t[0].x, t[0].y, t[1].x, t[1].y);
			for (int q = 0; q < 4; ++q) nb[q] = make_float4(t[2 + 2 * q].x, t[2 + 2 * q].y, t[3 + 2 * q].x, t[3 + 2 * q].y);
		}
		if (V == 3) {
			float t[20];
			for (int q = 0; q < 20; ++q) t[q] = sf[o + q * 512];
			na = make_float4(t[0], t[1], t[2], t[3]);
			for (int q = 0; q < 4; ++q) nb[q] = make_float4(t[4 + 4 * q], t[5 + 4 * q], t[6 + 4 * q], t[7 + 4 * q]);
		}
		if (V == 4) { na = sm[o]; nb[0] = sm[o + 256]; nb[1] = sm[o + 512]; }
#pragma unroll
		for (int kk = 0; kk < 4; ++kk)
#pragma unroll
			for (int n = 0; n < 4; ++n) {
				const float av = kk == 0 ? a.x : kk == 1 ? a.y : kk == 2 ? a.z : a.w;
				const float bv = kk == 0 ? b[n].x : kk == 1 ? b[n].y : kk == 2 ? b[n].z : b[n].w;
				acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
			}
		if (V == 1) {
			for (int q = 0; q < 5; ++q) {
				__builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  // 3 MFMA
				__builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
			}
			__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
		}
		a = na; b[0] = nb[0]; b[1] = nb[1]; b[2] = nb[2]; b[3] = nb[3];
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0.f;
	for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name) {
	float* out; unsigned long long* cyc;
	hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
	const int iters = 20000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<V><<<256, 256>>>(out, 1000, cyc);
	hipEventRecord(e0);
	k<V><<<256, 256>>>(out, iters, cyc);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
	const double mf = (double)iters * 16;
	printf("%-40s %.1f TF/s  %.1f cycles/MFMA  clock %.2f GHz\n", name, 256.0 * 4 * mf * 4096 * 2 / ms / 1e9, c / mf, c / (ms * 1e6));
}

int main() {
	run<0>("V0 5 x b128 burst"); run<1>("V1 5 x b128 spread (sched_group_barrier)"); run<2>("V2 10 x b64"); run<3>("V3 20 x b32"); run<4>("V4 3 x b128");
	return 0;
}
