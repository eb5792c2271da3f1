// Probe 3: fused6's activation planes (row-major, 32 rows of a block x 256 bf16, padded rows): cycles per ds_read_b128 of the fragment reads
// (lane (i, h): row i, 16 bytes at half h) and per ds_write_b64 of the epilogue (lane (i, h): row i, 8 bytes at 8 h) for several row strides.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k(unsigned* out, int iters, int S, int write, unsigned long long* cyc) {
	extern __shared__ __attribute__((aligned(16))) char sm[];
	for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(sm)[i] = i;
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned base = (unsigned)(uintptr_t)sm + (lane & 31) * S + (write ? (lane >> 5) * 8 + wave * 64 : (lane >> 5) * 16);
	u4 acc = {1, 2, 3, 4};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
		if (write) {
#pragma unroll
			for (int j = 0; j < 16; ++j) asm volatile("ds_write_b64 %0, %1" :: "v"(base + (j & 3) * 16), "v"(u2{acc.x, acc.y}) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		} else {
			u4 v[16];
#pragma unroll
			for (int j = 0; j < 16; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(base + (j & 7) * 32));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int j = 0; j < 16; ++j) acc ^= v[j];
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 512 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
	unsigned* out; unsigned long long* cyc;
	hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
	const int iters = 4000;
	for (int S : {528, 544, 560, 576, 592, 608, 640, 1040})
		for (int write = 0; write < 2; ++write) {
			k<<<256, 512, 64 * 1024>>>(out, 100, S, write, cyc);
			k<<<256, 512, 64 * 1024>>>(out, iters, S, write, cyc);
			hipDeviceSynchronize();
			unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
			double t = 0; for (int i = 0; i < 256; ++i) t += h[i]; t /= 256;
			printf("row stride %4d %s: %.2f ticks per instruction (CU-wide, 8 waves issuing)\n", S, write ? "ds_write_b64 (row i, 8 bytes at 8 h)" : "ds_read_b128 (row i, half h)   ", t / (iters * 16.0 * 8));
		}
	return 0;
}
