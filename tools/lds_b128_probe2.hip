// Probe 2: cycles per ds_write_b128 / ds_read_b128 for dw6's transposed-plane layout ([column'][16 rows] bf16 = 32 B per column, two 16-byte
// halves) under column swizzles:  which one is bank-conflict free for the WRITES (lane l stores columns 4 l + e) and for the fragment READS
// (lane (i, h) reads column 32 t + i, half h)?  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_b128_probe2.hip -o /tmp/p2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ int swz(int col, int half, int mode, int* half_out) {
	int c = col, h = half;
	switch (mode) {
	case 0: c = col ^ ((col >> 2) & 7); break;                       // dw6 today
	case 1: c = col; break;                                           // none
	case 2: c = col ^ ((col >> 2) & 7); h = half ^ ((col >> 5) & 1); break;
	case 3: c = col ^ ((col >> 2) & 7); h = half ^ ((col >> 4) & 1); break;
	case 4: c = col ^ ((col >> 2) & 7); h = half ^ ((col >> 3) & 1); break;
	case 5: c = col ^ ((col >> 3) & 7); break;
	case 6: c = col ^ ((col >> 2) & 3); break;
	case 7: c = col ^ ((col >> 2) & 7) ^ (((col >> 5) & 1) << 2); break;
	case 8: c = col ^ ((col >> 2) & 7); h = half ^ ((col >> 2) & 1); break;
	case 9: c = col ^ ((col >> 1) & 7); break;
	}
	*half_out = h;
	return c;
}
// layouts 10+: [half][column'] x 16 B (the two row halves of a plane 4 KB apart)
__device__ unsigned addr2(int col, int half, int mode) {
	int c = col;
	if (mode == 10) c = col ^ ((col >> 4) & 3);
	if (mode == 11) c = col;
	if (mode == 12) c = col ^ ((col >> 2) & 3);
	if (mode == 13) c = col ^ ((col >> 4) & 3) ^ (((col >> 6) & 3) << 2);
	if (mode == 14) { c = col ^ ((col >> 2) & 7); const int h = half ^ (((col >> 4) ^ (col >> 5)) & 1); return c * 32 + h * 16; }
	if (mode == 15) { c = col ^ ((col >> 2) & 7); return c * 32 + (c >> 3) * 16 + half * 16; }            // + 16 B of padding per 8 columns
	if (mode == 16) { c = col ^ ((col >> 2) & 7); const int h = half ^ ((col >> 4) & 1); return c * 32 + (c >> 5) * 16 + h * 16; }   // mode 3 + 16 B per 32 columns
	if (mode == 17) { c = col; return c * 32 + (c >> 2) * 16 + half * 16; }                              // no swizzle, 16 B of padding per 4 columns (144-B groups)
	if (mode == 18) { c = col ^ ((col >> 2) & 7); const int h = half ^ ((col >> 4) & 1) ^ ((col >> 6) & 1); return c * 32 + h * 16; }
	if (mode == 19) { c = col ^ ((col >> 2) & 7); const int h = half ^ ((col >> 4) & 1); return c * 32 + h * 16 + ((col >> 7) & 1) * 0; }
	return half * 4096 + c * 16;
}
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int mode, int write, unsigned long long* cyc) {
	extern __shared__ __attribute__((aligned(16))) char sm[];
	for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = i;
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	unsigned addr[4];
	for (int e = 0; e < 4; ++e) {
		int h;
		if (mode >= 10) addr[e] = write ? addr2(4 * lane + e, wave & 1, mode) : addr2(32 * e + (lane & 31), lane >> 5, mode);
		else if (write) {   // lane l stores column 4 l + e, row half = wave & 1
			const int c = swz(4 * lane + e, wave & 1, mode, &h);
			addr[e] = c * 32 + h * 16;
		} else {       // lane (i, fh) reads column 32 e + i, half fh
			const int c = swz(32 * e + (lane & 31), lane >> 5, mode, &h);
			addr[e] = c * 32 + h * 16;
		}
		addr[e] += (unsigned)(uintptr_t)sm;
	}
	u4 acc = {1, 2, 3, 4};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
		if (write) {
#pragma unroll
			for (int j = 0; j < 16; ++j) asm volatile("ds_write_b128 %0, %1" :: "v"(addr[j & 3] + (j >> 2) * 8192), "v"(acc) : "memory");
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		} else {
			u4 v[16];
#pragma unroll
			for (int j = 0; j < 16; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(addr[j & 3] + (j >> 2) * 8192));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int j = 0; j < 16; ++j) acc ^= v[j];
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
	unsigned* out; unsigned long long* cyc;
	hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
	const int iters = 4000;
	for (int mode = 14; mode < 20; ++mode)
		for (int write = 0; write < 2; ++write) {
			k<<<256, 256, 64 * 1024>>>(out, 100, mode, write, cyc);
			k<<<256, 256, 64 * 1024>>>(out, iters, mode, write, cyc);
			hipDeviceSynchronize();
			unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
			double t = 0; for (int i = 0; i < 256; ++i) t += h[i]; t /= 256;
			printf("swizzle %d %s: %.2f ticks per ds_%s_b128 (CU-wide, 4 waves issuing)\n", mode, write ? "writes (lane l: columns 4 l + e)" : "reads  (lane (i, h): column 32 t + i)", t / (iters * 16.0 * 4), write ? "write" : "read");
		}
	return 0;
}
