// Probe: cycles per ds_read_b128 for lane -> address mappings of the form  (lane % A) * S + (lane / A) * H  (bytes): which fragment layouts of
// the bf16x3 kernels are bank-conflict free on gfx950?  (SQ_LDS_BANK_CONFLICT says the 16-row x 4-quarter layout of gemm7's first 16x16x32
// version conflicts although every run of 16 consecutive lanes covers all 64 banks.)  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int A, int S, int H, unsigned long long* cyc) {
	extern __shared__ __attribute__((aligned(16))) char sm[];
	for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = i;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int base = (lane % A) * S + (lane / A) * H;
	u4 acc = {0, 0, 0, 0};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	const unsigned lbase = (unsigned)(uintptr_t)sm + base;
	for (int it = 0; it < iters; ++it) {
		u4 v[16];
#pragma unroll
		for (int j = 0; j < 16; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(lbase + (j & 7) * 64));
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int j = 0; j < 16; ++j) acc ^= v[j];
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
	unsigned* out; unsigned long long* cyc;
	hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	struct { int A, S, H; const char* what; } cfg[] = {
		{32, 528, 16, "32 rows x 2 halves, stride 528 (gemm5/gemm6 layout)"}, {16, 528, 16, "16 rows x 4 quarters, stride 528 (gemm7)"},
		{16, 528, 32, "16 rows x 4 quarters 32 B apart"}, {16, 528, 64, "16 rows x 4 quarters 64 B apart"}, {16, 528, 128, "16 x 4, 128 B apart"},
		{16, 544, 16, "16 x 4, stride 544"}, {16, 560, 16, "16 x 4, stride 560"}, {16, 576, 16, "16 x 4, stride 576"}, {16, 592, 16, "16 x 4, stride 592"},
		{16, 640, 16, "16 x 4, stride 640"}, {16, 1040, 16, "16 x 4, stride 1040"}, {16, 512, 16, "16 x 4, stride 512 (no padding)"}, {64, 16, 0, "64 consecutive 16-B words"},
		{16, 528 + 64, 16, "16 x 4, stride 592"}, {8, 528, 16, "8 rows x 8 eighths"}, {16, 272, 16, "16 x 4, stride 272"}, {16, 4112, 16, "16 x 4, stride 4112"}};
	const int iters = 4000;
	for (auto& c : cfg) {
		k<<<256, 256, 160 * 1024>>>(out, 100, c.A, c.S, c.H, cyc);
		k<<<256, 256, 160 * 1024>>>(out, iters, c.A, c.S, c.H, cyc);
		hipDeviceSynchronize();
		unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
		double t = 0; for (int i = 0; i < 256; ++i) t += h[i]; t /= 256;
		// 4 waves per workgroup share the CU's LDS: ticks per ds_read_b128 per CU = t / (iters * 16 * 4)
		printf("%-56s A=%2d S=%4d H=%3d: %.2f ticks per ds_read_b128 (CU-wide, 4 waves issuing)\n", c.what, c.A, c.S, c.H, t / (iters * 16.0 * 4));
	}
	return 0;
}
