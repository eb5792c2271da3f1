// Microbenchmark: v_mfma_f32_32x32x16_bf16 interleaved with independent VALU work -- how much of the splitting arithmetic of the bf16x3
// kernels (mlp_gemm6.h) does the matrix pipe hide?  W waves per SIMD (1 or 2), NACC independent accumulators, V VALU instructions
// (v_pk_add_f32 / v_cvt_pk_bf16_f32 / v_and_b32 mix, as in split_pair) between consecutive MFMAs.  Prints cycles per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_mix.hip -o tools/mfma_bf16_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V, int NACC, int AG = 0>   // AG: the MFMA's row operand lives in accumulation registers (as the stationary weights of gemm7 do)
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
	f32x16 acc[NACC];
	for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
	bf16x8 a, b;
	for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.f + j); }
	if (iters < 0) {   // "random" operands (different bit patterns in every lane and element; rotated every MFMA): what real data costs in power / clocks
		iters = -iters;
		unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
		unsigned wa[4], wb[4];
		for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; wa[j] = (h & 0x7fff7fffu) | 0x30003000u; h = h * 1664525u + 1013904223u; wb[j] = (h & 0x7fff7fffu) | 0x30003000u; }
		typedef unsigned u4 __attribute__((ext_vector_type(4)));
		a = __builtin_bit_cast(bf16x8, u4{wa[0], wa[1], wa[2], wa[3]}); b = __builtin_bit_cast(bf16x8, u4{wb[0], wb[1], wb[2], wb[3]});
	}
	if (AG) asm volatile("" : "+a"(a));
	f32x2 x = {threadIdx.x * 1.0f, 2.f}, y = {1.5f, 0.25f};
	unsigned u = threadIdx.x;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
			for (int v = 0; v < V; ++v) {
				if (v % 3 == 0) x = x - y;                                                       // v_pk_add_f32
				else if (v % 3 == 1) u = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2)) ^ u;   // v_cvt_pk_bf16_f32 (+ xor)
				else y = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};    // shl + and
			}
			__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
			__builtin_amdgcn_sched_group_barrier(0x002, V + V / 3, 0);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = x.x + y.y + __uint_as_float(u);
	for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
	out[blockIdx.x * 512 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int NACC, int AG = 0>
void run(int threads, int sign = 1) {
	float* out; unsigned long long* cyc;
	hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
	const int iters = 20000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<V, NACC, AG><<<256, threads>>>(out, 1000, cyc);
	hipEventRecord(e0);
	k<V, NACC, AG><<<256, threads>>>(out, sign * iters, cyc);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
	const double waves_per_simd = threads / 256.0;
	const double mf = (double)iters * 8 * waves_per_simd;   // MFMAs per SIMD
	printf("%s%sV=%d VALU per MFMA, %d accumulators, %.0f wave(s)/SIMD: %.3f ms, %.1f memtime ticks per MFMA per SIMD, %.2f us per 1000 MFMA per SIMD (32 cycles @2.4GHz = 13.3 us)\n",
		   AG ? "[row operand in AGPRs] " : "", sign < 0 ? "[random operands] " : "", V, NACC, waves_per_simd, ms, c / mf, ms * 1e3 / (mf / 1000));
	hipFree(out); hipFree(cyc);
}

int main() {
	run<0, 4>(256); run<0, 4>(256);   // (twice: the first run also brings the clocks up)
	run<0, 1, 1>(256); run<0, 1, 1>(256, -1); run<1, 1, 1>(256, -1); run<0, 1, 0>(256, -1); run<0, 4, 1>(512, -1); return 0;
	for (int rep = 0; rep < 2; ++rep) { run<0, 4>(256); run<0, 4>(256, -1); run<0, 4>(512); run<0, 4>(512, -1); run<1, 1>(256, -1); run<3, 4>(512, -1); }
	for (int t : {256, 512}) {
		run<0, 4>(t); run<1, 4>(t); run<2, 4>(t); run<3, 4>(t); run<4, 4>(t); run<6, 4>(t); run<8, 4>(t);
		run<0, 2>(t); run<3, 2>(t); run<6, 2>(t);
		run<0, 1>(t); run<1, 1>(t); run<3, 1>(t);   // ONE accumulator: every MFMA waits for the one before it
	}
	return 0;
}
