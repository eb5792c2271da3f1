// Does gfx950 read the data registers of a 128-bit buffer store AFTER the instruction has issued?  Every wave stores v[20:23] with
// buffer_store_dwordx4 and overwrites v20 (the first data dword) with a VALU instruction right behind it, thousands of times, while the
// memory pipeline is backed up.  The ISA's hazard rule (and LLVM's GCNHazardRecognizer::createsVALUHazard) asks for one wait state between
// a VMEM store of more than 64 bits and a VALU write of its data registers ONLY when the store has no SGPR offset.  Modes:
//   0: SGPR soffset, VALU overwrite immediately          1: immediate soffset (0), VALU overwrite immediately (the documented hazard)
//   2: SGPR soffset, s_nop 0 between (1 wait state)       3: SGPR soffset, s_nop 4 between        4: SGPR soffset, no overwrite (control)
//   5: SGPR soffset, overwrite after 2 independent VALU instructions
//   6 / 7: the same as 0 / 1 with a 64-bit store (buffer_store_dwordx2; the manual's rule is about stores of MORE than 64 bits)   8: buffer_store_dword, SGPR soffset
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/store_hazard_probe tools/store_hazard_probe.hip ; run on the GPU box: tools/store_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	float* base = out + (size_t)wave * iters * 256;   // 1 KB per store
	typedef int i4 __attribute__((ext_vector_type(4)));
	const unsigned long long pb = (unsigned long long)base;
	const i4 rs = i4{__builtin_amdgcn_readfirstlane((int)(unsigned)pb), __builtin_amdgcn_readfirstlane((int)((pb >> 32) & 0xffff)), iters * 1024, 0x00020000};
	const float x = 1000.f + lane, y = 2000.f + lane, z = 3000.f + lane, w = 4000.f + lane;
	for (int i = 0; i < iters; ++i) {
		const int soff = __builtin_amdgcn_readfirstlane(i * 1024);
		const int voff = lane * 16;
		const int voff_full = voff + i * 1024;
		if constexpr (MODE == 6 || MODE == 7 || MODE == 8) {
			asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"
						 ".if %7 == 6\n\tbuffer_store_dwordx2 v[20:21], %4, %5, %6 offen\n\tbuffer_store_dwordx2 v[22:23], %4, %5, %6 offen offset:8\n\t.endif\n\t"
						 ".if %7 == 7\n\tbuffer_store_dwordx2 v[20:21], %8, %5, 0 offen\n\tbuffer_store_dwordx2 v[22:23], %8, %5, 0 offen offset:8\n\t.endif\n\t"
						 ".if %7 == 8\n\tbuffer_store_dword v20, %4, %5, %6 offen\n\tbuffer_store_dword v21, %4, %5, %6 offen offset:4\n\tbuffer_store_dwordx2 v[22:23], %4, %5, %6 offen offset:8\n\t.endif\n\t"
						 "v_mov_b32 v22, 0x7fc00000\n\t"
						 ".if %7 == 8\n\tv_mov_b32 v21, 0x7fc00000\n\t.endif\n\t"
						 :: "v"(x), "v"(y), "v"(z), "v"(w), "v"(voff), "s"(rs), "s"(soff), "n"(MODE), "v"(voff_full) : "v20", "v21", "v22", "v23", "memory");
		} else if constexpr (MODE == 1) {
			asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"
						 "buffer_store_dwordx4 v[20:23], %4, %5, 0 offen\n\t"
						 "v_mov_b32 v20, 0x7fc00000\n\t"
						 :: "v"(x), "v"(y), "v"(z), "v"(w), "v"(voff_full), "s"(rs) : "v20", "v21", "v22", "v23", "memory");
		} else {
			asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"
						 "buffer_store_dwordx4 v[20:23], %4, %5, %6 offen\n\t"
						 ".if %7 == 2\n\ts_nop 0\n\t.endif\n\t"
						 ".if %7 == 3\n\ts_nop 4\n\t.endif\n\t"
						 ".if %7 == 5\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %1\n\t.endif\n\t"
						 ".if %7 != 4\n\tv_mov_b32 v20, 0x7fc00000\n\t.endif\n\t"
						 :: "v"(x), "v"(y), "v"(z), "v"(w), "v"(voff), "s"(rs), "s"(soff), "n"(MODE) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
		}
	}
}

int main(int argc, char** argv) {
	const int iters = argc > 1 ? atoi(argv[1]) : 2048;
	const int blocks = 256 * 4, threads = 256;
	const size_t waves = (size_t)blocks * threads / 64;
	const size_t n = waves * iters * 256;
	float* d;
	if (hipMalloc(&d, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
	std::vector<float> h(n);
	const char* names[9] = {"SGPR soffset, overwrite at once", "imm soffset, overwrite at once", "SGPR soffset, s_nop 0", "SGPR soffset, s_nop 4", "no overwrite (control)",
							"SGPR soffset, overwrite after 2 VALU", "dwordx2, SGPR soffset, overwrite (of the last store's first dword) at once", "dwordx2, imm soffset, overwrite at once",
							"dword + dword + dwordx2, SGPR soffset, overwrite at once"};
	for (int mode = 0; mode < 9; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			hipMemset(d, 0, n * 4);
			switch (mode) {
			case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 5: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 6: hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 7: hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			case 8: hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(threads), 0, 0, d, iters); break;
			}
			if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
			hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
			size_t bad_x = 0, bad_other = 0;
			for (size_t i = 0; i < n; i += 4) {
				const unsigned lane = (i / 4) & 63;
				if (h[i] != 1000.f + lane) ++bad_x;
				if (h[i + 1] != 2000.f + lane || h[i + 2] != 3000.f + lane || h[i + 3] != 4000.f + lane) ++bad_other;
			}
			printf("mode %d (%s) run %d: %zu of %zu stores with a wrong first dword, %zu with another wrong dword\n", mode, names[mode], rep, bad_x, n / 4, bad_other);
		}
	}
	return 0;
}
