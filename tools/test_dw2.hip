// Standalone check of dw2_kernel against a host reference.  hipcc --offload-arch=gfx950 -O3 -I../find_amd/csrc -I../include
#include "mlp_dw2.h"
#include <vector>
#include <cmath>
namespace find { void set_error(const char*, ...) {} }
using namespace find::mlp;
int main(int argc, char** argv) {
	const int rows = argc > 3 ? atoi(argv[3]) : 42;
	std::vector<float> Z(rows * 256), X(rows * 256);
	const int mode = argc > 1 ? atoi(argv[1]) : 0; const int r0 = argc > 2 ? atoi(argv[2]) : 0;
	for (int r = 0; r < rows; ++r) for (int c = 0; c < 256; ++c) { if (mode == 0) { Z[r * 256 + c] = (float)((r * 7 + c * 3) % 11 - 5); X[r * 256 + c] = (float)((r * 5 + c) % 13 - 6); } else if (mode == 1) { Z[r*256+c] = r == r0 ? c : 0; X[r*256+c] = r == r0 ? 1 : 0; } else { Z[r*256+c] = r == r0 ? 1 : 0; X[r*256+c] = r == r0 ? c : 0; } }
	float *dz, *dx, *pw, *pb;
	hipMalloc(&dz, Z.size() * 4); hipMalloc(&dx, X.size() * 4); hipMalloc(&pw, 65536 * 4 * 2); hipMalloc(&pb, 256 * 4 * 2);
	hipMemcpy(dz, Z.data(), Z.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
	hipFuncSetAttribute(reinterpret_cast<const void*>(&dw2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW2_LDS);
	Dw2Args a{}; a.dz = dz; a.dz_foot_stride = 0; a.x = dx; a.x_foot_stride = 0; a.chunks_per_foot = rows / 16; a.tail_rows = rows % 16; a.spf = 1; a.cps = rows / 16 > 0 ? rows / 16 : 1; a.pw = pw; a.pb = pb;
	hipLaunchKernelGGL(dw2_kernel, dim3(1), dim3(256), DW2_LDS, 0, a);
	std::vector<float> out(65536), ob(256);
	hipMemcpy(out.data(), pw, 65536 * 4, hipMemcpyDeviceToHost); hipMemcpy(ob.data(), pb, 256 * 4, hipMemcpyDeviceToHost);
	printf("err %s\n", hipGetErrorString(hipGetLastError()));
	int bad = 0; double maxe = 0;
	for (int n = 0; n < 256; ++n) for (int k = 0; k < 256; ++k) {
		double ref = 0; for (int r = 0; r < rows; ++r) ref += (double)Z[r * 256 + n] * X[r * 256 + k];
		const double e = std::fabs(out[n * 256 + k] - ref); if (e > maxe) maxe = e;
		if (e > 1e-3 && bad < 8) { printf("n=%d k=%d got %g ref %g\n", n, k, out[n * 256 + k], ref); ++bad; }
	}
	double be = 0; for (int n = 0; n < 256; ++n) { double ref = 0; for (int r = 0; r < rows; ++r) ref += Z[r * 256 + n]; be = std::fmax(be, std::fabs(ob[n] - ref)); }
	printf("max err W %g  bias %g\n", maxe, be);
	// find where out[0][0..7] values appear in the reference (to spot permutations)
	if (mode) { for (int n = 0; n < 40; ++n) { printf("n=%d:", n); for (int k = 0; k < 12; ++k) printf(" %g", out[n * 256 + k]); printf("\n"); } }
	return 0;
}
