// Persistent LDS-DMA GEMM for the FIND MLP (gfx950):  Y[(foot,v), 0:256] = epi( sum_k A[(foot,v), k] * W[n, k] ).
//
// Structure (one workgroup of 4 waves per CU, one wave per SIMD):
//   * each workgroup owns a contiguous range of BM-row tiles (tiles never straddle two feet);
//   * operands stream HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write) into a
//     3-stage ring of [A: BM x 32 | W: 256 x 32] fp32 chunks; chunk g+2 is issued while chunk g is multiplied,
//     and the stream runs across tile boundaries, so only the very first chunk of a workgroup is exposed;
//   * LDS rows are 128 B (32 floats) with the 16-B slot index XOR-swizzled by ((row>>1)&7): the DMA destination
//     stays lane-linear (swizzle applied to the per-lane SOURCE address) and every ds_read_b128 of 16 rows hits
//     16 distinct slots of the 256-B bank row (cdna_hip_programming.md T2 / rule 21);
//   * the DMA is issued from inline asm: hipcc's waitcnt pass otherwise drains vmcnt(0) before every new
//     LDS-DMA batch and before the first ds_read of every chunk (conservative LDS alias tracking), which
//     serialises load and multiply.  Completion is tracked by hand with counted s_waitcnt vmcnt(N): VMEM
//     operations of a wave return in order on gfx9-class hardware (stores included), so "at most N outstanding"
//     means "everything older than the newest N has landed".  Compiler-issued loads/stores only ever add NEWER
//     operations behind a DMA batch, which keeps every hand-written N conservative;
//   * one s_barrier per chunk;  MFMA: v_mfma_f32_32x32x2_f32, exact fp32;  A/B fragments double-buffered in VGPRs.
#pragma once
#include <type_traits>

#include "mlp_kernels.h"

namespace find {
namespace mlp {

struct Gemm2Args {
	const float* a0;        // A for K-segment 0: rows (foot, v), row stride lda
	const float* a1;        // A for K-segment 1 (nseg == 2)
	int nseg;
	int64_t a_foot_stride;  // elements between feet; 0 = rows shared by all feet
	int lda;
	const float* pos;       // AMODE_PE: (.., V, 3)
	int64_t pos_foot_stride;
	const float* Bm;        // AMODE_PE: (3, pe)
	int pe;
	const float* w0;        // (256, ldw) K-contiguous weight of segment 0
	const float* w1;
	int ldw;
	int w_tr;               // gemm7 only: w0 is the layer's weight as the model holds it, the launch multiplies by its transpose (W_eff[n][k] = w0[k * ldw + n])
	int nchunk;             // 32-wide K chunks per segment
	const float* bias;      // EPI_BIAS_RELU
	int64_t bias_foot_stride;
	const float* mask;      // EPI_MASK: same layout as y
	int64_t mask_foot_stride;
	float* y;
	int64_t y_foot_stride;
	int ldy;
	int V;                  // rows per foot
	int tiles_per_foot;
	int ntiles;
	const float* va_bias;   // gemm5_kernel<.., VIRT>: bias rows of the virtual A operand relu(a0[v] + va_bias[foot]) (mlp_gemm5.h)
	int64_t va_bias_stride;
	const float* vm_bias;   // ... of the virtual ReLU mask (mask = the shared fp32 product)
	int64_t vm_bias_stride;
	int tile_major;         // gemm5 / gemm7<.., FSUM>: unit = tile * n_feet + foot instead of foot * tiles_per_foot + tile
	float* fs_out;          // gemm7<.., FSUM>: [2][V][ldy] sums over the feet (slot 0 + slot 1 = the sum: mlp_gemm7.h)
	int64_t fs_slot_stride; // floats between the two slots
	float* cs_out;          // gemm7<.., FSUM>: [workgroup pair][foot][256] per-foot column sums
	int ablate;             // profiling only: bit0 skip DMA issue, bit1 skip epilogue stores, bit2 skip MFMAs
	unsigned long long* dbg; // profiling only: per-workgroup [total, wait+barrier, epilogue, lgkm-wait] shader cycles (wave 0)
};

#define FIND_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1024).
// M0 carries the wave-uniform LDS base and is restored (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_dst) {
	unsigned keep;
	asm volatile(
		"s_mov_b32 %0, m0\n\t"
		"s_mov_b32 m0, %2\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %1, off\n\t"
		"s_mov_b32 m0, %0"
		: "=&s"(keep)
		: "v"(gsrc), "s"(lds_dst)
		: "memory");
}

template <int BM, int AMODE, int EPI>
__global__ __launch_bounds__(256, 1) void gemm2_kernel(const Gemm2Args g) {
	constexpr int MI = BM / 64;           // waves 2(M) x 2(N): wave tile (BM/2) x 128
	constexpr int NI = 4;
	constexpr int A_BYTES = BM * 128;
	constexpr int B_BYTES = 256 * 128;
	constexpr int STAGE = A_BYTES + B_BYTES;
	constexpr int NA = (AMODE == AMODE_MAT) ? BM / 32 : 0;  // A DMA instructions per wave per chunk
	constexpr int NB = 8;
	constexpr int ND = NA + NB;
	static_assert(ND == 8 || ND == 10 || ND == 12, "vmcnt immediates below assume these");

	extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][STAGE] then Bl[3*256]
	float* Bl = reinterpret_cast<float*>(smem + 3 * STAGE);
	const unsigned lds_base = (unsigned)(uintptr_t)smem;  // low 32 bits of the generic address = LDS byte offset

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1;
	const int t0 = (int)((int64_t)blockIdx.x * g.ntiles / gridDim.x);
	const int t1 = (int)((int64_t)(blockIdx.x + 1) * g.ntiles / gridDim.x);
	if (t0 >= t1) return;
	const int NC = g.nseg * g.nchunk;

	if constexpr (AMODE == AMODE_PE) {
		for (int i = tid; i < 3 * g.pe; i += 256) Bl[i] = g.Bm[i];
		__syncthreads();
	}

	// per-lane constants of the DMA source pattern: lane -> (row-in-8 = lane>>3, slot = lane&7)
	const int l_r8 = lane >> 3, l_slot = lane & 7;
	// per-lane constants of the fragment reads
	const int fsw = ((lane & 31) >> 1) & 7;
	const int fh = lane >> 5;
	const int a_frag = (wm * (BM / 2) + (lane & 31)) * 128;
	const int b_frag = A_BYTES + (wn * 128 + (lane & 31)) * 128;

	float px[BM / 32], py[BM / 32], pz[BM / 32];

	int issued = 0, consumed = 0;   // chunk counters of this workgroup's stream
	int pt = t0, pc = 0;            // next chunk to issue: tile pt, chunk pc

	auto issue_chunk = [&]() {
		const int stage = issued % 3;
		const unsigned sbo = lds_base + stage * STAGE;
		const int seg = (pc >= g.nchunk) ? 1 : 0;
		const int c = pc - seg * g.nchunk;
		const int foot = pt / g.tiles_per_foot;
		const int v0 = (pt - foot * g.tiles_per_foot) * BM;
		// W chunk: 32 DMA instructions of 8 rows each, 8 per wave
		const float* wb = (seg ? g.w1 : g.w0) + c * KC;
#pragma unroll
		for (int j = 0; j < NB; ++j) {
			const int q = wave * NB + j;
			const int n = q * 8 + l_r8;
			const int col = l_slot ^ ((n >> 1) & 7);
			dma16(wb + (int64_t)n * g.ldw + col * 4, sbo + A_BYTES + q * 1024);
		}
		if constexpr (AMODE == AMODE_MAT) {
			const float* ab = (seg ? g.a1 : g.a0) + (int64_t)foot * g.a_foot_stride + c * KC;
#pragma unroll
			for (int j = 0; j < NA; ++j) {
				const int q = wave * NA + j;
				const int r = q * 8 + l_r8;
				const int v = min(v0 + r, g.V - 1);  // rows past the end of a foot re-read its last row; never stored
				const int col = l_slot ^ ((r >> 1) & 7);
				dma16(ab + (int64_t)v * g.lda + col * 4, sbo + q * 1024);
			}
		} else {
			// Fourier features generated in registers and written into the same swizzled image
			char* sb = smem + stage * STAGE;
#pragma unroll
			for (int i = 0; i < BM / 32; ++i) {
				const int r = (tid >> 3) + 32 * i;
				const int col = (tid & 7) ^ ((r >> 1) & 7);
				const int kp = c * KC + col * 4;
				float4 v;
				v.x = pe_value(kp + 0, g.pe, px[i], py[i], pz[i], Bl);
				v.y = pe_value(kp + 1, g.pe, px[i], py[i], pz[i], Bl);
				v.z = pe_value(kp + 2, g.pe, px[i], py[i], pz[i], Bl);
				v.w = pe_value(kp + 3, g.pe, px[i], py[i], pz[i], Bl);
				*reinterpret_cast<float4*>(sb + r * 128 + (tid & 7) * 16) = v;
			}
		}
		++issued;
		if (++pc == NC) { pc = 0; ++pt; }
	};

	// chunk iterations since the last epilogue's stores were issued (they sit between in-flight DMAs only when the
	// stream runs across tiles, i.e. AMODE_MAT); tiny K falls back to full drains
	int since_epi = 99;
	const bool simple_wait = NC < 3;

	for (int t = t0; t < t1; ++t) {
		const int foot = t / g.tiles_per_foot;
		const int v0 = (t - foot * g.tiles_per_foot) * BM;
		if constexpr (AMODE == AMODE_PE) {
			const float* pp = g.pos + (int64_t)foot * g.pos_foot_stride;
#pragma unroll
			for (int i = 0; i < BM / 32; ++i) {
				const int v = min(v0 + (tid >> 3) + 32 * i, g.V - 1);
				px[i] = pp[(int64_t)v * 3 + 0];
				py[i] = pp[(int64_t)v * 3 + 1];
				pz[i] = pp[(int64_t)v * 3 + 2];
			}
		}
		// epilogue operands are fetched early so their latency hides under the K loop
		float bv[NI];
		if constexpr (EPI == EPI_BIAS_RELU) {
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) bv[ni] = g.bias[(int64_t)foot * g.bias_foot_stride + wn * 128 + ni * 32 + (lane & 31)];
		}
		float mv[MI][NI][16];
		const float* mp = (EPI == EPI_MASK) ? g.mask + (int64_t)foot * g.mask_foot_stride : nullptr;

		// (re)fill the pipeline: no-op in steady state for AMODE_MAT, per-tile prologue for AMODE_PE
		while (issued < consumed + 2 && pt < t1 && (AMODE == AMODE_MAT || pt == t)) issue_chunk();

		f32x16 acc[MI][NI];
#pragma unroll
		for (int mi = 0; mi < MI; ++mi)
#pragma unroll
			for (int ni = 0; ni < NI; ++ni)
#pragma unroll
				for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

		for (int cc = 0; cc < NC; ++cc) {
			// ---- W: chunk `consumed` has landed (own DMAs), then everybody's
			const bool has_next = issued > consumed + 1;
			if (!has_next || simple_wait) {
				FIND_WAIT_VMCNT(0);
			} else if (since_epi < 2) {
				FIND_WAIT_VMCNT(63);  // >= 63 newer ops exist: ND DMAs + >= 64 epilogue stores
			} else {
				if constexpr (ND == 8) FIND_WAIT_VMCNT(8);
				else if constexpr (ND == 10) FIND_WAIT_VMCNT(10);
				else FIND_WAIT_VMCNT(12);
			}
			if constexpr (AMODE == AMODE_PE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			++since_epi;
			if constexpr (EPI == EPI_MASK) {
				// ReLU mask of this tile: requested one chunk before the end so it is in registers by the epilogue
				if (cc == max(NC - 2, 0)) {
#pragma unroll
					for (int ni = 0; ni < NI; ++ni)
#pragma unroll
						for (int mi = 0; mi < MI; ++mi)
#pragma unroll
							for (int r = 0; r < 16; ++r) {
								const int row = wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
								const int v = min(v0 + row, g.V - 1);
								mv[mi][ni][r] = mp[(int64_t)v * g.ldy + wn * 128 + ni * 32 + (lane & 31)];
							}
				}
			}
			// ---- I: prefetch two chunks ahead (may belong to the next tile)
			if (pt < t1 && (AMODE == AMODE_MAT || pt == t)) issue_chunk();
			// ---- C: multiply chunk `consumed`; fragments for step j+1 are requested before the MFMAs of step j
			const char* sb = smem + (consumed % 3) * STAGE;
			float4 af[2][MI], bf[2][NI];
			{
				const int slot = ((0 + fh) ^ fsw) * 16;
#pragma unroll
				for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const float4*>(sb + a_frag + mi * 32 * 128 + slot);
#pragma unroll
				for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *reinterpret_cast<const float4*>(sb + b_frag + ni * 32 * 128 + slot);
			}
#pragma unroll
			for (int j = 0; j < KC / 8; ++j) {
				const int cur = j & 1, nxt = cur ^ 1;
				if (j + 1 < KC / 8) {
					const int slot = ((2 * (j + 1) + fh) ^ fsw) * 16;
#pragma unroll
					for (int mi = 0; mi < MI; ++mi) af[nxt][mi] = *reinterpret_cast<const float4*>(sb + a_frag + mi * 32 * 128 + slot);
#pragma unroll
					for (int ni = 0; ni < NI; ++ni) bf[nxt][ni] = *reinterpret_cast<const float4*>(sb + b_frag + ni * 32 * 128 + slot);
				}
#pragma unroll
				for (int mi = 0; mi < MI; ++mi)
#pragma unroll
					for (int ni = 0; ni < NI; ++ni)
						acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi].x, bf[cur][ni].x, acc[mi][ni], 0, 0, 0);
#pragma unroll
				for (int mi = 0; mi < MI; ++mi)
#pragma unroll
					for (int ni = 0; ni < NI; ++ni)
						acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi].y, bf[cur][ni].y, acc[mi][ni], 0, 0, 0);
#pragma unroll
				for (int mi = 0; mi < MI; ++mi)
#pragma unroll
					for (int ni = 0; ni < NI; ++ni)
						acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi].z, bf[cur][ni].z, acc[mi][ni], 0, 0, 0);
#pragma unroll
				for (int mi = 0; mi < MI; ++mi)
#pragma unroll
					for (int ni = 0; ni < NI; ++ni)
						acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi].w, bf[cur][ni].w, acc[mi][ni], 0, 0, 0);
			}
			++consumed;
		}

		// ---- E: epilogue.  Lane holds column (lane&31) of 16 rows per 32x32 block: each store instruction
		// writes two full 128-B row segments.
		float* yp = g.y + (int64_t)foot * g.y_foot_stride;
		auto store_tile = [&](auto guard) {
			constexpr bool GUARD = decltype(guard)::value;
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) {
				const int col = wn * 128 + ni * 32 + (lane & 31);
#pragma unroll
				for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						const int row = wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
						const int v = v0 + row;
						if (!GUARD || v < g.V) {
							float val = acc[mi][ni][r];
							if constexpr (EPI == EPI_BIAS_RELU) val = fmaxf(val + bv[ni], 0.f);
							if constexpr (EPI == EPI_MASK) val = (mv[mi][ni][r] > 0.f) ? val : 0.f;
							yp[(int64_t)v * g.ldy + col] = val;
						}
					}
				}
			}
		};
		if (v0 + BM <= g.V) store_tile(std::false_type{});
		else store_tile(std::true_type{});
		since_epi = (AMODE == AMODE_MAT) ? 0 : 99;
	}
}

template <int BM>
constexpr int gemm2_lds_bytes() { return 3 * (BM * 128 + 256 * 128) + 3 * 256 * 4; }

}  // namespace mlp
}  // namespace find
