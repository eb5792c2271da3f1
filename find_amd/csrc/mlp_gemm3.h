// gemm3_kernel: second-generation persistent LDS-DMA GEMM (matrix A operand only; the Fourier-feature layer keeps
// gemm2_kernel<.., AMODE_PE, ..>).  Same data movement as gemm2 (3-stage LDS ring of [A: BM x 32 | W: 256 x 32] chunks,
// XOR-swizzled 128-B rows, asm-issued global_load_lds with hand-counted vmcnt), but scheduled so the matrix pipe never
// drains at a chunk boundary:
//   * EARLY BARRIER: the wait + s_barrier that publishes chunk g+1 sits before the LAST 16-MFMA block of chunk g
//     (all of chunk g's fragment reads have completed by then); the first 4 MFMAs of that block are issued immediately
//     after the barrier, and chunk g+1's first fragment reads and the DMA of chunk g+3 are issued under them;
//   * nothing scalar-heavy sits between the barrier and those MFMAs: kernel arguments are hoisted into registers (a
//     scalar load inside the loop shares lgkmcnt with the LDS reads and would drain them), tile coordinates are advanced
//     incrementally (one integer division per tile, before the barrier), ring stages are counters not modulos;
//   * DMA addressing = wave-uniform SGPR base + per-lane 32-bit VGPR offsets that are CONSTANT for the whole kernel
//     (W) / for every full tile (A): no 64-bit address arithmetic per DMA; M0 is stepped with s_add inside one asm block;
//   * no MFMA sits inside a conditional (accumulators must not become phi nodes: that costs v_accvgpr_mov per chunk);
//   * epilogue = buffer stores with SGPR/immediate offsets and the SRD size as the row bound (no exec masking).
#pragma once
#include "mlp_gemm2.h"

namespace find {
namespace mlp {

// make wave-uniformity provable to the compiler so the value may sit in an "s" asm operand (cdna_hip_programming.md T20)
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
	const uint64_t b = reinterpret_cast<uint64_t>(p);
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
	const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
	return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ char* uniform_ptr(const char* p) {
	return const_cast<char*>(reinterpret_cast<const char*>(uniform_ptr(reinterpret_cast<const float*>(p))));
}

// 4 LDS-DMA instructions: lanes fetch 16 B from base + off_i, landing at LDS [dst + i*1024, +1024)
__device__ __forceinline__ void dma4(const float* base, unsigned dst, unsigned o0, unsigned o1, unsigned o2, unsigned o3) {
	unsigned keep;
	asm volatile(
		"s_mov_b32 %0, m0\n\t"
		"s_mov_b32 m0, %6\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %1, %5\n\t"
		"s_add_u32 m0, m0, 0x400\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %2, %5\n\t"
		"s_add_u32 m0, m0, 0x400\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %3, %5\n\t"
		"s_add_u32 m0, m0, 0x400\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %4, %5\n\t"
		"s_mov_b32 m0, %0"
		: "=&s"(keep)
		: "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(base), "s"(dst)
		: "memory", "scc");
}

__device__ __forceinline__ void dma2(const float* base, unsigned dst, unsigned o0, unsigned o1) {
	unsigned keep;
	asm volatile(
		"s_mov_b32 %0, m0\n\t"
		"s_mov_b32 m0, %4\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %1, %3\n\t"
		"s_add_u32 m0, m0, 0x400\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %2, %3\n\t"
		"s_mov_b32 m0, %0"
		: "=&s"(keep)
		: "v"(o0), "v"(o1), "s"(base), "s"(dst)
		: "memory", "scc");
}

template <int BM, int EPI>
// (launch bounds of two workgroups per CU although the ring admits one: that caps the kernel at 256 registers -- the masked variant took 328
// when it was allowed to -- in line with the rule that came out of the co-residence fault, mlp_kernels.h.  The 328-register build itself ran
// clean under the stress configuration: the cap is policy, not a fix.)
#ifndef FIND_GEMM3_MIN_WGS
#define FIND_GEMM3_MIN_WGS 2   // (-DFIND_GEMM3_MIN_WGS=1 rebuilds the 328-register variant)
#endif
__global__ __launch_bounds__(256, FIND_GEMM3_MIN_WGS) void gemm3_kernel(const Gemm2Args g) {
	constexpr int MI = BM / 64;
	constexpr int NI = 4;
	constexpr int A_BYTES = BM * 128;
	constexpr int B_BYTES = 256 * 128;
	constexpr int STAGE = A_BYTES + B_BYTES;
	constexpr int NA = BM / 32;
	constexpr int ND = NA + 8;
	static_assert(BM == 64 || BM == 128, "tile heights supported");

	extern __shared__ __attribute__((aligned(16))) char smem[];
	const unsigned lds_base = (unsigned)(uintptr_t)smem;

	// ---- kernel arguments hoisted once (no scalar loads inside the main loop)
	const float* const ga0 = g.a0;
	const float* const ga1 = g.a1;
	const float* const gw0 = g.w0;
	const float* const gw1 = g.w1;
	const int64_t a_foot_stride = g.a_foot_stride;
	const int lda = g.lda, ldw = g.ldw, ldy = g.ldy, V = g.V, nchunk = g.nchunk, tpf = g.tiles_per_foot;
	const int NC = g.nseg * g.nchunk;
	const int ablate = FIND_DIAG_ON ? g.ablate : 0;   // (bits 1, 2: diagnostics build only)
	unsigned long long* const dbg = FIND_DBG(g.dbg);

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1;
	const int t0 = (int)((int64_t)blockIdx.x * g.ntiles / gridDim.x);
	const int t1 = (int)((int64_t)(blockIdx.x + 1) * g.ntiles / gridDim.x);
	if (t0 >= t1) return;
	const int total = (t1 - t0) * NC;

	const int l_r8 = lane >> 3, l_slot = lane & 7;
	const int fsw = ((lane & 31) >> 1) & 7;
	const int fh = lane >> 5;
	const int a_frag = (wm * (BM / 2) + (lane & 31)) * 128;
	const int b_frag = A_BYTES + (wn * 128 + (lane & 31)) * 128;
	int fslot[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) fslot[j] = ((2 * j + fh) ^ fsw) * 16;

	// per-lane DMA byte offsets: W rows are fixed for the whole kernel; A rows are fixed for every full tile
	unsigned woff[8], aoff[NA];
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int n = (wave * 8 + j) * 8 + l_r8;
		woff[j] = (unsigned)((n * ldw + (l_slot ^ ((n >> 1) & 7)) * 4) * 4);
	}
#pragma unroll
	for (int j = 0; j < NA; ++j) {
		const int r = (wave * NA + j) * 8 + l_r8;
		aoff[j] = (unsigned)((r * lda + (l_slot ^ ((r >> 1) & 7)) * 4) * 4);
	}
	const unsigned w_dst_off = A_BYTES + (wave * 8) * 1024;
	const unsigned a_dst_off = (wave * NA) * 1024;

	// ---- prefetch stream state (all wave-uniform): tile pt, chunk pc, ring stage ps; tile base pointers advance 128 B / chunk
	int issued = 0, consumed = 0;
	int pt = t0, pc = 0, ps = 0;
	int p_foot = t0 / tpf;
	int p_tif = t0 - p_foot * tpf;  // tile index within the foot
	const float* pa0 = nullptr;     // A row block of tile pt, segment 0 / 1
	const float* pa1 = nullptr;
	bool p_full = true;
	unsigned pa_off[NA];
	auto tile_setup = [&]() {
		const int v0 = p_tif * BM;
		const int64_t ro = (int64_t)p_foot * a_foot_stride + (int64_t)v0 * lda;
		pa0 = ga0 + ro;
		pa1 = ga1 ? ga1 + ro : nullptr;
		p_full = v0 + BM <= V;
		if (p_full) {
#pragma unroll
			for (int j = 0; j < NA; ++j) pa_off[j] = aoff[j];
		} else {  // last tile of a foot: rows past the end re-read the last valid row (never stored)
#pragma unroll
			for (int j = 0; j < NA; ++j) {
				const int r = (wave * NA + j) * 8 + l_r8;
				const int rc = min(r, V - 1 - v0);
				pa_off[j] = (unsigned)((rc * lda + (l_slot ^ ((r >> 1) & 7)) * 4) * 4);
			}
		}
	};
	tile_setup();
	// operands of the NEXT issue, prepared ahead of the barrier
	const float* iw = nullptr;
	const float* ia = nullptr;
	unsigned idst = 0;
	auto issue_prepare = [&]() {
		const bool seg1 = pc >= nchunk;
		const int c = seg1 ? pc - nchunk : pc;
		iw = uniform_ptr((seg1 ? gw1 : gw0) + c * KC);
		ia = uniform_ptr((seg1 ? pa1 : pa0) + c * KC);
		idst = __builtin_amdgcn_readfirstlane(lds_base + ps * STAGE);
	};
	auto issue_w0 = [&]() { dma4(iw, idst + w_dst_off, woff[0], woff[1], woff[2], woff[3]); };
	auto issue_w1 = [&]() { dma4(iw, idst + w_dst_off + 4096, woff[4], woff[5], woff[6], woff[7]); };
	auto issue_a = [&]() {
		if constexpr (NA == 2) dma2(ia, idst + a_dst_off, pa_off[0], pa_off[1]);
		else dma4(ia, idst + a_dst_off, pa_off[0], pa_off[1], pa_off[2], pa_off[3]);
	};
	auto issue_advance = [&]() {  // scalar bookkeeping only; safe anywhere
		++issued;
		ps = (ps == 2) ? 0 : ps + 1;
		if (++pc == NC) {
			pc = 0;
			++pt;
			if (++p_tif == tpf) { p_tif = 0; ++p_foot; }
			if (pt < t1) tile_setup();
		}
	};

	// ---- prologue: fill the ring, publish chunk 0, read its first fragments
	for (int k = 0; k < 3 && pt < t1 && !(ablate & 1); ++k) { issue_prepare(); issue_w0(); issue_w1(); issue_a(); issue_advance(); }
	if (issued >= 3) {
		if constexpr (ND == 10) FIND_WAIT_VMCNT(20); else FIND_WAIT_VMCNT(24);
	} else if (issued == 2) {
		if constexpr (ND == 10) FIND_WAIT_VMCNT(10); else FIND_WAIT_VMCNT(12);
	} else {
		FIND_WAIT_VMCNT(0);
	}
	__builtin_amdgcn_s_barrier();

	float4 af0[MI], bf0[NI], af1[MI], bf1[NI];
	auto load_frags = [&](const char* sb, int j, float4 (&af)[MI], float4 (&bf)[NI]) {
#pragma unroll
		for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const float4*>(sb + a_frag + mi * 32 * 128 + fslot[j]);
#pragma unroll
		for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const float4*>(sb + b_frag + ni * 32 * 128 + fslot[j]);
	};
	load_frags(smem, 0, af0, bf0);

	int cs = 0;  // ring stage of the chunk being multiplied
	int since_epi = 99;
	unsigned long long c_wait = 0, c_epi = 0, c_lgkm = 0;
	const unsigned long long c_start = dbg ? __builtin_amdgcn_s_memtime() : 0;

	int foot = t0 / tpf;
	int tif = t0 - foot * tpf;
	for (int t = t0; t < t1; ++t) {
		const int v0 = tif * BM;
		float bv[NI];
		if constexpr (EPI == EPI_BIAS_RELU) {
#pragma unroll
			for (int ni = 0; ni < NI; ++ni) bv[ni] = g.bias[(int64_t)foot * g.bias_foot_stride + wn * 128 + ni * 32 + (lane & 31)];
		}
		float mv[MI][NI][16];
		const float* mp = (EPI == EPI_MASK) ? g.mask + (int64_t)foot * g.mask_foot_stride : nullptr;

		f32x16 acc[MI][NI];
#pragma unroll
		for (int mi = 0; mi < MI; ++mi)
#pragma unroll
			for (int ni = 0; ni < NI; ++ni)
#pragma unroll
				for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

		auto mfma_k = [&](const float4 (&af)[MI], const float4 (&bf)[NI], int k) {
#pragma unroll
			for (int mi = 0; mi < MI; ++mi)
#pragma unroll
				for (int ni = 0; ni < NI; ++ni) {
					const float a = k == 0 ? af[mi].x : (k == 1 ? af[mi].y : (k == 2 ? af[mi].z : af[mi].w));
					const float b = k == 0 ? bf[ni].x : (k == 1 ? bf[ni].y : (k == 2 ? bf[ni].z : bf[ni].w));
					acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][ni], 0, 0, 0);
				}
		};
		auto mfma_all = [&](const float4 (&af)[MI], const float4 (&bf)[NI]) {
			mfma_k(af, bf, 0); mfma_k(af, bf, 1); mfma_k(af, bf, 2); mfma_k(af, bf, 3);
		};

		for (int cc = 0; cc < NC; ++cc) {
			const char* sb = smem + cs * STAGE;
			if constexpr (EPI == EPI_MASK) {
				if (cc == max(NC - 2, 0)) {
#pragma unroll
					for (int ni = 0; ni < NI; ++ni)
#pragma unroll
						for (int mi = 0; mi < MI; ++mi)
#pragma unroll
							for (int r = 0; r < 16; ++r) {
								const int row = wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
								const int v = min(v0 + row, V - 1);
								mv[mi][ni][r] = mp[(int64_t)v * ldy + wn * 128 + ni * 32 + (lane & 31)];
							}
				}
			}
			// k-groups 0..2 of this chunk; fragments of group j+1 are requested before the MFMAs of group j
			load_frags(sb, 1, af1, bf1);
			mfma_all(af0, bf0);
			load_frags(sb, 2, af0, bf0);
			mfma_all(af1, bf1);
			load_frags(sb, 3, af1, bf1);
			// scalar preparation of the next publish / issue, under the MFMAs below and BEFORE the barrier
			const bool more = consumed + 1 < total;
			const bool do_issue = more && pt < t1 && !(ablate & 1);
			if (do_issue) issue_prepare();
			const int ns = (cs == 2) ? 0 : cs + 1;
			const char* nsb = smem + ns * STAGE;
			const int ahead = issued - (consumed + 2);  // chunks in flight beyond the one being published
			mfma_all(af0, bf0);
			// ---- early barrier
			if (more) {
				const unsigned long long c0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				const unsigned long long c1 = dbg ? __builtin_amdgcn_s_memtime() : 0;
				if (ahead <= 0) {
					FIND_WAIT_VMCNT(0);
				} else if (since_epi < 2) {
					FIND_WAIT_VMCNT(63);  // newer ops: ND DMAs + >= 64 epilogue stores
				} else {
					if constexpr (ND == 10) FIND_WAIT_VMCNT(10); else FIND_WAIT_VMCNT(12);
				}
				__builtin_amdgcn_s_barrier();
				++since_epi;
				if (dbg) { const unsigned long long c2 = __builtin_amdgcn_s_memtime(); c_wait += c2 - c1; c_lgkm += c1 - c0; }
			}
			__builtin_amdgcn_sched_barrier(0);
			mfma_k(af1, bf1, 0);  // issued straight after the barrier: depends on nothing new
			__builtin_amdgcn_sched_barrier(0);
			// first fragments of the next chunk (at the end of the stream this reads a stale stage; never used)
			load_frags(nsb, 0, af0, bf0);
			if (do_issue) issue_w0();
			__builtin_amdgcn_sched_barrier(0);
			mfma_k(af1, bf1, 1);
			__builtin_amdgcn_sched_barrier(0);
			if (do_issue) issue_w1();
			__builtin_amdgcn_sched_barrier(0);
			mfma_k(af1, bf1, 2);
			__builtin_amdgcn_sched_barrier(0);
			if (do_issue) issue_a();
			__builtin_amdgcn_sched_barrier(0);
			mfma_k(af1, bf1, 3);
			if (do_issue) issue_advance();
			cs = ns;
			++consumed;
		}

		// ---- epilogue: buffer stores.  Address = SRD base (tile origin) + per-lane voffset (fixed) + SGPR soffset +
		// immediate, so no per-store address arithmetic; the SRD's size is the number of VALID bytes of the tile, so rows
		// past the end of a foot are dropped by the hardware bounds check instead of exec masking.
		const unsigned long long ce0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
		{
			float* ytile = g.y + (int64_t)foot * g.y_foot_stride + (int64_t)v0 * ldy;
			const int valid_rows = min(BM, V - v0);
			const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(ytile)), 0, valid_rows * ldy * 4, 0x00020000);
			const int voff = ((wm * (BM / 2) + 4 * fh) * ldy + wn * 128 + (lane & 31)) * 4;
#pragma unroll
			for (int ni = 0; ni < NI; ++ni)
#pragma unroll
				for (int mi = 0; mi < MI; ++mi)
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						float val = acc[mi][ni][r];
						if constexpr (EPI == EPI_BIAS_RELU) val = fmaxf(val + bv[ni], 0.f);
						if constexpr (EPI == EPI_MASK) val = (mv[mi][ni][r] > 0.f) ? val : 0.f;
						const int soff = ((mi * 32 + 8 * (r >> 2)) * ldy) * 4;       // wave-uniform
						const int ioff = ((r & 3) * ldy + ni * 32) * 4;
						if (!(ablate & 2)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc, voff + ioff, soff, 0);
					}
		}
		if (dbg) c_epi += __builtin_amdgcn_s_memtime() - ce0;
		since_epi = 0;
		if (++tif == tpf) { tif = 0; ++foot; }
	}
	if (dbg && tid == 0) {
		unsigned long long* o = dbg + (size_t)blockIdx.x * 4;
		o[0] = __builtin_amdgcn_s_memtime() - c_start; o[1] = c_wait; o[2] = c_epi; o[3] = c_lgkm;
	}
}

}  // namespace mlp
}  // namespace find
