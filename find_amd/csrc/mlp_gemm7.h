// gemm7_kernel: the K = 256 Linear layer in bf16x3 arithmetic (fp32-faithful on the bf16 matrix pipe: mlp_gemm6.h says what that is and
// why its error is that of fp32 accumulation alone), built around what bounded gemm6 (tools/ablate_x3.py: of its 93 us at the C2 shape,
// 34 were the A loads -- every lane reading its own row: 32 cache lines per load instruction, four column groups re-reading and
// RE-SPLITTING the same rows --, 17 the LDS traffic of the weight fragments, 15 the split arithmetic; the matrix pipe 50 % busy):
//   * THE WEIGHTS STAY IN REGISTERS.  A wave owns 16 columns of the output for its whole life and holds those rows of W for all of K
//     as MFMA operands: 8 k-steps x 3 bf16 planes x 4 registers = 96 ACCUMULATION registers (the MFMA reads its row operand from there
//     directly; `asm("" : "+a")` pins them: left alone, the allocator keeps them in VGPRs, runs out, and copies them back and forth),
//     split once in the prologue.  No weight fragment is ever read again -- from LDS or anywhere else;
//   * THE ACTIVATIONS GO THROUGH LDS ONCE PER WORKGROUP.  The 8 waves of a workgroup (two per SIMD; 128 columns: the two column halves
//     of the same rows run 8 blocks apart, on the same XCD) take a 32-row unit together: every wave loads whole rows (64 lanes x 16 B =
//     one contiguous KB per instruction), each thread splits the 16 values it loaded -- every element is split ONCE per workgroup --
//     and writes the three bf16 planes row-major into LDS (rows padded to 544 B: the fragment reads are conflict-free); the MFMA column
//     operands are then 6 ds_read_b128 per k-step and wave (v_mfma_f32_16x16x32_bf16: two 16-row blocks x three planes);
//   * two unit buffers (2 x 50 KB): the unit after the one being multiplied is split and stored, the one three units ahead is loaded
//     and the block of the previous unit is stored UNDER the MFMAs of the current one, a piece behind every other MFMA (written down in
//     this order and pinned with sched_barrier: the scheduler's own order put the MFMAs in one run and the VALU work in another); one
//     barrier per unit;
//   * the weights are the MFMA's ROW operand, so a block comes out transposed: lane (row i, quarter h) holds four consecutive COLUMNS
//     of its row -- the epilogue is one 16-byte store per 16-row block (with four-byte stores it was a third of the kernel's time);
//   * fewer than 256 registers per lane: two waves per SIMD hide each other's waits (the first version of this kernel -- four waves with
//     32 columns each, 480 registers -- had the matrix pipe 45 % busy with nothing else above 20 %: a lone wave per SIMD stalls on every
//     dependency; tools/mfma_bf16_mix.hip: three VALU instructions per MFMA cost a lone wave 26 %, two waves 12 %).
// Epilogues as gemm4: bias + ReLU (the bias initialises the accumulator), the ReLU mask of the backward's dX, or none.
#pragma once
#include "mlp_gemm6.h"

namespace find {
namespace mlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G7_ROW = 544;                    // bytes per activation row of one plane in LDS: 256 bf16 + 32 B of padding (tools/lds_b128_probe.hip:
                                               // the 16-row x 4-quarter fragment reads cost 8.3 cycles at a stride of 528 B, 7.2 -- conflict-free -- at 544)
constexpr int G7_PLANE = 32 * G7_ROW;          // 17 408 B
constexpr int G7_BUF = 3 * G7_PLANE;           // 52 224 B
constexpr int GEMM7_LDS = 2 * G7_BUF;          // 104 448 B
constexpr int GEMM7_NW = 8;

// ABL: profiling only (tools/ablate_x3.py; results are wrong under every bit): 1 = no split / LDS writes, 2 = no fragment reads after a unit's
// first, 4 = no activation loads after the prologue, 8 = no stores
// VIRT ("bcast_fold", mlp.hip): the layer behind a head's broadcast first layer forms that layer's output h1 = relu(P[v] + bias[foot]) itself
// instead of reading n_feet x V x 1 KB of it from HBM -- from the V x 256 product P (a0 / mask with foot stride 0: L2) and the foot's bias
// row: EPI_BIAS_RELU: the A operand, two packed adds and four max per staged row in front of its split (g.va_bias); EPI_MASK: the ReLU
// mask of the epilogue (g.vm_bias).  Bit for bit the values bias_relu_bcast_kernel would have stored.
// FSUM ("footsum_fold", mlp.hip; with the virtual mask only): the dX GEMM of a head's second layer produces dZ of the BROADCAST first layer,
// of which the backward reads nothing but two sums -- over the feet (what the shared trunk receives) and, per foot, over the rows (what the
// latent codes and the bias receive).  Both are formed here and dZ (n_feet x V x 1 KB) is never stored: units run TILE-major (consecutive
// units = consecutive feet of the same 32 rows); a wave adds the masked blocks of a tile's feet in its registers (its 16 columns: no
// exchange between waves) and stores the sum when the tile -- or the workgroup's unit range -- ends: slot 0 of g.fs_out for a run that
// began at the tile's first foot, slot 1 for one that did not (every range holds at least n_feet units, so a tile has at most two runs; a
// run that covers a whole tile writes zeros to slot 1); the per-foot column sums go through 16-lane DPP sums into an LDS table
// [foot][this workgroup's 128 columns] (each wave its own columns, in unit order) and out to g.cs_out [range][foot][256] at the end.
// Every sum in a fixed order: bit-reproducible.
template <int EPI, int ABL = 0, bool VIRT = false, bool FSUM = false>
__global__ __launch_bounds__(GEMM7_NW * 64, 1) void gemm7_kernel(const Gemm2Args g) {
	static_assert(!VIRT || EPI != EPI_NONE, "gemm7_kernel: a virtual operand needs the epilogue it belongs to");
	static_assert(!FSUM || (VIRT && EPI == EPI_MASK), "gemm7_kernel: the folded foot sum belongs to the virtual-mask dX GEMM");
	constexpr bool VA = VIRT && EPI == EPI_BIAS_RELU, VM = VIRT && EPI == EPI_MASK;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int i16 = lane & 15, h4 = lane >> 4;
	const int b = blockIdx.x;
	const int npairs = gridDim.x / 2;
	const int pair = (b / 16) * 8 + (b & 7);      // the two column halves of the same rows are 8 blocks apart: same XCD
	const int col0 = ((b >> 3) & 1) * 128 + wave * 16;
	const int V = g.V, lda = g.lda, ldy = g.ldy, upf = g.tiles_per_foot;
	const int u0 = (int)((int64_t)pair * g.ntiles / npairs);
	const int u1 = (int)((int64_t)(pair + 1) * g.ntiles / npairs);
	if (u0 >= u1) return;
	const int nf = g.ntiles / upf;   // feet
	// unit -> (foot, first row): foot-major, or -- FSUM -- tile-major
	auto decode = [&](int uu, int& foot, int& v0) {
		if constexpr (FSUM) { const int t = uu / nf; foot = uu - t * nf; v0 = t * 32; }
		else { foot = uu / upf; v0 = (uu - foot * upf) * 32; }
	};

	// ---- prologue: this wave's 16 rows of W, all of K, as MFMA row operands: lane (n = i16, quarter h4), k-step s: W[col0 + i16][32 s + 8 h4 .. +8]
	bf16x8 B1[8], B2[8], B3[8];
	{
		const float4* wrow = reinterpret_cast<const float4*>(g.w0 + (int64_t)(col0 + i16) * g.ldw + h4 * 8);
		const float* wcol = g.w0 + (int64_t)(h4 * 8) * g.ldw + col0 + i16;   // (w_tr: the dX launches read the model's weight itself -- no transposed copy)
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			float4 lo, hi;
			if (g.w_tr) {
				const float* q = wcol + (int64_t)(s * 32) * g.ldw;
				lo = make_float4(q[0], q[g.ldw], q[2 * (int64_t)g.ldw], q[3 * (int64_t)g.ldw]);
				hi = make_float4(q[4 * (int64_t)g.ldw], q[5 * (int64_t)g.ldw], q[6 * (int64_t)g.ldw], q[7 * (int64_t)g.ldw]);
			} else {
				lo = wrow[s * 8]; hi = wrow[s * 8 + 1];
			}
			split3(lo, hi, B1[s], B2[s], B3[s]);
			asm volatile("" : "+a"(B1[s]));
			asm volatile("" : "+a"(B2[s]));
			asm volatile("" : "+a"(B3[s]));
		}
	}

	// ---- staging: thread (wave w, lane l) loads the float4 at columns 4 l .. 4 l + 3 of rows 8 r + w, r = 0 .. 3, of a unit
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	typedef unsigned u2 __attribute__((ext_vector_type(2)));
	u4 st[2][4];   // two units in flight: a load is consumed two units later (HBM latency under load is 1-2 us, a unit takes ~1.5 us)
	// (Tried: issuing the activation loads through inline asm with hand-counted s_waitcnt vmcnt(N), because the compiler starts every unit
	// with vmcnt(0) -- it cannot count the loads in flight across the unit loop's back edge.  2 us here, and WRONG RESULTS in dw6, where the
	// allocator moved such a register before its wait: a load the compiler does not know about is not safe.  The loads stay visible.)
	typedef int i4 __attribute__((ext_vector_type(4)));
	auto make_srd = [&](const float* base, int nbytes) -> __amdgpu_buffer_rsrc_t {
		return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(base)), 0, nbytes, 0x00020000);
	};
	auto unit_rsrc = [&](int uu) -> __amdgpu_buffer_rsrc_t {
		// (a unit past the end of the range: size 0, every load comes back as zeros; so do the rows past the end of a foot)
		const int ua = FIND_ABL(g.ablate, 512) ? u0 : uu;   // profiling only: every unit reads the range's first rows (served by L2)
		int foot, v0;
		decode(ua, foot, v0);
		const int valid = uu < u1 ? min(32, V - v0) : 0;
		return make_srd(g.a0 + (int64_t)foot * g.a_foot_stride + (int64_t)v0 * lda, valid * lda * 4);
	};
	const int lvoff = lane * 16;
	auto vm_load = [&](u4& dst, const __amdgpu_buffer_rsrc_t& srd, int voff, int soff) { dst = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0); };
#define FIND_VM_WAIT(reg, n) ((void)0)
	auto load_row = [&](const __amdgpu_buffer_rsrc_t& rs, u4 (&slot)[4], int r) { vm_load(slot[r], rs, lvoff, (8 * r + wave) * lda * 4); };
	// Row 8 r + w of a unit goes to LDS as the thread's four values of every plane, one 8-byte word per plane.  The split of its two
	// pairs is cut into pieces (RowSplit::stage) that the k loop places between its MFMAs.
	struct RowSplit {
		f32x2 r[2];        // what is left of the two pairs
		unsigned p[2][3];  // the pieces found so far
		__device__ __forceinline__ void begin(const u4& v) {
			r[0] = f32x2{__uint_as_float(v.x), __uint_as_float(v.y)};
			r[1] = f32x2{__uint_as_float(v.z), __uint_as_float(v.w)};
		}
		// VA: the staged values are the shared product's; the operand is relu(product + the foot's bias)
		__device__ __forceinline__ void begin_virtual(const u4& v, const u4& bias) {
			begin(v);
			r[0] = r[0] + f32x2{__uint_as_float(bias.x), __uint_as_float(bias.y)};
			r[1] = r[1] + f32x2{__uint_as_float(bias.z), __uint_as_float(bias.w)};
			r[0] = f32x2{fmaxf(r[0][0], 0.f), fmaxf(r[0][1], 0.f)};
			r[1] = f32x2{fmaxf(r[1][0], 0.f), fmaxf(r[1][1], 0.f)};
		}
		// piece k of pair h: round what is left to bf16, take it off (4 instructions; the last piece is the rounding alone)
		__device__ __forceinline__ void stage(int h, int k) {
			p[h][k] = __builtin_bit_cast(unsigned, __builtin_convertvector(r[h], bf16x2));
			if (k < 2) r[h] = r[h] - f32x2{__uint_as_float(p[h][k] << 16), __uint_as_float(p[h][k] & 0xffff0000u)};
		}
	};
	const int wbase = wave * G7_ROW + lane * 8;
	auto write_plane = [&](char* buf, const RowSplit& q, int r, int k) {
		*reinterpret_cast<u2*>(buf + wbase + r * (8 * G7_ROW) + k * G7_PLANE) = u2{q.p[0][k], q.p[1][k]};
	};
	// VA: bias row (columns 4 lane .. + 3) of the foot whose unit is being split; VM: (columns col0 + 4 h4 .. + 3) of the foot whose block is being stored
	u4 vab = {0u, 0u, 0u, 0u};
	int vab_foot = -1;
	auto virt_bias = [&](int uu, int col) {   // (wave-uniform branch; a foot changes every tiles_per_foot units)
		if constexpr (VIRT) {
			int foot, v0_unused;
			decode(min(uu, u1 - 1), foot, v0_unused);
			if (foot != vab_foot) {
				const float* bp = VA ? g.va_bias + (int64_t)foot * g.va_bias_stride : g.vm_bias + (int64_t)foot * g.vm_bias_stride;
				vab = __builtin_amdgcn_raw_buffer_load_b128(make_srd(bp, 256 * 4), col * 4, 0, 0);
				vab_foot = foot;
			}
		}
	};
	auto store_row = [&](char* buf, const u4 (&slot)[4], int r) {   // (the prologue's unit: all at once)
		RowSplit q;
		if constexpr (VA) q.begin_virtual(slot[r], vab);
		else q.begin(slot[r]);
#pragma unroll
		for (int k = 0; k < 3; ++k) { q.stage(0, k); q.stage(1, k); }
#pragma unroll
		for (int k = 0; k < 3; ++k) write_plane(buf, q, r, k);
	};
	// MFMA column operand of k-step s, plane p, 16-row block rb: row 16 rb + i16, 8 bf16 at k = 32 s + 8 h4
	const int abase = i16 * G7_ROW + h4 * 16;
	auto frag = [&](const char* buf, int p, int s, int rb) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(buf + abase + p * G7_PLANE + rb * (16 * G7_ROW) + s * 64); };

	// (the bias of a unit is loaded one unit ahead, by hand like the activations: a load the compiler knows about would have it wait for
	// everything in flight before the unit's first MFMA)
	u4 bnext = {0u, 0u, 0u, 0u};
	auto load_bias = [&](int uu) {
		if constexpr (EPI == EPI_BIAS_RELU) {
			const int foot = min(uu, u1 - 1) / upf;
			vm_load(bnext, make_srd(g.bias + (int64_t)foot * g.bias_foot_stride, 256 * 4), (col0 + 4 * h4) * 4, 0);
		}
	};
	{
		load_bias(u0);
		const __amdgpu_buffer_rsrc_t r0 = unit_rsrc(u0);
#pragma unroll
		for (int r = 0; r < 4; ++r) load_row(r0, st[0], r);
		const __amdgpu_buffer_rsrc_t r1 = unit_rsrc(u0 + 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) load_row(r1, st[1], r);
		if constexpr (VA) virt_bias(u0, lane * 4);
#pragma unroll
		for (int r = 0; r < 4; ++r) { FIND_VM_WAIT(st[0][r], 4); store_row(smem, st[0], r); }   // (the four loads of unit u0 + 1 may stay in flight)
		const __amdgpu_buffer_rsrc_t r2 = unit_rsrc(u0 + 2);
#pragma unroll
		for (int r = 0; r < 4; ++r) load_row(r2, st[0], r);
	}

	int cb = 0;
	unsigned long long t_bar = 0, t_loop = 0;
	const unsigned long long t_start = FIND_DBG(g.dbg) ? __builtin_amdgcn_s_memtime() : 0ull;

	// lane (row i16 of block rb, quarter h4) holds columns col0 + 4 h4 .. + 3 of its row in the four registers of acc[rb]
	f32x4 acc[2];
	f32x4 pend[2];   // the finished blocks of the previous unit: stored under the current unit's MFMAs
	auto init_acc = [&]() {
		f32x4 t = {0.f, 0.f, 0.f, 0.f};
		if constexpr (EPI == EPI_BIAS_RELU) {
			FIND_VM_WAIT(bnext, 4);   // (behind it: the loads and stores of the unit that has just ended)
			t = f32x4{__uint_as_float(bnext.x), __uint_as_float(bnext.y), __uint_as_float(bnext.z), __uint_as_float(bnext.w)};
		}
		acc[0] = t; acc[1] = t;
	};
	struct OutTile { __amdgpu_buffer_rsrc_t y, m; };
	auto out_tile = [&](int uq) -> OutTile {
		const int uu = FIND_ABL(g.ablate, 1024) ? u0 : uq;   // profiling only: every unit's block goes to the range's first rows
		int foot, v0;
		decode(uu, foot, v0);
		const int nbytes = min(32, V - v0) * ldy * 4;
		OutTile t;
		if constexpr (FSUM) {
			// (no dZ store: y of this tile = slot 0 / 1 of the foot-sum output, chosen at flush time)
			t.y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(g.fs_out + (int64_t)v0 * ldy)), 0, nbytes, 0x00020000);
		} else {
			t.y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(g.y + (int64_t)foot * g.y_foot_stride + (int64_t)v0 * ldy)), 0, nbytes, 0x00020000);
		}
		t.m = t.y;
		if constexpr (EPI == EPI_MASK) t.m = make_srd(g.mask + (int64_t)foot * g.mask_foot_stride + (int64_t)v0 * ldy, nbytes);
		return t;
	};
	const int ovoff = (i16 * ldy + col0 + 4 * h4) * 4;
	u4 mv[2];
	// FSUM state: the running sum over the feet of the current tile (this wave's 16 columns x 32 rows), the column sums of the unit being
	// finished, and what the unit being finished is (set by unit_body before its k loop)
	f32x4 fs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
	float cst[4] = {0.f, 0.f, 0.f, 0.f};
	int fs_first_foot = 0;     // foot the current run of this tile began with (0: slot 0)
	int prev_foot = 0, prev_v0 = 0;   // foot / first row of the unit whose blocks store_block is finishing
	bool prev_ends_run = false;
	float* const cs_lds = reinterpret_cast<float*>(smem + GEMM7_LDS);   // [foot][128]: this workgroup's columns
	auto store_block = [&](const OutTile& t, int rb) {
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			v[e] = pend[rb][e];
			if constexpr (EPI == EPI_BIAS_RELU) v[e] = fmaxf(v[e], 0.f);
			if constexpr (VM) v[e] = (__uint_as_float(mv[rb][e]) + __uint_as_float(vab[e]) > 0.f) ? v[e] : 0.f;
			else if constexpr (EPI == EPI_MASK) v[e] = (__uint_as_float(mv[rb][e]) > 0.f) ? v[e] : 0.f;
		}
		if constexpr (FSUM) {
			// (rows past the end of the template: their dZ rows were loaded as zeros, so v is zero there whatever the mask says)
#pragma unroll
			for (int e = 0; e < 4; ++e) { fs[rb][e] += v[e]; cst[e] = (rb == 0) ? v[e] : cst[e] + v[e]; }
			if (rb == 1) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const float c = row16_sum(cst[e]);   // the unit's 32 rows of column col0 + 4 h4 + e
					if (i16 == 0) atomicAdd(&cs_lds[prev_foot * 128 + wave * 16 + 4 * h4 + e], c);   // (ds_add_f32; this wave alone owns these columns: unit order)
				}
				if (prev_ends_run) {
					// (t.y: the tile's rows of slot 0, bounded at the template's last row; slot 1 lies g.fs_slot_stride floats further)
					const __amdgpu_buffer_rsrc_t s1 = make_srd(g.fs_out + g.fs_slot_stride + (int64_t)prev_v0 * ldy, min(32, V - prev_v0) * ldy * 4);
					const bool whole = fs_first_foot == 0 && prev_foot == nf - 1;   // the whole tile in one run: nobody else writes its slot 1
#pragma unroll
					for (int q = 0; q < 2; ++q) {
						const u4 sum = u4{__float_as_uint(fs[q][0]), __float_as_uint(fs[q][1]), __float_as_uint(fs[q][2]), __float_as_uint(fs[q][3])};
						if (fs_first_foot == 0) store_b128(sum, t.y, ovoff, q * 16 * ldy * 4);
						else store_b128(sum, s1, ovoff, q * 16 * ldy * 4);
						if (whole) store_b128(u4{0u, 0u, 0u, 0u}, s1, ovoff, q * 16 * ldy * 4);
						fs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
					}
					fs_first_foot = 0;   // (what follows a finished tile starts the next one; a finished range is followed by nothing)
				}
			}
		} else {
			store_b128(u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, t.y, ovoff, rb * 16 * ldy * 4);   // (no SGPR offset: common.h)
		}
	};
	auto mm = [&](const bf16x8& w, const bf16x8& x, f32x4& c) { c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0); };

	// Unit u: multiply it (buffer cb).  Under its 96 MFMAs: split and store unit u + 1 from `slot` (one PAIR of values per k-step: nine
	// VALU instructions against twelve MFMAs), refill `slot` with unit u + 3, store the previous unit's blocks (`first`: there is none).
	auto unit_body = [&](int u, u4 (&slot)[4], bool first) {
		const unsigned long long tb0 = FIND_DBG(g.dbg) ? __builtin_amdgcn_s_memtime() : 0ull;
		lds_barrier();   // unit u is complete in buffer cb; nobody reads buffer cb ^ 1 (unit u - 1) any more (the loads in flight stay in flight)
		const unsigned long long tb1 = FIND_DBG(g.dbg) ? __builtin_amdgcn_s_memtime() : 0ull;
		const char* buf = smem + cb * G7_BUF;
		char* other = smem + (cb ^ 1) * G7_BUF;
		const __amdgpu_buffer_rsrc_t rs2 = unit_rsrc(u + 3);
		const OutTile prev = out_tile(first ? u : u - 1);
		if constexpr (FSUM) {
			if (!first) {
				decode(u - 1, prev_foot, prev_v0);
				prev_ends_run = prev_foot == nf - 1;   // (the range's last unit is finished below the loop)
			}
		}
		pend[0] = acc[0]; pend[1] = acc[1];
		init_acc();
		load_bias(u + 1);
		if constexpr (VA) virt_bias(u + 1, lane * 4);                      // the unit split under this one's MFMAs
		if constexpr (VM) virt_bias(first ? u : u - 1, col0 + 4 * h4);     // the unit whose blocks are stored under them
		if constexpr (EPI == EPI_MASK) {
			vm_load(mv[0], prev.m, ovoff, 0);
			vm_load(mv[1], prev.m, ovoff, 16 * ldy * 4);
		}
		bf16x8 a1[2][2], a2[2][2], a3[2][2];   // [buffer][row block]
#pragma unroll
		for (int rb = 0; rb < 2; ++rb) { a1[0][rb] = frag(buf, 0, 0, rb); a2[0][rb] = frag(buf, 1, 0, rb); a3[0][rb] = frag(buf, 2, 0, rb); }
		RowSplit q;
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const int cu = s & 1, nx = cu ^ 1, row = s >> 1, h = s & 1;
			// The further a wave is into its unit, the lower its priority: the two waves of a SIMD then SHARE the matrix pipe.  (The
			// arbiter prefers the older wave: without this, wave w ran its k loop at full speed and waited at the barrier -- a quarter of the
			// kernel's time -- while wave w + 4 ran the rest of its own alone, with every stall of a lone wave exposed.)
			if (h == 0 && !FIND_ABL(g.ablate, 16)) {   // (the switch exists in the laboratory build only)
				if (row == 0) __builtin_amdgcn_s_setprio(3);
				else if (row == 1) __builtin_amdgcn_s_setprio(2);
				else if (row == 2) __builtin_amdgcn_s_setprio(1);
				else __builtin_amdgcn_s_setprio(0);
			}
			if (s + 1 < 8) {
#pragma unroll
				for (int rb = 0; rb < 2; ++rb) {
					if constexpr (!(ABL & 2)) { a1[nx][rb] = frag(buf, 0, s + 1, rb); a2[nx][rb] = frag(buf, 1, s + 1, rb); a3[nx][rb] = frag(buf, 2, s + 1, rb); }
					else { a1[nx][rb] = a1[cu][rb]; a2[nx][rb] = a2[cu][rb]; a3[nx][rb] = a3[cu][rb]; }
				}
			}
			// VMEM operations behind the load of slot[row] when it is needed (buffer loads and stores complete in order): the rest of its own
			// unit's, those of the unit after it and this unit's so far -- at least 7 in every unit of the pipeline (10-11 in the steady
			// state: two units of loads stay in flight)
			if (h == 0 && !(ABL & 1)) {
				FIND_VM_WAIT(slot[row], 7);
				if constexpr (VA) q.begin_virtual(slot[row], vab);
				else q.begin(slot[row]);
			}
			// smallest terms first
			mm(B1[s], a3[cu][0], acc[0]); mm(B1[s], a3[cu][1], acc[1]);
			if constexpr (!(ABL & 1)) q.stage(h, 0);
			__builtin_amdgcn_sched_barrier(0);
			mm(B3[s], a1[cu][0], acc[0]); mm(B3[s], a1[cu][1], acc[1]);
			__builtin_amdgcn_sched_barrier(0);
			mm(B2[s], a2[cu][0], acc[0]); mm(B2[s], a2[cu][1], acc[1]);
			if constexpr (!(ABL & 1)) q.stage(h, 1);
			__builtin_amdgcn_sched_barrier(0);
			mm(B1[s], a2[cu][0], acc[0]); mm(B1[s], a2[cu][1], acc[1]);
			if (h == 1 && !first && s >= 4 && !(ABL & 8)) {
				if constexpr (EPI == EPI_MASK) { if (s == 5) FIND_VM_WAIT(mv[0], 2); else FIND_VM_WAIT(mv[1], 3); }   // (behind it: the other mask load, this unit's loads and store so far)
				store_block(prev, (s - 4) >> 1);
			}
			__builtin_amdgcn_sched_barrier(0);
			mm(B2[s], a1[cu][0], acc[0]); mm(B2[s], a1[cu][1], acc[1]);
			if constexpr (!(ABL & 1)) q.stage(h, 2);
			__builtin_amdgcn_sched_barrier(0);
			mm(B1[s], a1[cu][0], acc[0]); mm(B1[s], a1[cu][1], acc[1]);
			if (h == 1) {
				if constexpr (!(ABL & 1)) { write_plane(other, q, row, 0); write_plane(other, q, row, 1); write_plane(other, q, row, 2); }
				if constexpr (!(ABL & 4)) load_row(rs2, slot, row);   // unit u + 3: on its way for two units
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if (FIND_DBG(g.dbg)) {   // profiling only (tools/prof_x3.py, FIND_DBG): ticks of wave 0 at the barrier / in the k loop
			const unsigned long long te1 = __builtin_amdgcn_s_memtime();
			t_bar += tb1 - tb0; t_loop += te1 - tb1;
		}
		cb ^= 1;
	};
	if constexpr (FSUM) {
		int f0, v0_unused;
		decode(u0, f0, v0_unused);
		fs_first_foot = f0;
		for (int k = lane; k < nf * 16; k += 64) cs_lds[(k >> 4) * 128 + wave * 16 + (k & 15)] = 0.f;   // (this wave's columns; its LDS operations run in order)
	}
	unit_body(u0, st[1], true);
	for (int u = u0 + 1; u < u1; u += 2) {
		unit_body(u, st[0], false);
		if (u + 1 < u1) unit_body(u + 1, st[1], false);
	}
	{   // the last unit's blocks
		const OutTile last = out_tile(u1 - 1);
		if constexpr (FSUM) { decode(u1 - 1, prev_foot, prev_v0); prev_ends_run = true; }
		pend[0] = acc[0]; pend[1] = acc[1];
		if constexpr (VM) virt_bias(u1 - 1, col0 + 4 * h4);
		if constexpr (EPI == EPI_MASK) {
			vm_load(mv[0], last.m, ovoff, 0);
			vm_load(mv[1], last.m, ovoff, 16 * ldy * 4);
			FIND_VM_WAIT(mv[0], 0); FIND_VM_WAIT(mv[1], 0);
		}
		store_block(last, 0); store_block(last, 1);
	}
	if constexpr (FSUM) {   // the per-foot column sums of this range: cs_out [range = pair][foot][256]
		float* out = g.cs_out + (int64_t)pair * nf * 256 + ((b >> 3) & 1) * 128 + wave * 16;
		for (int k = lane; k < nf * 16; k += 64) out[(k >> 4) * 256 + (k & 15)] = cs_lds[(k >> 4) * 128 + wave * 16 + (k & 15)];
	}
	if (FIND_DBG(g.dbg) && tid == 0) {
		g.dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - t_start;
		g.dbg[blockIdx.x * 4 + 1] = t_bar; g.dbg[blockIdx.x * 4 + 2] = 0; g.dbg[blockIdx.x * 4 + 3] = t_loop;
	}
}

}  // namespace mlp
}  // namespace find
