// C-ABI: find_mlp_fwd / find_mlp_bwd  -- launch sequences over the kernels in mlp_kernels.h.
// Replaces NeuralDisplacementField.forward (reference src/model/model.py:393-453) and its autograd backward.
#include "mlp_dw2.h"
#include "mlp_dw4.h"
#include "mlp_dwpe.h"
#include "mlp_dwpe6.h"
#include "mlp_gemm5.h"
#include "mlp_gemm6.h"
#include "mlp_gemm7.h"
#include "mlp_dw3.h"
#include "mlp_dw6.h"
#include "mlp_gemm4.h"
#include "mlp_fused.h"
#include "mlp_fused6.h"

namespace find {
namespace mlp {

struct Dims {
	int64_t pos_batch, n_feet, V;
	bool shared;      // one set of positions evaluated for every foot: trunk computed once
	int64_t rows_t;   // trunk rows
	int64_t rows_h;   // head rows
	int64_t feet_t;   // grid.y of trunk launches
	int nchunk0;      // K chunks of trunk layer 0
	int nkt0;         // 256-wide k tiles of trunk layer 0 (dW)
};

static int make_dims(const find_mlp_params* p, int64_t pos_batch, int64_t n_feet, int64_t V, Dims* d) {
	FIND_REQUIRE(p != nullptr, "find_mlp: params is NULL");
	FIND_REQUIRE(p->width == W, "find_mlp: only width=256 is supported (got %d)", p->width);
	FIND_REQUIRE(p->in_dim == 3, "find_mlp: only in_dim=3 is supported (got %d)", p->in_dim);
	FIND_REQUIRE(p->pe_size >= 0 && p->pe_size <= 256 && p->pe_size % 32 == 0, "find_mlp: pe_size must be a multiple of 32 in [0,256] (got %d)", p->pe_size);
	FIND_REQUIRE(p->n_trunk >= 1 && p->n_trunk <= FIND_MAX_LAYERS, "find_mlp: n_trunk out of range (%d)", p->n_trunk);
	FIND_REQUIRE(p->n_disp >= 1 && p->n_disp < FIND_MAX_LAYERS, "find_mlp: n_disp out of range (%d)", p->n_disp);
	FIND_REQUIRE(p->n_col >= 1 && p->n_col < FIND_MAX_LAYERS, "find_mlp: n_col out of range (%d)", p->n_col);
	FIND_REQUIRE(p->lat_disp >= 0 && p->lat_col >= 0, "find_mlp: negative latent size");
	FIND_REQUIRE(n_feet >= 1 && V >= 1, "find_mlp: empty batch (n_feet=%lld, n_pts=%lld)", (long long)n_feet, (long long)V);
	FIND_REQUIRE(pos_batch == 1 || pos_batch == n_feet, "find_mlp: pos_batch must be 1 or n_feet (got %lld vs %lld)", (long long)pos_batch, (long long)n_feet);
	FIND_REQUIRE(V < (1ll << 31) / 256 && n_feet < (1 << 16), "find_mlp: batch too large for the launch grid");
	d->pos_batch = pos_batch; d->n_feet = n_feet; d->V = V;
	d->shared = (pos_batch == 1 && n_feet > 1);
	d->rows_t = pos_batch * V;
	d->rows_h = n_feet * V;
	d->feet_t = pos_batch;
	const int nsc = p->pe_size >> 4;
	d->nchunk0 = nsc + 1;
	d->nkt0 = (int)cdiv((nsc + 1) * 32, 256);
	return FIND_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int check_weights(const find_mlp_params* p) {
	FIND_REQUIRE(p->pe_size == 0 || p->B != nullptr, "find_mlp: B is NULL");
	for (int i = 0; i < p->n_trunk; ++i) {
		FIND_REQUIRE(p->trunk_w[i] && p->trunk_b[i], "find_mlp: trunk layer %d weight/bias NULL", i);
		if (i > 0) FIND_REQUIRE(aligned16(p->trunk_w[i]), "find_mlp: trunk_w[%d] must be 16-byte aligned", i);
	}
	for (int i = 0; i <= p->n_disp; ++i) {
		FIND_REQUIRE(p->disp_w[i] && p->disp_b[i], "find_mlp: disp layer %d weight/bias NULL", i);
		if (i > 0) FIND_REQUIRE(aligned16(p->disp_w[i]), "find_mlp: disp_w[%d] must be 16-byte aligned", i);
	}
	for (int i = 0; i <= p->n_col; ++i) {
		FIND_REQUIRE(p->col_w[i] && p->col_b[i], "find_mlp: col layer %d weight/bias NULL", i);
		if (i > 0) FIND_REQUIRE(aligned16(p->col_w[i]), "find_mlp: col_w[%d] must be 16-byte aligned", i);
	}
	return FIND_OK;
}

// Forward workspace.  With save=false the hidden activations ping-pong between two buffers per stage.
// 16-k steps of the weight stream a chain of this model can need at most (forward: Fourier layer + trunk + both heads; backward: both
// heads + the two-operand trunk-output product + trunk), and the bytes of its fragment-ordered planes (mlp_fused6.h)
static int64_t chain_w6_bytes(const find_mlp_params* p) {
	const int64_t steps = 2 * (KP0 / KC + 1) + 16 * (int64_t)(p->n_trunk + p->n_disp + p->n_col + 4);
	return 8 * steps * F6_STEP_BYTES;
}

struct FwdWs {
	float* w0p;   // (256, KP0) layer-0 weight in padded PE order
	float* wd0;   // (256,256) main block of mlp_disp.0.weight
	float* wc0;   // (256,256) main block of mlp_col.0.weight
	float* fbd;   // (n_feet,256) per-foot bias of the disp head's first layer
	float* fbc;
	float* H[FIND_MAX_LAYERS];
	float* D[FIND_MAX_LAYERS];
	float* C[FIND_MAX_LAYERS];
	float* zd;    // (rows_h,3)
	float* zc;
	float* hp;    // shared template: (V,256) product of the trunk output with a head's first-layer weight (no bias), reused per head
	float* hp2;
	void* w6;     // fused6_kernel: the chain's weights as fragment-ordered bf16 planes (chain_w6_bytes)
	int64_t bytes;
};

static void carve_fwd(const find_mlp_params* p, const Dims& d, bool save, void* ws, FwdWs* o) {
	Carver c(ws);
	o->w0p = c.take<float>((int64_t)W * KP0);
	o->wd0 = c.take<float>((int64_t)W * W);
	o->wc0 = c.take<float>((int64_t)W * W);
	o->fbd = c.take<float>(d.n_feet * W);
	o->fbc = c.take<float>(d.n_feet * W);
	o->zd = c.take<float>(d.rows_h * 3);
	o->zc = c.take<float>(d.rows_h * 3);
	o->hp = (d.shared && d.n_feet > 1) ? c.take<float>(d.V * W) : nullptr;
	o->hp2 = (d.shared && d.n_feet > 1) ? c.take<float>(d.V * W) : nullptr;   // the colour head's copy: the heads run on two streams
	o->w6 = c.take<char>(chain_w6_bytes(p));
	float* pt[2] = {nullptr, nullptr};
	float* pd[2] = {nullptr, nullptr};
	float* pc[2] = {nullptr, nullptr};
	for (int i = 0; i < p->n_trunk; ++i) {
		if (save || !pt[i & 1]) pt[i & 1] = c.take<float>(d.rows_t * W);
		o->H[i] = pt[i & 1];
	}
	for (int i = 0; i < p->n_disp; ++i) {
		if (save || !pd[i & 1]) pd[i & 1] = c.take<float>(d.rows_h * W);
		o->D[i] = pd[i & 1];
	}
	for (int i = 0; i < p->n_col; ++i) {
		if (save || !pc[i & 1]) pc[i & 1] = c.take<float>(d.rows_h * W);
		o->C[i] = pc[i & 1];
	}
	o->bytes = c.off;
}

}  // namespace mlp
}  // namespace find

// Per-device state of the MLP entry points (find_hip.h: find_ctx_create).  Nothing below is process-global.
enum { K_GEMM2_PE = 0, K_GEMM3_RELU, K_GEMM3_MASK, K_GEMM3_NONE, K_GEMM4_4_RELU, K_GEMM4_4_MASK, K_GEMM4_4_NONE, K_GEMM4_2_RELU, K_GEMM4_2_MASK,
	   K_GEMM4_2_NONE, K_GEMM5_RELU, K_GEMM5_MASK, K_GEMM5_NONE, K_GEMM6_RELU, K_GEMM6_MASK, K_GEMM6_NONE, K_GEMM7_RELU, K_GEMM7_MASK, K_GEMM7_NONE, K_DW2, K_DW3, K_DW6, K_DW6G, K_FUSED, K_FUSED2, K_FUSED6, K_FUSED6_2, K_DW2G, K_REDUCE, K_DW2_REPRO, K_GEMM5_RELU_H, K_GEMM5_MASK_H, K_DW3_H, K_GEMM5_RELU_V, K_GEMM5_MASK_V, K_DW3_V, K_GEMM7_RELU_V, K_GEMM7_MASK_V, K_DW6_V, K_GEMM7_MASK_VF, K_COUNT };
constexpr int N_SIDE = 4;       // internal streams: 0 = q (large head layers' dW), 1 / 2 = first head layers + trunk layers, 3 = slab reduces
constexpr int N_EVENTS = 512;   // event ring: an MLP call with 3 x 8 layers uses ~170; checked per call

struct find_ctx {
	int device = 0;
	int num_cus = 256;
	int lds_bytes = 160 * 1024;   // largest dynamic LDS one workgroup may ask for on this device
	// knobs (find_hip.h: find_ctx_set)
	int ablate = 0;               // switches; the product accepts MLP_SWITCHES only (common.h), the diagnostics build every bit
	int x3_abl = 0;               // diagnostics build: ablation variant of gemm6 / gemm7 (tools/ablate_x3.py)
	unsigned long long* dbg = nullptr;
	unsigned long long* dw2_verify = nullptr;   // diagnosis (tools/probe_lds_fault.py): log buffer of dw2_kernel's stage verification
	int64_t gemm4_min_units = 1024;
	int gemm4_small = 64;         // column-quarter gemm4 for launches of at least this many 32-row units (0: never)
	int dw_pe_target = 256;       // workgroups of the Fourier layer's weight-gradient launch (one round over the chip)
	int dw_pe_lds_free = 1;       // Fourier layer's weight gradient: 1 = dwpe_kernel (no LDS, one frequency per lane), 0 = dw_kernel<AMODE_PE> (round 1, LDS-staged)
	int dw2_min_cps = 8;          // at least this many 16-row chunks per dw2 workgroup (4: 2.257, 8: 2.243, 12: 2.266 ms/step at C2)
	int bwd_streams = 1;          // weight gradients on the side streams
	int fwd_streams = 1;          // colour head on a side stream beside the displacement head
	int reduce_stream = 0;        // 1 = slab reduces of the large head layers on their own stream R (two alternating slab sets): what the LDS-ring weight
	                              // gradient needed (its reduce only got a CU when a ring workgroup retired); with dw4_kernel the reduce behind its
	                              // launch on Q is 0.6 - 0.9 % faster (train_3d 3.245 -> 3.225 ms, C2 2.220 -> 2.199), so off by default
	int gemm5_min_units = 1024;
	int gemm6_min_units = 1024;
	int gemm7 = 1;                // bf16x3 Linear kernel: 1 = gemm7 (W in registers, activations through LDS), 0 = gemm6 (W planes in LDS; kept for A/B)
	int dwpe6 = 1;                // knob: the Fourier layer's weight gradient of bf16x3 calls on dwpe6_kernel (0: dwpe_kernel, fp32 MFMA)
	int dw6_group = 1;            // knob: grouped weight gradients of bf16x3 calls on dw6_group_kernel (0: dw4_group, fp32 MFMA)
	int direct_w = 1;             // knob: bf16x3 kernels read the model's weights themselves (transposed / Fourier order) instead of repacked copies (0: A/B)
	int mlp_f16 = 0;              // default precision of calls that do not name one
	int lds_exclusive = 0;        // 1 = the LDS-DMA ring kernels reserve the whole LDS of their CU: round 1's containment of the co-residence fault, which
	                              // round 2 showed to be about registers, not LDS (see "Co-residence" below); off by default now
	int reduce_exclusive = 0;     // diagnosis only: 1 = the slab reduce (16 KB of LDS) reserves its CU's whole LDS; 2 = LDS-free, slow reduce: the stress
	                              // configuration for the co-residence fault (long-lived foreign waves beside the weight-gradient kernels)
	int dw_lds_free = 1;          // 256 x 256 weight gradients: 1 = dw4_kernel (no LDS, <= 256 registers), 0 = dw2_kernel (LDS-DMA ring, whole register file claimed);
	                              // reproducers of the co-residence fault: 2 = dw4_wide_kernel (no LDS, 328 registers), 3 = dw2_repro_kernel (312 registers)
	int group_spf = 0;            // grouped weight gradients: splits per foot (0 = cost model of group_geometry)
	int fused_max_units = 512;    // chains of layers over at most this many 32-row tiles run as ONE fused_chain_kernel launch (0: never)
	int dw6_wgs = 0;              // knob: workgroups (= slabs) of a dw6 launch; 0 = one per CU, half that beside the dX chain (weight_grad)
	int fused6 = 1;               // knob: bf16x3 calls run their chains on fused6_kernel (0: the fp32-MFMA chain, as the other precisions)
	// internal streams / events
	hipStream_t side[N_SIDE] = {nullptr, nullptr, nullptr, nullptr};
	bool side_bound = false;      // the side streams have been chosen against the hardware queue of a caller's stream (bind_side_streams)
	int bind_streams = 1;         // knob: 0 = keep the side streams as created
	int r_queue = 2;              // knob: the side stream (0 = Q, 1 = T1, 2 = T2) whose hardware queue the slab-reduce stream R shares
	int cu_reserve = 0;           // knob: CUs per XCD the side streams may NOT use (hipExtStreamCreateWithCUMask): the short kernels on the caller's
	                              // stream -- the loss-side chain the main backward waits for -- then always find a free CU beside the side streams'
	                              // long weight-gradient workgroups.  Takes effect when the side streams are bound (first fork).  Masked streams are
	                              // BLOCKING streams (HIP offers no other kind with a mask): only for callers on a non-default stream
	int side_cus = 256;           // CUs a side stream may use (num_cus - 8 cu_reserve once bound): what the weight-gradient launches are sized for
	hipEvent_t ev[N_EVENTS];
	int n_events = 0;
	int next = 0;
	int events_per_call_max = 0;
	bool attr_done[K_COUNT] = {};
	// how the last forward calls that saved a workspace stored the heads' activations (act16): the backward of a workspace follows its
	// forward's decision even if a knob was turned in between (ring of the last 16; a workspace not found falls back to the rule)
	struct Act16Note { const void* ws; bool a16; bool fold; };
	Act16Note act16_notes[16] = {};
	int act16_next = 0;
	int pe_on_t2 = 1;             // knob: the Fourier layer's weight gradient of a shared-template backward runs on T2 instead of behind dw6 on Q
	int footsum_fold = 1;         // knob: the foot sums of a shared template's first-layer dZ are formed inside the dX GEMM that produces it (mlp_gemm7.h FSUM)
	int group_head0 = 1;          // knob: a shared template's first head layers' weight gradients ride in the trunk's grouped launch (mlp_bwd_body)
	int bcast_fold = 1;           // knob: inside act16 the broadcast first head layer's output is formed by its readers instead of stored (use_fold)
	int act16 = 1;                // knob: in the opt-in fp16 mode the heads' hidden activations and their gradients are STORED as fp16 at the large
	                              // shared-template shapes (use_act16): those layers are HBM-bound, and the matrix pipe rounds them to fp16 anyway
	int defer_join = 0;           // knob, read by the next find_mlp_bwd: leave the weight-gradient side streams running behind the call (find_hip.h)
	hipEvent_t pend_ev[N_SIDE] = {nullptr, nullptr, nullptr, nullptr};   // end of the deferred work on each side stream
	bool pend[N_SIDE] = {};       // side stream k carries deferred work nobody has waited for yet
	// set per call
	bool f16 = false;
	bool x3 = false;              // this call runs its 256 -> 256 layers as bf16x3 (fp32-faithful on the bf16 matrix pipe, mlp_gemm6.h)
};

namespace find {
namespace mlp {

#define FIND_HIP_OK(expr, what)                                                                  \
	do {                                                                                         \
		hipError_t _e = (expr);                                                                  \
		if (_e != hipSuccess) {                                                                  \
			set_error("%s: %s", what, hipGetErrorString(_e));                                    \
			return FIND_ELAUNCH;                                                                 \
		}                                                                                        \
	} while (0)

#define FIND_TRY(expr)                    \
	do {                                  \
		const int _r = (expr);            \
		if (_r != FIND_OK) return _r;     \
	} while (0)

// Co-residence fault: what it was.
// Round 1: with a second workgroup of another stream resident on the same CU, dw2_kernel (weight gradient, both MFMA operands through
// an LDS-DMA ring, one wave per SIMD) produced rare wrong partial tiles -- a rank-1 error of ~1 % in a handful of dW elements, 3 % of the
// backward passes at 4 x 1002 rows, 13 % at 16 x 6890, every pass when the dX GEMMs ran on gemm3 beside it -- with every vmcnt / barrier of
// the ring in place.  Launching the LDS-DMA ring kernels with the WHOLE LDS of their CU made it disappear (0 of 500 passes) and was taken
// for the cure: "an LDS-using neighbour disturbs the ring".  It was a coincidence of which neighbours it kept out.
// Round 2 (tools/check_determinism.py with the knobs named; numbers = wrong tensors per 150 passes of a 16 x 6890 backward):
//   * a stress configuration reproduces it in EVERY pass: the slab reduce replaced by an LDS-free, slow one ("reduce_exclusive" = 2),
//     so that reduces of earlier layers stay resident beside the weight-gradient kernels of later ones: ~850 -- with the LDS reservation
//     on as well as off (it cannot keep an LDS-free kernel out).  The layers that break are exactly those whose weight-gradient launch
//     overlaps a running reduce; on one stream ("bwd_streams" = 0): 0.
//   * not the slabs (a private slab set per weight gradient: same rate); not the ring (every stage of every chunk equals HBM when it is
//     published AND after the wave has consumed it, 2.35 M stages per run); not barrier timing, DMA in flight, M0 hazards, operand-register
//     reuse, barrier flavour (each padded / changed: same rate).
//   * not LDS at all: dw4_wide_kernel -- no LDS, no DMA, no barrier, dw2's tile shape read straight from global memory -- breaks the
//     same way (663), while dw4_kernel, the same code with half the tile per wave, never does (0 in 1450 passes).
//   * what the victims share is their REGISTER SHAPE: dw2 312, dw2_group 300, dw4_wide 328 registers per lane -- all 256 accumulator
//     registers (the whole AGPR set) behind fewer than 256 architectural ones, one wave per SIMD.  Every kernel with at most 256 registers
//     was clean; so was the masked gemm3 when it still took 328 (200 architectural + 128 accumulator registers: 0 in 160 stress passes,
//     rebuilt with -DFIND_GEMM3_MIN_WGS=1), so the trigger is narrower than "more than 256".  And dw2 UNCHANGED except for its allocation
//     padded to all 512 registers of the SIMD (FIND_CLAIM_WHOLE_REGISTER_FILE: no foreign wave fits beside it any more): 0 in 600
//     passes, LDS reservation off.
// So: a wave that owns the full accumulator set within an allocation of fewer than 512 registers gets wrong register contents when
// waves of another kernel are allocated on its SIMD.
// The kernel descriptors are right (dw2: granulated VGPR count 38 = 312 registers, accum_offset 56); whether the silicon, the firmware's
// wave save / restore or the runtime mishandles such waves cannot be told from inside a kernel (a stand-alone two-kernel program with a
// 292-register victim ran clean: something else of the step's setting takes part).  The rule adopted, a superset of every shape that
// broke and enforced by tests/test_host_api.py on the compiler's output: a kernel either fits in 256 registers or claims the whole file.  dw4_kernel (<= 256,
// two waves per SIMD, no LDS) is the default weight gradient; dw2 / dw2_group / dw3 claim the file; gemm3 is capped at 256 through its
// launch bounds; dw2_repro_kernel and dw4_wide_kernel stay as the reproducers ("dw_lds_free" = 3 / 2).  The whole-LDS reservation is
// off by default ("lds_exclusive"): the stress runs are clean without it, and what it really did was keep most neighbours away.
template <typename K>
static int prepare_kernel(find_ctx* c, int id, K kernel, int need_bytes, int* launch_bytes, bool reserve = true) {
	const int want = (reserve && c->lds_exclusive) ? c->lds_bytes : need_bytes;
	if (need_bytes > c->lds_bytes) {
		set_error("find_mlp: kernel needs %d bytes of LDS, device %d grants %d per workgroup", need_bytes, c->device, c->lds_bytes);
		return FIND_EINVAL;
	}
	if (!c->attr_done[id]) {
		FIND_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes),
					"hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
		c->attr_done[id] = true;
	}
	*launch_bytes = want;
	return FIND_OK;
}

// Fork / join of one entry-point call onto the context's side streams.  fork_to(k) makes side stream k wait for everything issued on
// the caller's stream so far; chain(a, b) orders side stream b behind a; join() -- called on EVERY exit path after the first fork,
// error returns included -- makes the caller's stream wait for every side stream this call touched, so that when the call returns
// the caller may free or reuse any buffer it passed in (stream-ordered).  Every HIP return code is kept: the first failure is
// reported by join().  Works under stream capture: a captured call forks and joins the same streams, so the capture stays closed.
static int bind_side_streams(find_ctx* c, hipStream_t caller);   // (below, with the probe)

struct Fork {
	find_ctx* c;
	hipStream_t s;
	bool on;              // side streams in use for this call
	bool capturing = false;   // the caller's stream is being captured into a HIP graph
	bool used[N_SIDE] = {};
	int n_ev = 0;
	int rc = FIND_OK;

	Fork(find_ctx* ctx, hipStream_t caller, bool enable) : c(ctx), s(caller), on(enable && ctx->side[0] != nullptr) {
		hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
		if (hipStreamIsCapturing(caller, &st) == hipSuccess) capturing = st != hipStreamCaptureStatusNone;
		if (on && !capturing && !ctx->side_bound && ctx->bind_streams) (void)bind_side_streams(ctx, caller);   // (a failed probe keeps the streams as created)
	}
	hipStream_t stream(int k) const { return on ? c->side[k] : s; }
	void fail(hipError_t e, const char* what) {
		if (e != hipSuccess && rc == FIND_OK) {
			set_error("%s: %s", what, hipGetErrorString(e));
			rc = FIND_ELAUNCH;
		}
	}
	hipEvent_t event() {
		hipEvent_t e = c->ev[c->next];
		c->next = (c->next + 1) % N_EVENTS;
		if (++n_ev > N_EVENTS && rc == FIND_OK) {
			set_error("find_mlp: more than %d events in one call", N_EVENTS);
			rc = FIND_ELAUNCH;
		}
		return e;
	}
	void order(hipStream_t from, hipStream_t to) {
		hipEvent_t e = event();
		fail(hipEventRecord(e, from), "hipEventRecord");
		fail(hipStreamWaitEvent(to, e, 0), "hipStreamWaitEvent");
	}
	void fork_to(int k) {
		if (!on) return;
		order(s, c->side[k]);
		used[k] = true;
	}
	// an event that fires when side stream k has run what was issued so far
	hipEvent_t mark(int k) {
		if (!on) return nullptr;
		hipEvent_t e = event();
		fail(hipEventRecord(e, c->side[k]), "hipEventRecord");
		return e;
	}
	void wait(int k, hipEvent_t e) {
		if (on && e) fail(hipStreamWaitEvent(c->side[k], e, 0), "hipStreamWaitEvent");
	}
	void chain(int from, int to) {
		if (!on || from == to) return;
		// a stream enters the call through a fork from the CALLER's stream first, never only through another side stream: under
		// stream capture, hipStreamEndCapture (ROCm 7.0 / 7.2) faults on a capture whose parallel stream was pulled in by a stream
		// that is itself a fork (nested fork); eagerly the extra wait is implied by the one on `from`
		if (!used[to]) fork_to(to);
		order(c->side[from], c->side[to]);
	}
	int join() {
		// (also what an earlier call left running there: find_ctx.pend -- streams are FIFO, waiting for this call's end covers it)
		for (int k = 0; k < N_SIDE; ++k)
			if ((on && used[k]) || (c->pend[k] && !capturing)) { order(c->side[k], s); used[k] = false; c->pend[k] = false; }
		c->events_per_call_max = std::max(c->events_per_call_max, n_ev);
		return rc;
	}
	// instead of join(): the side streams this call touched keep running behind it; whoever needs their results waits for pend_ev
	// (find_ctx_join, or the join() of a later call).  Only the caller may know that nothing reads them before that.
	int defer() {
		if (on)
			for (int k = 0; k < N_SIDE; ++k)
				if (used[k]) { fail(hipEventRecord(c->pend_ev[k], c->side[k]), "hipEventRecord"); c->pend[k] = true; used[k] = false; }
		c->events_per_call_max = std::max(c->events_per_call_max, n_ev);
		return rc;
	}
};

static int launch_gemm2_pe(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {
	constexpr int BM = 64;
	int lds = 0;
	const int rc = prepare_kernel(c, K_GEMM2_PE, &gemm2_kernel<BM, AMODE_PE, EPI_BIAS_RELU>, gemm2_lds_bytes<BM>(), &lds);
	if (rc != FIND_OK) return rc;
	a.tiles_per_foot = (int)cdiv(a.V, BM);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	const int grid = std::min(a.ntiles, c->num_cus);
	hipLaunchKernelGGL((gemm2_kernel<BM, AMODE_PE, EPI_BIAS_RELU>), dim3(grid), dim3(256), lds, s, a);
	return FIND_OK;
}

template <int BM, int EPI>
static int launch_gemm3_t(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {
	int lds = 0;
	const int rc = prepare_kernel(c, K_GEMM3_RELU + (EPI == EPI_BIAS_RELU ? 0 : EPI == EPI_MASK ? 1 : 2), &gemm3_kernel<BM, EPI>, gemm2_lds_bytes<BM>(), &lds);
	if (rc != FIND_OK) return rc;
	a.tiles_per_foot = (int)cdiv(a.V, BM);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	const int grid = std::min(a.ntiles, c->num_cus);
	hipLaunchKernelGGL((gemm3_kernel<BM, EPI>), dim3(grid), dim3(256), lds, s, a);
	return FIND_OK;
}

template <int EPI, int NI, int NW = 8>
static int launch_gemm4_t(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {
	int lds = 0;
	const int id = (NI == 4 ? K_GEMM4_4_RELU : K_GEMM4_2_RELU) + (EPI == EPI_BIAS_RELU ? 0 : EPI == EPI_MASK ? 1 : 2);
	const int rc = prepare_kernel(c, id, &gemm4_kernel<EPI, NI, NW>, NI * 32 * 1024, &lds);
	if (rc != FIND_OK) return rc;
	a.tiles_per_foot = (int)cdiv(a.V, 32);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	constexpr int G = 8 * (8 / NI);  // the column groups of a row range sit 8 blocks apart (same XCD)
	const int grid = std::max(G, (c->num_cus / G) * G);
	hipLaunchKernelGGL((gemm4_kernel<EPI, NI, NW>), dim3(grid), dim3(NW * 64), lds, s, a);
	return FIND_OK;
}

template <int NI>
static int launch_gemm4(find_ctx* c, int epi, const Gemm2Args& a, int64_t feet, hipStream_t s) {
	if (epi == EPI_BIAS_RELU) return launch_gemm4_t<EPI_BIAS_RELU, NI>(c, a, feet, s);
	if (epi == EPI_MASK) return launch_gemm4_t<EPI_MASK, NI>(c, a, feet, s);
	return launch_gemm4_t<EPI_NONE, NI>(c, a, feet, s);
}

template <int EPI, bool H16 = false, bool VIRT = false>
static int launch_gemm5_t(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {
	int lds = 0;
	const int id = VIRT ? (EPI == EPI_BIAS_RELU ? K_GEMM5_RELU_V : K_GEMM5_MASK_V)
				 : H16 ? (EPI == EPI_BIAS_RELU ? K_GEMM5_RELU_H : K_GEMM5_MASK_H) : K_GEMM5_RELU + (EPI == EPI_BIAS_RELU ? 0 : EPI == EPI_MASK ? 1 : 2);
	const int rc = prepare_kernel(c, id, &gemm5_kernel<EPI, H16, VIRT>, VIRT ? GEMM5_LDS_VIRT : GEMM5_LDS, &lds);
	if (rc != FIND_OK) return rc;
	a.tiles_per_foot = (int)cdiv(a.V, 32);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	a.tile_major = VIRT ? 1 : 0;   // the feet of a tile side by side: the shared product behind the virtual operand is read from HBM once
	const int grid = (int)std::min<int64_t>(c->num_cus, cdiv(a.ntiles, GEMM5_NW));
	hipLaunchKernelGGL((gemm5_kernel<EPI, H16, VIRT>), dim3(grid), dim3(GEMM5_NW * 64), lds, s, a);
	return FIND_OK;
}

static int launch_gemm5(find_ctx* c, int epi, const Gemm2Args& a, int64_t feet, hipStream_t s, bool h16) {
	if (h16) {   // fp16-stored A / y / mask (act16): the two epilogues the heads' hidden layers use
		if (epi == EPI_BIAS_RELU && a.va_bias) return launch_gemm5_t<EPI_BIAS_RELU, true, true>(c, a, feet, s);   // (bcast_fold: mlp_gemm5.h)
		if (epi == EPI_MASK && a.vm_bias) return launch_gemm5_t<EPI_MASK, true, true>(c, a, feet, s);
		if (epi == EPI_BIAS_RELU) return launch_gemm5_t<EPI_BIAS_RELU, true>(c, a, feet, s);
		if (epi == EPI_MASK) return launch_gemm5_t<EPI_MASK, true>(c, a, feet, s);
		set_error("launch_gemm5: no fp16-stored variant of this epilogue");
		return FIND_EINVAL;
	}
	if (epi == EPI_BIAS_RELU) return launch_gemm5_t<EPI_BIAS_RELU>(c, a, feet, s);
	if (epi == EPI_MASK) return launch_gemm5_t<EPI_MASK>(c, a, feet, s);
	return launch_gemm5_t<EPI_NONE>(c, a, feet, s);
}

#ifdef FIND_DIAG   // gemm6 (weight planes in LDS), superseded by gemm7: kept in the diagnostics build for A/B runs ("gemm7" = 0)
template <int EPI>
static int launch_gemm6_t(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {
	int lds = 0;
	const int rc = prepare_kernel(c, K_GEMM6_RELU + (EPI == EPI_BIAS_RELU ? 0 : EPI == EPI_MASK ? 1 : 2), &gemm6_kernel<EPI>, GEMM6_LDS, &lds);
	if (rc != FIND_OK) return rc;
	a.tiles_per_foot = (int)cdiv(a.V, 32);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	constexpr int G = 8 * (8 / G6_NI);  // the column groups of a row range sit 8 blocks apart (same XCD)
	const int grid = std::max(G, (c->num_cus / G) * G);
	if constexpr (EPI == EPI_BIAS_RELU) {
		const int abl = c->x3_abl & 7;   // profiling only (tools/ablate_x3.py)
		if (abl) {
#define FIND_G6_ABL(N) case N: { FIND_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm6_kernel<EPI, N>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes), "hipFuncSetAttribute"); \
			hipLaunchKernelGGL((gemm6_kernel<EPI, N>), dim3(grid), dim3(GEMM6_NW * 64), lds, s, a); return FIND_OK; }
			switch (abl) { FIND_G6_ABL(1) FIND_G6_ABL(2) FIND_G6_ABL(3) FIND_G6_ABL(4) FIND_G6_ABL(5) FIND_G6_ABL(6) FIND_G6_ABL(7) }
#undef FIND_G6_ABL
		}
	}
	hipLaunchKernelGGL((gemm6_kernel<EPI>), dim3(grid), dim3(GEMM6_NW * 64), lds, s, a);
	return FIND_OK;
}
#endif

template <int EPI>
static int launch_gemm7_t(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {
	int lds = 0;
	const int rc = prepare_kernel(c, K_GEMM7_RELU + (EPI == EPI_BIAS_RELU ? 0 : EPI == EPI_MASK ? 1 : 2), &gemm7_kernel<EPI>, GEMM7_LDS, &lds);
	if (rc != FIND_OK) return rc;
	a.tiles_per_foot = (int)cdiv(a.V, 32);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	const int grid = std::max(16, (c->num_cus / 16) * 16);   // the two column halves of a row range sit 8 blocks apart (same XCD)
#ifdef FIND_DIAG
	if constexpr (EPI == EPI_BIAS_RELU) {
		const int abl = c->x3_abl & 15;   // profiling only (tools/ablate_x3.py)
		if (abl) {
#define FIND_G7_ABL(N) case N: { FIND_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm7_kernel<EPI, N>), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes), "hipFuncSetAttribute"); \
			hipLaunchKernelGGL((gemm7_kernel<EPI, N>), dim3(grid), dim3(GEMM7_NW * 64), lds, s, a); return FIND_OK; }
			switch (abl) { FIND_G7_ABL(1) FIND_G7_ABL(2) FIND_G7_ABL(4) FIND_G7_ABL(8) FIND_G7_ABL(3) FIND_G7_ABL(7) FIND_G7_ABL(15) FIND_G7_ABL(12) FIND_G7_ABL(14) }
#undef FIND_G7_ABL
		}
	}
#endif
	hipLaunchKernelGGL((gemm7_kernel<EPI>), dim3(grid), dim3(GEMM7_NW * 64), lds, s, a);
	return FIND_OK;
}

template <int EPI>
static int launch_gemm7_virt(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {   // (bcast_fold: mlp_gemm7.h VIRT)
	int lds = 0;
	FIND_TRY(prepare_kernel(c, EPI == EPI_BIAS_RELU ? K_GEMM7_RELU_V : K_GEMM7_MASK_V, &gemm7_kernel<EPI, 0, true>, GEMM7_LDS, &lds));
	a.tiles_per_foot = (int)cdiv(a.V, 32);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	const int grid = std::max(16, (c->num_cus / 16) * 16);
	hipLaunchKernelGGL((gemm7_kernel<EPI, 0, true>), dim3(grid), dim3(GEMM7_NW * 64), lds, s, a);
	return FIND_OK;
}

// workgroup pairs of the folded-foot-sum launch: the usual count, but never so many that a pair's unit range is shorter than a tile's run of
// feet (mlp_gemm7.h FSUM: a tile then has at most two partial sums); 0 = the shape does not qualify
static int gemm7_fsum_pairs(const find_ctx* c, int64_t V, int64_t feet) {
	const int64_t upf = cdiv(V, 32);
	const int def = std::max(16, (c->num_cus / 16) * 16) / 2;
	const int pairs = (int)std::min<int64_t>(def, (upf / 8) * 8);   // (the two column halves of a pair sit 8 blocks apart: pairs come in eights)
	return (feet >= 2 && feet <= 64 && pairs >= 8) ? pairs : 0;
}

static int launch_gemm7_fsum(find_ctx* c, Gemm2Args a, int64_t feet, hipStream_t s) {   // (footsum_fold: mlp_gemm7.h FSUM)
	const int pairs = gemm7_fsum_pairs(c, a.V, feet);
	FIND_REQUIRE(pairs > 0, "launch_gemm7_fsum: shape does not qualify (footsum_fold and the kernel selection disagree)");
	int lds = 0;
	FIND_TRY(prepare_kernel(c, K_GEMM7_MASK_VF, &gemm7_kernel<EPI_MASK, 0, true, true>, GEMM7_LDS + 64 * 128 * 4, &lds));
	a.tiles_per_foot = (int)cdiv(a.V, 32);
	a.ntiles = (int)(a.tiles_per_foot * feet);
	a.tile_major = 1;
	hipLaunchKernelGGL((gemm7_kernel<EPI_MASK, 0, true, true>), dim3(2 * pairs), dim3(GEMM7_NW * 64), lds, s, a);
	return FIND_OK;
}

static int launch_gemm7(find_ctx* c, int epi, const Gemm2Args& a, int64_t feet, hipStream_t s) {
	if (epi == EPI_MASK && a.vm_bias && a.fs_out) return launch_gemm7_fsum(c, a, feet, s);
	if (epi == EPI_BIAS_RELU && a.va_bias) return launch_gemm7_virt<EPI_BIAS_RELU>(c, a, feet, s);
	if (epi == EPI_MASK && a.vm_bias) return launch_gemm7_virt<EPI_MASK>(c, a, feet, s);
	if (epi == EPI_BIAS_RELU) return launch_gemm7_t<EPI_BIAS_RELU>(c, a, feet, s);
	if (epi == EPI_MASK) return launch_gemm7_t<EPI_MASK>(c, a, feet, s);
	return launch_gemm7_t<EPI_NONE>(c, a, feet, s);
}

#ifdef FIND_DIAG
static int launch_gemm6(find_ctx* c, int epi, const Gemm2Args& a, int64_t feet, hipStream_t s) {
	if (epi == EPI_BIAS_RELU) return launch_gemm6_t<EPI_BIAS_RELU>(c, a, feet, s);
	if (epi == EPI_MASK) return launch_gemm6_t<EPI_MASK>(c, a, feet, s);
	return launch_gemm6_t<EPI_NONE>(c, a, feet, s);
}
#endif

static int launch_gemm3(find_ctx* c, int epi, const Gemm2Args& a, int64_t feet, hipStream_t s) {
	if (epi == EPI_BIAS_RELU) return launch_gemm3_t<64, EPI_BIAS_RELU>(c, a, feet, s);
	if (epi == EPI_MASK) return launch_gemm3_t<64, EPI_MASK>(c, a, feet, s);
	return launch_gemm3_t<64, EPI_NONE>(c, a, feet, s);
}

// Kernel choice for one Linear-shaped launch (all of them exact fp32 unless the call runs in the opt-in fp16 mode):
//   Fourier layer                                   gemm2<64, PE>   (sin / cos generated into the A tile)
//   fp16 mode, K = 256, >= gemm5_min_units units    gemm5           (W resident as fp16, HBM-bound)
//   K = 256, >= gemm4_min_units / 2 32-row units    gemm4<4>        (W half resident in LDS, matrix-pipe-bound)
//   K = 256, >= gemm4_small units                   gemm4<2>        (column quarters: the shared trunk's V rows)
//   anything else (two-segment K, tiny launches)    gemm3<64>       (LDS-DMA ring)
static int launch_gemm(find_ctx* c, int amode, int epi, const GemmArgs& a, int64_t feet, hipStream_t s) {
	Gemm2Args b;
	memset(&b, 0, sizeof(b));
	b.a0 = a.a0; b.a1 = a.a1; b.nseg = a.nbase; b.a_foot_stride = a.a_foot_stride; b.lda = a.lda;
	b.pos = a.pos; b.pos_foot_stride = a.pos_foot_stride; b.Bm = a.Bm; b.pe = a.pe;
	b.w0 = a.w0; b.w1 = a.w1; b.ldw = a.ldw; b.nchunk = a.nchunk; b.w_tr = a.w_tr;
	b.bias = a.bias; b.bias_foot_stride = a.bias_foot_stride; b.mask = a.mask; b.mask_foot_stride = a.mask_foot_stride;
	b.y = a.y; b.y_foot_stride = a.y_foot_stride; b.ldy = a.ldy; b.V = a.V; b.ablate = c->ablate; b.dbg = c->dbg;
	b.va_bias = a.va_bias; b.va_bias_stride = a.va_bias_stride; b.vm_bias = a.vm_bias; b.vm_bias_stride = a.vm_bias_stride;
	b.fs_out = a.fs_out; b.fs_slot_stride = a.fs_slot_stride; b.cs_out = a.cs_out;
	if (amode == AMODE_PE) return launch_gemm2_pe(c, b, feet, s);
	const int64_t units = cdiv(a.V, 32) * feet;
	const bool k256 = b.nseg == 1 && b.nchunk == 8;
	// (gemm5 keeps the whole W per workgroup, so 216 units occupy 27 CUs: 22 us against 13 us for gemm4 on column quarters)
	if (c->f16 && k256 && units >= c->gemm5_min_units) return launch_gemm5(c, epi, b, feet, s, a.h16 != 0);
	FIND_REQUIRE(!a.h16, "launch_gemm: an fp16-stored layer reached a kernel that reads fp32 (act16 and the kernel selection disagree)");
#ifdef FIND_DIAG
	if (c->x3 && k256 && units >= c->gemm6_min_units && !c->gemm7) { FIND_REQUIRE(!a.va_bias && !a.vm_bias, "launch_gemm: gemm6 forms no virtual operand"); return launch_gemm6(c, epi, b, feet, s); }
#endif
	if (c->x3 && k256 && units >= c->gemm6_min_units) return launch_gemm7(c, epi, b, feet, s);
	FIND_REQUIRE(!a.va_bias && !a.vm_bias && !a.fs_out, "launch_gemm: a virtual operand reached a kernel that cannot form it (bcast_fold / footsum_fold and the kernel selection disagree)");
	FIND_REQUIRE(!a.w_tr, "launch_gemm: an untransposed weight reached a kernel that cannot read it (gemm7_direct and the kernel selection disagree)");
	if (k256 && units * 2 >= c->gemm4_min_units) return launch_gemm4<4>(c, epi, b, feet, s);
	if (k256 && c->gemm4_small && units >= c->gemm4_small) return launch_gemm4<2>(c, epi, b, feet, s);
	return launch_gemm3(c, epi, b, feet, s);
}

// a K = 256, one-segment launch of this many rows per foot goes to gemm7 (launch_gemm's rule): its dX form may then read the model's weight
// itself (Gemm2Args::w_tr) instead of a transposed copy
static bool gemm7_direct(const find_ctx* c, int64_t V, int64_t feet) {
	const int64_t units = cdiv(V, 32) * feet;
#ifdef FIND_DIAG
	if (!c->gemm7) return false;
#endif
	return c->direct_w && c->x3 && !(c->f16 && units >= c->gemm5_min_units) && units >= c->gemm6_min_units;
}
// ... whatever the weight's orientation: the launch will go to gemm7 (launch_gemm's rule)
static bool gemm7_direct_units(const find_ctx* c, int64_t V, int64_t feet) {
	const int64_t units = cdiv(V, 32) * feet;
#ifdef FIND_DIAG
	if (!c->gemm7) return false;
#endif
	return c->x3 && !(c->f16 && units >= c->gemm5_min_units) && units >= c->gemm6_min_units;
}
// a fused chain of this call will run on fused6_kernel (chain_prepare's rule): split_w_kernel reads every weight once anyway, in whatever
// order the step asks for (FusedStep::wmode) -- no repack launch in front of the chain
static bool chain_direct(const find_ctx* c) { return c->direct_w && c->x3 && c->fused6; }

// ---- fused chains (mlp_fused.h): one launch takes every 32-row tile through a list of layers
struct Chain {
	FusedArgs a;
	int nt = 1;             // 32-row blocks per tile (chain_prepare)
	bool x3 = false;        // runs on fused6_kernel: its weights have been split (chain_prepare)
	bool prepared = false;
	int in_dim = 3;         // of the Fourier layer (wmode 2)
	Chain() { memset(&a, 0, sizeof(a)); }
	FusedStep& add() { return a.step[a.n_steps++]; }
	bool full(int more) const { return a.n_steps + more > FUSED_MAX_STEPS; }
	// y = epi(x @ w^T): x from LDS (the previous step's result) unless src is given
	FusedStep& gemm(const float* w, int ldw, int nchunk, const float* src = nullptr) {
		FusedStep& s = add();
		s.kind = FS_GEMM; s.w = w; s.ldw = ldw; s.nchunk = nchunk;
		s.src = src; s.src_kind = src ? FS_SRC_GLOBAL : FS_SRC_LDS;
		return s;
	}
};

// A chain runs in two halves: chain_prepare fixes the tile geometry and -- bf16x3 -- launches split_w_kernel (it needs the weights only, so a
// caller may issue it long before the chain's inputs exist: the shared trunk's backward chain does, right behind the transposes, instead of
// between the large weight-gradient kernels, where this 5-us launch waited 90 us for a free CU on the step's critical path); chain_launch
// starts the chain itself.
static int chain_prepare(find_ctx* c, Chain& ch, int64_t V, int64_t feet, hipStream_t s, void* w6, int64_t w6_bytes) {
	// more 32-row blocks than CUs: 64-row tiles (one round of workgroups instead of two, half the weight staging per MFMA)
	ch.nt = (cdiv(V, 32) * feet > c->num_cus && !(c->ablate & 128)) ? 2 : 1;
	ch.a.V = (int)V;
	ch.a.tiles_per_foot = (int)cdiv(V, 32 * ch.nt);
	ch.a.ntiles = (int)(ch.a.tiles_per_foot * feet);
	ch.a.ablate = c->ablate;
	ch.x3 = false;
	ch.prepared = true;
	if (c->x3 && c->fused6 && w6 != nullptr) {
		// bf16x3: the chain's weights as fragment-ordered bf16 planes (one small launch), then the chain on the bf16 matrix pipe
		SplitWArgs sa;
		memset(&sa, 0, sizeof(sa));
		int maxn = 0;
		for (int i = 0; i < ch.a.n_steps; ++i) {
			const FusedStep& st = ch.a.step[i];
			if (st.kind != FS_GEMM) continue;
			const int n = 2 * st.nchunk;
			sa.job[sa.njobs++] = SplitWJob{st.w, st.ldw, n, sa.total, st.wmode, ch.a.pe, ch.in_dim};
			sa.total += n;
			maxn = std::max(maxn, n);
		}
		if (sa.njobs > 0 && 8 * (int64_t)sa.total * F6_STEP_BYTES <= w6_bytes) {
			sa.dst = reinterpret_cast<u32x4*>(w6);
			hipLaunchKernelGGL(split_w_kernel, dim3((unsigned)cdiv(8 * maxn * 64, 256), (unsigned)sa.njobs), dim3(256), 0, s, sa);
			FIND_LAUNCH_CHECK("split_w_kernel");
			ch.a.w6 = w6;
			ch.a.total_steps = sa.total;
			ch.x3 = true;
		}
	}
	if (!ch.x3)
		for (int i = 0; i < ch.a.n_steps; ++i)
			FIND_REQUIRE(ch.a.step[i].kind != FS_GEMM || ch.a.step[i].wmode == 0, "find_mlp: a chain with weights in model order did not get its bf16x3 kernel (chain_direct and chain_prepare disagree)");
	return FIND_OK;
}

static int chain_launch(find_ctx* c, Chain& ch, hipStream_t s) {
	const int nt = ch.nt;
	int lds = 0;
	if (ch.x3) {
		if (nt == 2) FIND_TRY(prepare_kernel(c, K_FUSED6_2, &fused6_kernel<2>, fused6_lds(2), &lds, false));
		else FIND_TRY(prepare_kernel(c, K_FUSED6, &fused6_kernel<1>, fused6_lds(1), &lds, false));
		if (nt == 2) hipLaunchKernelGGL(fused6_kernel<2>, dim3(ch.a.ntiles), dim3(FUSED_NW * 64), lds, s, ch.a);   // (one tile per workgroup)
		else hipLaunchKernelGGL(fused6_kernel<1>, dim3(ch.a.ntiles), dim3(FUSED_NW * 64), lds, s, ch.a);
		FIND_LAUNCH_CHECK("fused6_kernel");
		return FIND_OK;
	}
	const int grid = std::min(ch.a.ntiles, c->num_cus);
	if (nt == 2) FIND_TRY(prepare_kernel(c, K_FUSED2, &fused_chain_kernel<2>, fused_lds(2), &lds, false));
	else FIND_TRY(prepare_kernel(c, K_FUSED, &fused_chain_kernel<1>, fused_lds(1), &lds, false));   // no LDS-DMA in this kernel: no reservation
	if (nt == 2) hipLaunchKernelGGL(fused_chain_kernel<2>, dim3(grid), dim3(FUSED_NW * 64), lds, s, ch.a);
	else hipLaunchKernelGGL(fused_chain_kernel<1>, dim3(grid), dim3(FUSED_NW * 64), lds, s, ch.a);
	FIND_LAUNCH_CHECK("fused_chain_kernel");
	return FIND_OK;
}

static int launch_chain(find_ctx* c, Chain& ch, int64_t V, int64_t feet, hipStream_t s, void* w6 = nullptr, int64_t w6_bytes = 0) {
	if (!ch.prepared) FIND_TRY(chain_prepare(c, ch, V, feet, s, w6, w6_bytes));
	return chain_launch(c, ch, s);
}

static bool use_fused(const find_ctx* c, int64_t V, int64_t feet) {
	const int64_t units = cdiv(V, 32) * feet;
	if (c->f16 && units >= c->gemm5_min_units) return false;   // the opt-in fp16 kernels take launches of this size (the chain is fp32)
	return c->fused_max_units > 0 && units <= c->fused_max_units;
}

static GemmArgs gemm_args_zero() {
	GemmArgs a;
	memset(&a, 0, sizeof(a));
	a.nbase = 1;
	a.nseg_per_base = 1;
	return a;
}

// Linear + ReLU forward:  y = relu(x @ w^T + bias[foot])
static int linear_fwd(find_ctx* c, const float* x, int64_t x_foot_stride, const float* w, int ldw, const float* bias,
					  int64_t bias_foot_stride, float* y, int64_t V, int64_t feet, hipStream_t s, bool h16 = false, const float* va_bias = nullptr,
					  int64_t va_bias_stride = 0) {
	GemmArgs a = gemm_args_zero();
	a.h16 = h16;
	a.va_bias = va_bias; a.va_bias_stride = va_bias_stride;   // (then x is the shared fp32 product and x_foot_stride 0: bcast_fold)
	a.a0 = x; a.a_foot_stride = x_foot_stride; a.lda = W;
	a.w0 = w; a.ldw = ldw; a.nchunk = W / KC;
	a.bias = bias; a.bias_foot_stride = bias_foot_stride;
	a.y = y; a.y_foot_stride = V * W; a.ldy = W; a.V = (int)V;
	return launch_gemm(c, AMODE_MAT, EPI_BIAS_RELU, a, feet, s);
}

static void split_policy(int64_t n_feet, int64_t V, int* spf, int* cps, int64_t target = 128) {
	const int64_t cpf = cdiv(V, 32);
	int64_t s0 = std::max<int64_t>(1, std::min<int64_t>(cpf, cdiv(target, n_feet)));
	*cps = (int)cdiv(cpf, s0);
	*spf = (int)cdiv(cpf, *cps);
}

// act16: the opt-in fp16 mode STORES the heads' hidden activations (w.D / w.C) and their gradients (b.dzD / b.dzC) as fp16 when every
// kernel that touches them is one of the HBM-bound large-shape kernels: a template shared by more than one foot (bias_relu_bcast ->
// gemm5 -> head_out forward; head_out_bwd -> dw3 / gemm5 / footsum backward) with enough rows for gemm5.  Forward and backward of a
// call agree on it through a note the forward leaves in the context (note_act16), so a knob turned in between cannot split them.
static bool use_act16(const find_ctx* c, bool f16, bool shared, int64_t n_feet, int64_t V) {
	return f16 && c->act16 && shared && n_feet > 1 && cdiv(V, 32) * n_feet >= c->gemm5_min_units;
}
// bcast_fold (round 6): inside act16 the output of a head's broadcast first layer, h1 = fp16(relu(P[v] + bias[foot])), is never stored: its
// three readers (the second layer's forward GEMM, the ReLU mask of that layer's dX GEMM, the x operand of its weight gradient) form it from the
// V x 256 product P (w.hp / w.hp2, kept until the backward) and the bias rows -- gemm5_kernel<.., VIRT>, dw3_h16v_kernel.  Needs a second hidden layer.
// The same in the default bf16x3 arithmetic (fp32-stored activations): gemm7_kernel<.., VIRT>, dw6v_kernel -- wherever the heads' hidden
// layers of a shared template go to gemm7 / dw6 (launch_gemm's and weight_grad's rule).
static bool use_fold(const find_ctx* c, bool a16, bool shared, int64_t n_feet, int64_t V, const find_mlp_params* p) {
	if (!c->bcast_fold || p->n_disp < 2 || p->n_col < 2) return false;
	if (a16) return true;
	const int64_t units = cdiv(V, 32) * n_feet;
	bool x3_large = c->x3 && !c->f16 && shared && n_feet > 1 && units >= c->gemm6_min_units;
#ifdef FIND_DIAG
	x3_large = x3_large && c->gemm7;
#endif
	return x3_large;
}
static void note_act16(find_ctx* c, const void* ws, bool a16, bool fold) {
	for (auto& n : c->act16_notes) if (n.ws == ws) { n.a16 = a16; n.fold = fold; return; }
	c->act16_notes[c->act16_next] = find_ctx::Act16Note{ws, a16, fold};
	c->act16_next = (c->act16_next + 1) & 15;
}
static int noted_act16(const find_ctx* c, const void* ws, bool by_rule, bool fold_by_rule) {   // bit 0: act16, bit 1: bcast_fold
	for (const auto& n : c->act16_notes) if (n.ws == ws) return (n.a16 ? 1 : 0) | (n.fold ? 2 : 0);
	return (by_rule ? 1 : 0) | (fold_by_rule ? 2 : 0);
}

static bool call_f16(const find_ctx* c, const find_mlp_params* p) { return p->precision == 2 || (p->precision == 0 && c->mlp_f16 == 1); }
static bool call_x3(const find_ctx* c, const find_mlp_params* p) { return p->precision == 3 || (p->precision == 0 && c->mlp_f16 == 2); }

static int mlp_fwd_body(find_ctx* c, Fork& fk, const find_mlp_params* p, const Dims& d, const FwdWs& w, const float* pos, const float* lat_disp,
						const float* lat_col, float* disp, float* col) {
	hipStream_t s = fk.s;
	const int64_t V = d.V, n_feet = d.n_feet;
	const int ld_d0 = W + p->lat_disp, ld_c0 = W + p->lat_col;
	const bool a16 = use_act16(c, c->f16, d.shared, n_feet, V);
	const bool fold = use_fold(c, a16, d.shared, n_feet, V, p);
	note_act16(c, w.fbd, a16, fold);   // (every forward leaves its note, keyed by a buffer every workspace has: the backward follows it)

	// 1. repack: layer-0 weight into padded PE order; main blocks of the two head input layers.  Not for a bf16x3 chain that carries the
	// whole call (or everything up to the heads' broadcast first layers): its weight split reads the model's tensors directly (round 6: this
	// launch sat in front of both MLP passes of a training step, 8 - 20 us each on the critical path)
	const bool fused_early = use_fused(c, V, d.feet_t) && p->pe_size > 0;
	const bool direct = fused_early && chain_direct(c);
	if (!direct) {
		RepackArgs ra;
		memset(&ra, 0, sizeof(ra));
		ra.njobs = 3;
		ra.job[0] = RepackJob{p->trunk_w[0], w.w0p, W, KP0, p->in_dim + 2 * p->pe_size, 0, KP0, 2, p->pe_size, p->in_dim};
		ra.job[1] = RepackJob{p->disp_w[0], w.wd0, W, W, ld_d0, 0, W, 0, 0, 0};
		ra.job[2] = RepackJob{p->col_w[0], w.wc0, W, W, ld_c0, 0, W, 0, 0, 0};
		hipLaunchKernelGGL(repack_kernel, dim3(96, ra.njobs), dim3(256), 0, s, ra);
		FIND_LAUNCH_CHECK("repack_kernel");
	}
	// 2. per-foot latent bias (model.py:428-437 as a bias)
	const float* bias_d0 = p->disp_b[0];
	const float* bias_c0 = p->col_b[0];
	int64_t bstride_d = 0, bstride_c = 0;
	// (only for the heads this call evaluates: the template pass of a 3-D-loss step leaves the colour head out, the texture pass the other one)
	if (p->lat_disp > 0 && disp != nullptr) {
		hipLaunchKernelGGL(latent_bias_kernel, dim3(W / 4, (unsigned)n_feet), dim3(256), 0, s, p->disp_w[0], ld_d0, p->disp_b[0], lat_disp, p->lat_disp, w.fbd);
		bias_d0 = w.fbd; bstride_d = W;
	}
	if (p->lat_col > 0 && col != nullptr) {
		hipLaunchKernelGGL(latent_bias_kernel, dim3(W / 4, (unsigned)n_feet), dim3(256), 0, s, p->col_w[0], ld_c0, p->col_b[0], lat_col, p->lat_col, w.fbc);
		bias_c0 = w.fbc; bstride_c = W;
	}
	FIND_LAUNCH_CHECK("latent_bias_kernel");

	// 3 (+ 4, 5 for small calls). trunk (model.py:421-426); layer 0 generates the Fourier features on the fly
	const bool fused = fused_early;
	const bool fused_heads = fused && !(d.shared && n_feet > 1);   // per-foot rows: the heads are as small as the trunk
	if (fused) {
		Chain ch;
		ch.a.pos = pos; ch.a.pos_foot_stride = V * 3; ch.a.Bm = p->B; ch.a.pe = p->pe_size; ch.in_dim = p->in_dim;
		const float* const wd0 = direct ? p->disp_w[0] : w.wd0;
		const float* const wc0 = direct ? p->col_w[0] : w.wc0;
		const int ldd0 = direct ? ld_d0 : W, ldc0 = direct ? ld_c0 : W;
		{
			// even chunk counts (the padding columns of w0p are zeros; wmode 2 reads them as zeros)
			FusedStep& s0 = direct ? ch.gemm(p->trunk_w[0], p->in_dim + 2 * p->pe_size, (d.nchunk0 + 1) & ~1) : ch.gemm(w.w0p, KP0, (d.nchunk0 + 1) & ~1);
			s0.wmode = direct ? 2 : 0;
			s0.src_kind = FS_SRC_PE; s0.relu = 1; s0.bias = p->trunk_b[0]; s0.dst = w.H[0]; s0.to_lds = 1;
		}
		for (int i = 1; i < p->n_trunk; ++i) {
			FusedStep& st = ch.gemm(p->trunk_w[i], W, W / KC);
			st.relu = 1; st.bias = p->trunk_b[i]; st.dst = w.H[i]; st.to_lds = 1;
		}
		const float* hlast = w.H[p->n_trunk - 1];
		if (!fused_heads) {
			// shared template: the first layer of each head is H W0^T on the V template rows (bias + ReLU are broadcast per foot below)
			if (disp) { FusedStep& st = ch.gemm(wd0, ldd0, W / KC); st.dst = w.hp; }
			if (col) { FusedStep& st = ch.gemm(wc0, ldc0, W / KC); st.dst = (disp || fold) ? w.hp2 : w.hp; }
		} else {
			auto head = [&](int which, float* const* act, int nl, const float* w0, int ld0, const float* b0, int64_t bstride, const float* const* hw, const float* const* hb,
							float* z, float* out, bool reload) {
				FusedStep& f0 = ch.gemm(w0, ld0, W / KC, reload ? hlast : nullptr);
				f0.relu = 1; f0.bias = b0; f0.bias_foot_stride = (int)bstride; f0.dst = act[0]; f0.to_lds = 1;
				for (int i = 1; i < nl; ++i) {
					FusedStep& st = ch.gemm(hw[i], W, W / KC);
					st.relu = 1; st.bias = hb[i]; st.dst = act[i]; st.to_lds = 1;
				}
				FusedStep& o = ch.add();
				o.kind = FS_OUT; o.w = hw[nl]; o.bias = hb[nl]; o.dst = out; o.dst2 = z; o.head = (unsigned char)which;
				o.aux = which ? p->avg_col : nullptr;
			};
			if (disp) head(0, w.D, p->n_disp, wd0, ldd0, bias_d0, bstride_d, p->disp_w, p->disp_b, w.zd, disp, false);
			if (col) head(1, w.C, p->n_col, wc0, ldc0, bias_c0, bstride_c, p->col_w, p->col_b, w.zc, col, disp != nullptr);
		}
		FIND_TRY(launch_chain(c, ch, V, d.feet_t, s, w.w6, chain_w6_bytes(p)));
		if (fused_heads) return FIND_OK;
	} else {
		GemmArgs a = gemm_args_zero();
		a.pos = pos; a.pos_foot_stride = V * 3; a.Bm = p->B; a.pe = p->pe_size;
		a.w0 = w.w0p; a.ldw = KP0; a.nchunk = d.nchunk0;
		a.bias = p->trunk_b[0]; a.bias_foot_stride = 0;
		a.y = w.H[0]; a.y_foot_stride = V * W; a.ldy = W; a.V = (int)V;
		FIND_TRY(launch_gemm(c, AMODE_PE, EPI_BIAS_RELU, a, d.feet_t, s));
		for (int i = 1; i < p->n_trunk; ++i)
			FIND_TRY(linear_fwd(c, w.H[i - 1], V * W, p->trunk_w[i], W, p->trunk_b[i], 0, w.H[i], V, d.feet_t, s));
		FIND_LAUNCH_CHECK("trunk gemm");
	}

	// 4. heads (model.py:439-440); the trunk rows are shared by every foot when d.shared
	const float* hl = w.H[p->n_trunk - 1];
	const int64_t hl_stride = d.shared ? 0 : V * W;
	// first layer of a head.  Shared template: every foot multiplies the SAME trunk rows, so  H W^T  is formed once on V rows
	// and each foot only adds its (latent-folded) bias and applies the ReLU -- a bandwidth-bound broadcast instead of a GEMM
	// over n_feet * V rows.  (The backward uses the same fact: footsum_kernel.)
	auto head_first = [&](const float* w0, const float* bias, int64_t bstride, float* out, float* hp, hipStream_t st) -> int {
		if (hp) {
			if (!fused) {   // (the fused trunk launch has already formed the product)
				GemmArgs a = gemm_args_zero();
				a.a0 = hl; a.a_foot_stride = 0; a.lda = W;
				a.w0 = w0; a.ldw = W; a.nchunk = W / KC;
				a.y = hp; a.y_foot_stride = V * W; a.ldy = W; a.V = (int)V;
				FIND_TRY(launch_gemm(c, AMODE_MAT, EPI_NONE, a, 1, st));
			}
			if (fold) return FIND_OK;   // (the second layer reads hp and the bias rows itself)
			hipLaunchKernelGGL(bias_relu_bcast_kernel, dim3((unsigned)cdiv(V * (W / 4), 256), (unsigned)cdiv(n_feet, BCAST_FEET)), dim3(256), 0, st, hp, bias,
							   bstride, (int)n_feet, V, out, a16 ? 1 : 0);
			return FIND_OK;
		}
		FIND_REQUIRE(!a16, "find_mlp_fwd: act16 without a shared template");
		return linear_fwd(c, hl, hl_stride, w0, W, bias, bstride, out, V, n_feet, st);
	};
	// 5. final 256->3 layers + tanh scalings (model.py:444-449), one launch per head
	auto head_out = [&](int head, hipStream_t st) {
		HeadOutArgs h;
		memset(&h, 0, sizeof(h));
		h.x[0] = w.D[p->n_disp - 1]; h.x[1] = w.C[p->n_col - 1];
		h.w[0] = p->disp_w[p->n_disp]; h.w[1] = p->col_w[p->n_col];
		h.b[0] = p->disp_b[p->n_disp]; h.b[1] = p->col_b[p->n_col];
		h.z[0] = w.zd; h.z[1] = w.zc;
		h.out[0] = disp; h.out[1] = col;
		h.avg_col = p->avg_col;
		h.rows = d.rows_h;
		h.head0 = head;
		h.x_half = a16 ? 1 : 0;
		const unsigned gx = (unsigned)std::min<int64_t>(cdiv(d.rows_h, 32), 2048);
		if (a16) hipLaunchKernelGGL(head_out_fwd_h16_kernel, dim3(gx, 1), dim3(256), 0, st, h);   // (fp16-stored activations: on the fp16 matrix pipe)
		else hipLaunchKernelGGL(head_out_fwd_kernel, dim3(gx, 1), dim3(256), 0, st, h);
	};
	// The heads are independent after the trunk.  With both active, the colour head runs on a side stream: its bandwidth-bound
	// pieces (the bias + ReLU broadcast, the 3-wide output layer: no LDS, so they can share CUs with the W-resident GEMMs) then
	// overlap the other head's matrix-pipe-bound layers.  Forked from and joined back into the caller's stream (Fork::join).
	hipStream_t sc = s;
	if (disp && col && fk.on) {
		fk.fork_to(1);
		sc = fk.stream(1);
	}
	if (disp) {
		FIND_TRY(head_first(w.wd0, bias_d0, bstride_d, w.D[0], w.hp, s));
		for (int i = 1; i < p->n_disp; ++i) {
			if (i == 1 && fold) FIND_TRY(linear_fwd(c, w.hp, 0, p->disp_w[i], W, p->disp_b[i], 0, w.D[i], V, n_feet, s, a16, bias_d0, bstride_d));
			else FIND_TRY(linear_fwd(c, w.D[i - 1], V * W, p->disp_w[i], W, p->disp_b[i], 0, w.D[i], V, n_feet, s, a16));
		}
		head_out(0, s);
	}
	if (col) {
		// (bcast_fold: the products stay until the backward -- the colour head's is ALWAYS hp2, the other head's hp)
		float* const hpc = (fold || sc != s || (fused && disp)) ? w.hp2 : w.hp;
		FIND_TRY(head_first(w.wc0, bias_c0, bstride_c, w.C[0], hpc, sc));
		for (int i = 1; i < p->n_col; ++i) {
			if (i == 1 && fold) FIND_TRY(linear_fwd(c, hpc, 0, p->col_w[i], W, p->col_b[i], 0, w.C[i], V, n_feet, sc, a16, bias_c0, bstride_c));
			else FIND_TRY(linear_fwd(c, w.C[i - 1], V * W, p->col_w[i], W, p->col_b[i], 0, w.C[i], V, n_feet, sc, a16));
		}
		head_out(1, sc);
	}
	FIND_LAUNCH_CHECK("head layers");
	return FIND_OK;
}

}  // namespace mlp
}  // namespace find

using namespace find;
using namespace find::mlp;

static int check_ctx(const find_ctx* c, const char* who) {
	FIND_REQUIRE(c != nullptr, "%s: ctx is NULL (find_ctx_create)", who);
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess || dev != c->device) {
		set_error("%s: the context belongs to device %d, the calling thread's current device is %d", who, c->device, dev);
		return FIND_EINVAL;
	}
	return FIND_OK;
}

extern "C" int64_t find_mlp_ws_bytes(const find_mlp_params* p, int64_t pos_batch, int64_t n_feet, int64_t n_pts, int save_for_bwd) {
	Dims d;
	if (make_dims(p, pos_batch, n_feet, n_pts, &d) != FIND_OK) return -1;
	FwdWs w;
	carve_fwd(p, d, save_for_bwd != 0, nullptr, &w);
	return w.bytes;
}

extern "C" int find_mlp_fwd(find_ctx* c, const find_mlp_params* p, const float* pos, int64_t pos_batch, int64_t n_feet, int64_t n_pts,
							const float* lat_disp, const float* lat_col, float* disp, float* col, void* ws,
							int64_t ws_bytes, int save_for_bwd, void* stream) {
	FIND_TRY(check_ctx(c, "find_mlp_fwd"));
	Dims d;
	FIND_TRY(make_dims(p, pos_batch, n_feet, n_pts, &d));
	FIND_TRY(check_weights(p));
	FIND_REQUIRE(pos && ws, "find_mlp_fwd: pos/ws is NULL");
	FIND_REQUIRE(disp || col, "find_mlp_fwd: both outputs NULL");
	// (the latents of a head the call does not evaluate may be NULL: nothing reads them)
	FIND_REQUIRE(p->lat_disp == 0 ? lat_disp == nullptr : (lat_disp != nullptr || disp == nullptr), "find_mlp_fwd: lat_disp pointer does not match params.lat_disp=%d", p->lat_disp);
	FIND_REQUIRE(p->lat_col == 0 ? lat_col == nullptr : (lat_col != nullptr || col == nullptr), "find_mlp_fwd: lat_col pointer does not match params.lat_col=%d", p->lat_col);
	FIND_REQUIRE(p->precision >= 0 && p->precision <= 3, "find_mlp_fwd: params.precision must be 0 (context default), 1 (fp32 MFMA), 2 (fp16) or 3 (bf16x3), got %d", p->precision);
	FwdWs w;
	carve_fwd(p, d, save_for_bwd != 0, ws, &w);
	if (ws_bytes < w.bytes) {
		set_error("find_mlp_fwd: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
		return FIND_EWORKSPACE;
	}
	c->f16 = call_f16(c, p);
	c->x3 = call_x3(c, p);
	Fork fk(c, reinterpret_cast<hipStream_t>(stream), c->fwd_streams != 0);
	const int rc = mlp_fwd_body(c, fk, p, d, w, pos, lat_disp, lat_col, disp, col);
	const int rj = fk.join();   // on every path: the caller may free `ws` right after an error return
	return rc != FIND_OK ? rc : rj;
}

extern "C" int find_linear_relu_fwd(find_ctx* c, const float* x, const float* w, const float* b, int64_t n_feet, int64_t n_pts, float* y, void* stream) {
	FIND_TRY(check_ctx(c, "find_linear_relu_fwd"));
	FIND_REQUIRE(x && w && b && y, "find_linear_relu_fwd: NULL argument");
	FIND_REQUIRE(n_feet >= 1 && n_pts >= 1 && n_feet < (1 << 16), "find_linear_relu_fwd: bad sizes");
	FIND_REQUIRE(aligned16(w) && aligned16(x), "find_linear_relu_fwd: x and w must be 16-byte aligned");
	c->f16 = c->mlp_f16 == 1;
	c->x3 = c->mlp_f16 == 2;
	FIND_TRY(linear_fwd(c, x, n_pts * W, w, W, b, 0, y, n_pts, n_feet, reinterpret_cast<hipStream_t>(stream)));
	FIND_LAUNCH_CHECK("find_linear_relu_fwd");
	return FIND_OK;
}

// ------------------------------------------------------------------------------------------- backward
namespace find {
namespace mlp {

struct BwdWs {
	float* Tt[FIND_MAX_LAYERS];  // transposed trunk weights (layers >= 1)
	float* Dt[FIND_MAX_LAYERS];  // transposed disp-head weights (layer 0: main block)
	float* Ct[FIND_MAX_LAYERS];
	float* dzD[FIND_MAX_LAYERS];  // one per head layer (as dzT)
	float* dzC[FIND_MAX_LAYERS];
	float* dzT[FIND_MAX_LAYERS];  // one per trunk layer: the dX chain never waits for the side stream's readers
	float* pw;    // dW partial slabs
	float* pb;    // bias partial slabs
	float* pw_t[4];  // slab sets: [0] aliases pw / pb (stream q), [1], [2] the trunk's side streams, [3] alternates with [0] on q
	float* pb_t[4];
	float* Sd;    // (n_feet,256) per-foot column sums of the disp head's first-layer dZ
	float* Sc;
	float* zsD;   // shared template: (V,256) sum over feet of the disp head's first-layer dZ
	float* zsC;
	float* fs1D;  // footsum_fold: the second partial sum of gemm7_kernel<.., FSUM> (the first lands in zsD / zsC)
	float* fs1C;
	float* pS;    // [nblk_fs][n_feet][256] partial per-foot column sums (disp head)
	float* pS2;   // same for the colour head: the reduces run on the side stream, so the heads cannot share one
	int nblk_fs;
	float* pwo[2];  // final-layer partials
	float* pbo[2];
	int nblk_out;
	int64_t max_split;
	float* grp_pw;   // small calls: slabs of the grouped weight-gradient launch, [job][slab][256][256], then the bias rows [job][slab][256]
	int64_t grp_slabs;  // slabs per job
	int grp_jobs;
	void* w6;        // fused6_kernel: the dX chain's weights as fragment-ordered bf16 planes
	int64_t bytes;
};

constexpr int64_t GROUP_MAX_UNITS = 1024;   // largest call (32-row tiles) that may take the fused / grouped small-call path
constexpr int GROUP_MIN_CPS = 8;            // 16-row chunks per workgroup of a grouped weight gradient, at least

static void carve_bwd(const find_mlp_params* p, const Dims& d, void* scratch, BwdWs* o) {
	Carver c(scratch);
	for (int i = 1; i < p->n_trunk; ++i) o->Tt[i] = c.take<float>((int64_t)W * W);
	for (int i = 0; i < p->n_disp; ++i) o->Dt[i] = c.take<float>((int64_t)W * W);
	for (int i = 0; i < p->n_col; ++i) o->Ct[i] = c.take<float>((int64_t)W * W);
	for (int i = 0; i < std::max(p->n_disp, 1); ++i) o->dzD[i] = c.take<float>(d.rows_h * W);
	for (int i = 0; i < std::max(p->n_col, 1); ++i) o->dzC[i] = c.take<float>(d.rows_h * W);
	for (int i = 0; i < std::max(p->n_trunk, 1); ++i) o->dzT[i] = c.take<float>(d.rows_t * W);
	int spf, cps;
	split_policy(d.n_feet, d.V, &spf, &cps);
	int64_t ms = d.n_feet * spf;
	split_policy(1, d.V, &spf, &cps);
	ms = std::max<int64_t>(ms, spf);
	o->max_split = ms;
	// dw2 policy: up to max(#CUs, feet) main slabs + one tail slab per foot, each 256x256
	const int64_t ms2 = std::max<int64_t>(512, d.n_feet) + d.n_feet + 16;
	o->pw = c.take<float>(std::max<int64_t>(ms * W * KP0, ms2 * W * W));
	o->pb = c.take<float>(std::max<int64_t>(ms, ms2) * W);
	o->pw_t[0] = o->pw; o->pb_t[0] = o->pb;
	for (int i = 1; i < 4; ++i) {  // trunk layers have matrix inputs (dw2 slabs) except layer 0, which always uses set 0
		o->pw_t[i] = c.take<float>(ms2 * W * W);
		o->pb_t[i] = c.take<float>(ms2 * W);
	}
	o->Sd = c.take<float>(d.n_feet * W);
	o->Sc = c.take<float>(d.n_feet * W);
	o->nblk_fs = (int)cdiv(d.V, FS_ROWS);
	if (d.shared) {
		o->zsD = c.take<float>(d.V * W);
		o->zsC = c.take<float>(d.V * W);
		o->fs1D = c.take<float>(d.V * W);
		o->fs1C = c.take<float>(d.V * W);
	} else {
		o->zsD = o->zsC = o->fs1D = o->fs1C = nullptr;
	}
	// (also without a shared template: the latents-only backward of a frozen network takes its per-foot column sums this way)
	o->pS = c.take<float>((int64_t)o->nblk_fs * d.n_feet * W);
	o->pS2 = c.take<float>((int64_t)o->nblk_fs * d.n_feet * W);
	o->w6 = c.take<char>(chain_w6_bytes(p));
	o->grp_pw = nullptr; o->grp_slabs = 0; o->grp_jobs = 0;
	if (cdiv(d.V, 32) * d.feet_t <= GROUP_MAX_UNITS) {
		// (a shared trunk groups its own layers only: the heads' layers there have n_feet times the rows and keep their own launches)
		// (round 6: ... and the heads' FIRST layers, whose foot-summed gradient has the trunk's V rows)
		o->grp_jobs = d.shared ? (p->n_trunk - 1) + 2 : (p->n_trunk - 1) + p->n_disp + p->n_col;
		o->grp_slabs = std::max<int64_t>(d.feet_t * (cdiv(d.V / 16, GROUP_MIN_CPS) + 1), std::min<int64_t>(d.feet_t * cdiv(d.V / 16, 2), 32));
		if (o->grp_jobs > 0) o->grp_pw = c.take<float>((int64_t)o->grp_jobs * o->grp_slabs * ((int64_t)W * W + W));
	}
	o->nblk_out = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(d.rows_h, 64), 512));
	for (int i = 0; i < 2; ++i) {
		o->pwo[i] = c.take<float>((int64_t)o->nblk_out * 3 * W);
		o->pbo[i] = c.take<float>((int64_t)o->nblk_out * 4);
	}
	o->bytes = c.off;
}

// (diagnosis) dynamic LDS of the slab reduce: 0, or everything the CU has left beside its 16 KB of static LDS
static int reduce_lds(find_ctx* c) {
	if (c->reduce_exclusive != 1) return 0;
	const int dyn = c->lds_bytes - 16 * 1024 - 256;
	if (!c->attr_done[K_REDUCE]) {
		if (hipFuncSetAttribute(reinterpret_cast<const void*>(&reduce_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) return 0;
		c->attr_done[K_REDUCE] = true;
	}
	return dyn;
}

// dW / db of one Linear layer from dz (rows (foot, v)) and its input x (or, for the Fourier layer, the positions).  The partial
// tiles go out on stream s; the slab reduce follows on s, or -- reduce_side >= 0 -- on that side stream of the fork, ordered behind s.
static int weight_grad(find_ctx* c, Fork* fk, const float* dz, const float* x, int64_t x_foot_stride, const float* pos, int64_t pos_foot_stride,
					   const find_mlp_params* p, int nkt, int64_t feet, int64_t V, const BwdWs& b, float* dw, int ld_out,
					   int k_valid, int pe_map, float* db, float* S, hipStream_t s, int s_side = -1, int reduce_side = -1, bool h16 = false,
					   const float* vx_bias = nullptr, int64_t vx_bias_stride = 0) {   // (vx_bias: bcast_fold -- x is the shared product P, the operand relu(P[v] + vx_bias[foot]))
	FIND_REQUIRE(!vx_bias || (!pos && ((h16 && c->f16) || (!c->f16 && c->x3 && cdiv(V, 32) * feet >= c->gemm6_min_units))),
				 "weight_grad: a virtual operand reached a kernel that cannot form it (bcast_fold and the kernel selection disagree)");
	auto reduce_stream = [&]() -> hipStream_t {
		if (!fk || !fk->on || reduce_side < 0 || s_side < 0 || reduce_side == s_side) return s;
		fk->chain(s_side, reduce_side);
		return fk->stream(reduce_side);
	};
	float* pbuf = (db || S) ? b.pb : nullptr;
	const int cus = (fk && fk->on) ? c->side_cus : c->num_cus;   // (the side streams may be confined to a part of the chip: cu_reserve)
	if (!pos) {
		int nmain, spf;
		if (c->f16) {
			// opt-in fp16 mode: 64-row chunks, rows past the end of a foot zero-filled by the kernel
			const int cpf64 = (int)cdiv(V, 64);
			const int want = (int)std::max<int64_t>(1, std::min<int64_t>(cpf64, cdiv(cus, feet)));
			const int cps3 = (int)std::max<int64_t>(cdiv(cpf64, want), std::min<int>(4, cpf64));  // at least 256 rows per slab (the small launches are slab-bound)
			spf = (int)cdiv(cpf64, cps3);
			nmain = (int)(feet * spf);
			int lds = 0;
			if (vx_bias) FIND_TRY(prepare_kernel(c, K_DW3_V, &dw3_h16v_kernel, DW3_LDS, &lds));
			else if (h16) FIND_TRY(prepare_kernel(c, K_DW3_H, &dw3_h16_kernel, DW3_LDS, &lds));
			else FIND_TRY(prepare_kernel(c, K_DW3, &dw3_kernel, DW3_LDS, &lds));
			Dw3Args d3;
			memset(&d3, 0, sizeof(d3));
			d3.dz = dz; d3.dz_foot_stride = V * W; d3.x = x; d3.x_foot_stride = x_foot_stride;
			d3.V = (int)V; d3.chunks_per_foot = cpf64; d3.spf = spf; d3.cps = cps3; d3.pw = b.pw; d3.pb = pbuf;
			d3.xbias = vx_bias; d3.xbias_stride = vx_bias_stride;
			if (vx_bias) hipLaunchKernelGGL(dw3_h16v_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d3);   // (x formed from the shared product: bcast_fold)
			else if (h16) hipLaunchKernelGGL(dw3_h16_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d3);   // (dz and x fp16-stored: act16)
			else hipLaunchKernelGGL(dw3_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d3);
			FIND_LAUNCH_CHECK("dw3_kernel");
		} else if (c->x3 && cdiv(V, 32) * feet >= c->gemm6_min_units) {
			// bf16x3: 16-row chunks, rows past the end of a foot zero-filled by the kernel; few, long runs (slab traffic)
			const int cpf16 = (int)cdiv(V, 16);
			// Workgroups = slabs: one per CU when the launch has the chip to itself; HALF that inside a backward with side streams -- there it
			// runs beside the dX GEMMs of the next layer, which take the other CUs anyway, and half the slabs are half the reduce's traffic
			// (64 -> 32 MB per layer: headline step 1.76 -> 1.71 ms on one box; 96 and 64 workgroups are slower again)
			const int wgs6 = c->dw6_wgs > 0 ? c->dw6_wgs : ((fk && fk->on) ? std::max(1, c->num_cus / 2) : c->num_cus);
			const int want = (int)std::max<int64_t>(1, std::min<int64_t>(cpf16, cdiv(wgs6, feet)));
			const int cps6 = (int)std::max<int64_t>(cdiv(cpf16, want), std::min<int>(c->dw2_min_cps, cpf16));
			spf = (int)cdiv(cpf16, cps6);
			nmain = (int)(feet * spf);
			int lds = 0;
			if (vx_bias) FIND_TRY(prepare_kernel(c, K_DW6_V, &dw6v_kernel, DW6_LDS, &lds));
			else FIND_TRY(prepare_kernel(c, K_DW6, &dw6_kernel, DW6_LDS, &lds));
			Dw3Args d6;
			memset(&d6, 0, sizeof(d6));
			d6.dz = dz; d6.dz_foot_stride = V * W; d6.x = x; d6.x_foot_stride = x_foot_stride;
			d6.V = (int)V; d6.chunks_per_foot = cpf16; d6.spf = spf; d6.cps = cps6; d6.pw = b.pw; d6.pb = pbuf;
			d6.xbias = vx_bias; d6.xbias_stride = vx_bias_stride;
			if (vx_bias) hipLaunchKernelGGL(dw6v_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d6);   // (x formed from the shared product: bcast_fold)
			else hipLaunchKernelGGL(dw6_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d6);
			FIND_LAUNCH_CHECK("dw6_kernel");
		} else {
			// LDS-DMA kernel: every foot's rows cut into spf contiguous runs of 16-row chunks, the <= 15 leftover rows
			// folded into the foot's last run
			const int cpf16 = (int)(V / 16);
			int cps2 = 1;
			spf = 1;
			if (cpf16 > 0) {
				const int want = (int)std::max<int64_t>(1, std::min<int64_t>(cpf16, cdiv(cus, feet)));
				// few, long runs (slab traffic) -- unless that leaves CUs without a workgroup: a launch of few rows (the head's first-layer gradient
				// over the foot-summed dZ of a shared template: 6890 rows = 54 runs of 8 chunks, 86 us) is bounded by its longest run, not by slabs
				const int floor_cps = (feet * cpf16 < (int64_t)c->dw2_min_cps * cus) ? std::min(2, c->dw2_min_cps) : c->dw2_min_cps;
				cps2 = (int)std::max<int64_t>(cdiv(cpf16, want), std::min<int>(floor_cps, cpf16));
				spf = (int)cdiv(cpf16, cps2);
			}
			nmain = (int)(feet * spf);
			int lds = 0;
			if (c->dw_lds_free == 0) FIND_TRY(prepare_kernel(c, K_DW2, &dw2_kernel, DW2_LDS, &lds));
#ifdef FIND_DIAG
			if (c->dw_lds_free == 3) FIND_TRY(prepare_kernel(c, K_DW2_REPRO, &dw2_repro_kernel, DW2_LDS, &lds));
#endif
			Dw2Args d2;
			memset(&d2, 0, sizeof(d2));
			d2.dz = dz; d2.dz_foot_stride = V * W; d2.x = x; d2.x_foot_stride = x_foot_stride;
			d2.chunks_per_foot = cpf16; d2.tail_rows = (int)(V % 16); d2.spf = spf; d2.cps = cps2; d2.pw = b.pw; d2.pb = pbuf;
			d2.dbg = FIND_DBG(c->dw2_verify);   // diagnosis: verify every published ring stage (tools/probe_lds_fault.py)
#ifdef FIND_DIAG   // the reproducers of the co-residence fault
			if (c->dw_lds_free == 3) hipLaunchKernelGGL(dw2_repro_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d2);
			else if (c->dw_lds_free == 2) hipLaunchKernelGGL(dw4_wide_kernel, dim3((unsigned)nmain), dim3(256), 0, s, d2);
			else
#endif
			if (c->dw_lds_free == 1) hipLaunchKernelGGL(dw4_kernel, dim3((unsigned)nmain), dim3(512), 0, s, d2);
			else hipLaunchKernelGGL(dw2_kernel, dim3((unsigned)nmain), dim3(256), lds, s, d2);
			FIND_LAUNCH_CHECK("dw2_kernel");
		}
		ReduceWArgs r;
		memset(&r, 0, sizeof(r));
		r.pw = b.pw; r.nsplit = nmain; r.Kp = 256; r.out = dw; r.ld_out = ld_out; r.K_valid = k_valid;
		r.pb = pbuf; r.n_feet = (int)feet; r.spf = spf; r.db = db; r.S = S;
		r.nwblk = 256 * 256 / 4 / 64;
		if (c->reduce_exclusive == 2) hipLaunchKernelGGL(reduce_w_nolds_kernel, dim3((unsigned)(r.nwblk + (pbuf ? (int)feet + 1 : 0))), dim3(1024), 0, reduce_stream(), r);
		else hipLaunchKernelGGL(reduce_w_kernel, dim3((unsigned)(r.nwblk + (pbuf ? (int)feet + 1 : 0))), dim3(1024), reduce_lds(c), reduce_stream(), r);
		FIND_LAUNCH_CHECK("reduce_w_kernel");
		return FIND_OK;
	}
	// Fourier layer: the inputs are regenerated from the positions.  dwpe_kernel (mlp_dwpe.h; pe >= 32): one k-tile of workgroups per 128
	// frequencies, slabs of 2 pe + 32 columns; dw_kernel<AMODE_PE> (round 1, "dw_pe_lds_free" = 0): nkt k-tiles, slabs of 256 nkt columns.
	// Either way the launch stays within one round of workgroups over the chip.
	const bool lds_free = c->dw_pe_lds_free != 0 && p->pe_size >= 32;
	const int nkt_launch = lds_free ? (int)cdiv(p->pe_size, 128) : nkt;
	int spf, cps;
	split_policy(feet, V, &spf, &cps, std::min<int64_t>(128, std::max<int64_t>(16, std::min(c->dw_pe_target, cus) / nkt_launch)));
	DwArgs a;
	memset(&a, 0, sizeof(a));
	a.dz = dz; a.dz_foot_stride = V * W;
	a.x = x; a.x_foot_stride = x_foot_stride; a.ldx = W;
	a.pos = pos; a.pos_foot_stride = pos_foot_stride; a.Bm = p->B; a.pe = p->pe_size;
	a.V = (int)V; a.spf = spf; a.cps = cps; a.Kp = lds_free ? 2 * p->pe_size + 32 : nkt * 256;
	a.pw = b.pw; a.pb = pbuf;
	a.all_blocks = (c->ablate & 32) ? 1 : 0;
	const int nsplit = (int)(feet * spf);
	if (lds_free && (c->x3 || c->f16) && c->dwpe6) {   // bf16x3 calls (and the opt-in fp16 mode, whose Fourier layer keeps fp32-class arithmetic): the sin / cos columns on the bf16 matrix pipe, the x, y, z columns and the bias sums beside them
		hipLaunchKernelGGL(dwpe6_kernel, dim3((unsigned)nkt_launch, (unsigned)nsplit), dim3(512), 0, s, a);
		hipLaunchKernelGGL(dwxyz_kernel, dim3((unsigned)nsplit), dim3(1024), 0, s, a);
	}
	else if (lds_free) hipLaunchKernelGGL(dwpe_kernel, dim3((unsigned)nkt_launch, (unsigned)nsplit), dim3(512), 0, s, a);
	else hipLaunchKernelGGL((dw_kernel<AMODE_PE>), dim3((unsigned)nkt, (unsigned)nsplit), dim3(512), 0, s, a);
	FIND_LAUNCH_CHECK("dw_kernel");
	ReduceWArgs r;
	memset(&r, 0, sizeof(r));
	r.pw = b.pw; r.nsplit = nsplit; r.Kp = a.Kp; r.out = dw; r.ld_out = ld_out; r.K_valid = k_valid;
	r.pe_map = pe_map; r.pe = p->pe_size; r.in_dim = p->in_dim;
	r.pb = a.pb; r.n_feet = (int)feet; r.spf = spf; r.db = db; r.S = S;
	r.nwblk = 256 * a.Kp / 4 / 64;
	if (c->reduce_exclusive == 2) hipLaunchKernelGGL(reduce_w_nolds_kernel, dim3((unsigned)(r.nwblk + (a.pb ? (int)feet + 1 : 0))), dim3(1024), 0, reduce_stream(), r);
	else hipLaunchKernelGGL(reduce_w_kernel, dim3((unsigned)(r.nwblk + (a.pb ? (int)feet + 1 : 0))), dim3(1024), reduce_lds(c), reduce_stream(), r);
	FIND_LAUNCH_CHECK("reduce_w_kernel");
	return FIND_OK;
}

// Grouped weight gradients of a small call: every 256 x 256 layer's dW / db (and per-foot column sums) in ONE dw2 launch + ONE slab reduce.
struct WgradGroup {
	Dw2Group d;
	Dw6Group d6;        // the same jobs for dw6_group_kernel (bf16x3 calls)
	bool x3 = false;
	ReduceWGroup r;
	int n = 0, nmain = 0;
	int live_jobs = 0;   // jobs this launch will really carry (a call that skips a head has fewer than the workspace reserves): what the geometry is sized for
	int64_t feet = 0;
	WgradGroup() { memset(&d, 0, sizeof(d)); memset(&d6, 0, sizeof(d6)); memset(&r, 0, sizeof(r)); }
};

// Splits per foot of a grouped weight gradient.  A workgroup owns a whole 256 x 256 tile over `cps` 16-row chunks (3.4 us of MFMA issue
// each at the fp32 peak) and writes a 256-KB slab that the grouped reduce reads back; one workgroup per CU (whole-LDS reservation), so the
// launch runs in ceil(workgroups / CUs) rounds.  Few long splits leave CUs idle or pay a whole round for a few leftover workgroups; many
// short ones pay the per-workgroup prologue + slab and make the reduce (~0.06 us per slab) the longer kernel: take the cheapest of
// the candidates under this model ("group_spf" knob: 0 = model, n = force n splits per foot; measured with tools/fused_ab.py).
static void group_geometry(const find_ctx* c, int cpf16, int64_t feet, int jobs, int64_t max_slabs, int* spf_out, int* cps_out) {
	auto take = [&](int s) { *cps_out = (int)cdiv(cpf16, s); *spf_out = (int)cdiv(cpf16, *cps_out); };
	const int s_max = (int)std::max<int64_t>(1, std::min<int64_t>(cpf16, max_slabs / std::max<int64_t>(feet, 1)));
	if (c->group_spf > 0) { take(std::min(c->group_spf, s_max)); return; }
	double best = 1e30;
	take(1);
	for (int s = 1; s <= s_max; ++s) {
		const int cps = (int)cdiv(cpf16, s);
		if ((int)cdiv(cpf16, cps) != s || (cps < 2 && s > 1)) continue;   // (same geometry as a smaller s; single-chunk splits)
		const int64_t wgs = feet * s * std::max(jobs, 1);
		const double cost = (double)cdiv(wgs, c->side_cus) * (cps * 3.6 + 6.0) + wgs * 0.06;
		if (cost < best) { best = cost; take(s); }
	}
}

static int wgrad_group_add(find_ctx* c, WgradGroup& G, const BwdWs& b, const float* dz, const float* x, int64_t x_foot_stride, int64_t feet, int64_t V,
						   float* dw, int ld_out, float* db, float* S) {
	FIND_REQUIRE(G.n < b.grp_jobs && G.n < DW2_MAX_JOBS && G.n < DW6_MAX_JOBS, "find_mlp_bwd: too many grouped weight gradients");
	// bf16x3 calls: dw6_group_kernel (16-row chunks, rows past a foot's end zero-filled by the loads: ceil); the others dw4_group (tail rows folded)
	G.x3 = c->x3 && c->dw6_group && c->dw_lds_free == 1;
	const int cpf16 = G.x3 ? (int)cdiv(V, 16) : (int)(V / 16);
	int spf = 1, cps2 = 1;
	if (cpf16 > 0) group_geometry(c, cpf16, feet, G.live_jobs > 0 ? G.live_jobs : b.grp_jobs, b.grp_slabs, &spf, &cps2);
	const int nmain = (int)(feet * spf);
	FIND_REQUIRE(nmain <= b.grp_slabs, "find_mlp_bwd: grouped weight gradient needs %d slabs, %lld reserved", nmain, (long long)b.grp_slabs);
	float* pw = b.grp_pw + (int64_t)G.n * b.grp_slabs * W * W;
	float* pb = b.grp_pw + (int64_t)b.grp_jobs * b.grp_slabs * W * W + (int64_t)G.n * b.grp_slabs * W;
	Dw2Args& d2 = G.d.job[G.n];
	d2.dz = dz; d2.dz_foot_stride = V * W; d2.x = x; d2.x_foot_stride = x_foot_stride;
	d2.chunks_per_foot = cpf16; d2.tail_rows = (int)(V % 16); d2.spf = spf; d2.cps = cps2; d2.pw = pw; d2.pb = (db || S) ? pb : nullptr;
	Dw3Args& d6 = G.d6.job[G.n];
	d6.dz = dz; d6.dz_foot_stride = V * W; d6.x = x; d6.x_foot_stride = x_foot_stride;
	d6.V = (int)V; d6.chunks_per_foot = cpf16; d6.spf = spf; d6.cps = cps2; d6.pw = pw; d6.pb = d2.pb;
	ReduceWArgs& r = G.r.job[G.n];
	r.pw = pw; r.nsplit = nmain; r.Kp = 256; r.out = dw; r.ld_out = ld_out; r.K_valid = W;
	r.pb = d2.pb; r.n_feet = (int)feet; r.spf = spf; r.db = db; r.S = S;
	r.nwblk = 256 * 256 / 4 / 64;
	G.nmain = nmain; G.feet = feet;   // (every job of a group has the same geometry)
	G.n += 1;
	return FIND_OK;
}

static int wgrad_group_launch(find_ctx* c, WgradGroup& G, hipStream_t s) {
	if (G.n == 0) return FIND_OK;
	if (G.x3) {
		int lds = 0;
		FIND_TRY(prepare_kernel(c, K_DW6G, &dw6_group_kernel, DW6_LDS, &lds));
		hipLaunchKernelGGL(dw6_group_kernel, dim3((unsigned)G.nmain, (unsigned)G.n), dim3(256), lds, s, G.d6);
	} else if (c->dw_lds_free == 1 || c->dw_lds_free == 2) {
		hipLaunchKernelGGL(dw4_group_kernel, dim3((unsigned)G.nmain, (unsigned)G.n), dim3(512), 0, s, G.d);
	} else {
		int lds = 0;
		FIND_TRY(prepare_kernel(c, K_DW2G, &dw2_group_kernel, DW2_LDS, &lds));
		hipLaunchKernelGGL(dw2_group_kernel, dim3((unsigned)G.nmain, (unsigned)G.n), dim3(256), lds, s, G.d);
	}
	FIND_LAUNCH_CHECK("grouped weight-gradient kernel");
	hipLaunchKernelGGL(reduce_w_group_kernel, dim3((unsigned)(256 * 256 / 4 / 64 + (int)G.feet + 1), (unsigned)G.n), dim3(1024), 0, s, G.r);
	FIND_LAUNCH_CHECK("reduce_w_group_kernel");
	return FIND_OK;
}

// masked dX:  y = (dz @ W) * (mask > 0), with W given pre-transposed
// (vm_bias: bcast_fold -- mask is the shared fp32 product P and the layer's output was relu(P[v] + vm_bias[foot]))
static int linear_bwd_dx(find_ctx* c, const float* dz, const float* wt, int ldw, int w_tr, const float* mask, float* y, int64_t V, int64_t feet, hipStream_t s, bool h16 = false,
						 const float* vm_bias = nullptr, int64_t vm_bias_stride = 0, float* fs_out = nullptr, int64_t fs_slot_stride = 0, float* cs_out = nullptr) {
	GemmArgs a = gemm_args_zero();
	a.h16 = h16;
	a.fs_out = fs_out; a.fs_slot_stride = fs_slot_stride; a.cs_out = cs_out;   // (footsum_fold: y is then not written)
	a.a0 = dz; a.a_foot_stride = V * W; a.lda = W;
	a.w0 = wt; a.ldw = ldw; a.w_tr = w_tr; a.nchunk = W / KC;
	a.mask = mask; a.mask_foot_stride = vm_bias ? 0 : V * W;
	a.vm_bias = vm_bias; a.vm_bias_stride = vm_bias_stride;
	a.y = y; a.y_foot_stride = V * W; a.ldy = W; a.V = (int)V;
	return launch_gemm(c, AMODE_MAT, EPI_MASK, a, feet, s);
}

static int mlp_bwd_body(find_ctx* c, Fork& fk, const find_mlp_params* p, const Dims& d, const FwdWs& w, const BwdWs& b, const float* pos,
						const float* lat_disp, const float* lat_col, const float* d_disp, const float* d_col, const find_mlp_grads* g, bool defer) {
	hipStream_t s = fk.s;
	const int64_t V = d.V, n_feet = d.n_feet;
	const int ld_d0 = W + p->lat_disp, ld_c0 = W + p->lat_col;
	const int K0 = p->in_dim + 2 * p->pe_size;
	const bool act_d = d_disp != nullptr, act_c = d_col != nullptr;
	// no weight-gradient buffer at all: the network is frozen (requires_grad False on every weight -- stage 3 of train.py refines the
	// latent codes only, train.py:217-224) and the call returns the latent gradients alone
	const bool frozen = g->trunk_w[0] == nullptr;
	// side streams of the fork: Q carries the large head layers' weight gradients, T1 / T2 the first head layers (foot-summed, with
	// their column-sum reduce and latent gradients) and the trunk layers, R the slab reduces of the large head layers
	enum { Q = 0, T1 = 1, T2 = 2, R = 3 };

	// gradients of skipped parts are exact zeros: one launch for all of them (they were up to nine memsets in front of the texture pass's
	// backward, ~5 us each on the step's critical path)
	ZeroArgs za;
	memset(&za, 0, sizeof(za));
	int mrc = FIND_OK;
	auto zero = [&](float* ptr, int64_t n) {
		if (!ptr || n <= 0) return;
		if (za.njobs == ZERO_MAX_JOBS) { hipLaunchKernelGGL(zero_many_kernel, dim3(32, za.njobs), dim3(256), 0, s, za); za.njobs = 0; }
		za.p[za.njobs] = ptr; za.n[za.njobs] = n; za.njobs += 1;
	};
	auto zero_flush = [&]() {
		if (za.njobs > 0) hipLaunchKernelGGL(zero_many_kernel, dim3(32, za.njobs), dim3(256), 0, s, za);
		za.njobs = 0;
		if (hipGetLastError() != hipSuccess) { set_error("find_mlp_bwd: zero_many_kernel launch failed"); mrc = FIND_ELAUNCH; }
	};
	if (!act_d) {
		zero(g->disp_w[0], (int64_t)W * ld_d0); zero(g->disp_b[0], W);
		for (int i = 1; i < p->n_disp; ++i) { zero(g->disp_w[i], (int64_t)W * W); zero(g->disp_b[i], W); }
		zero(g->disp_w[p->n_disp], 3 * W); zero(g->disp_b[p->n_disp], 3);
		if (p->lat_disp) zero(g->lat_disp, n_feet * p->lat_disp);
	}
	if (!act_c) {
		zero(g->col_w[0], (int64_t)W * ld_c0); zero(g->col_b[0], W);
		for (int i = 1; i < p->n_col; ++i) { zero(g->col_w[i], (int64_t)W * W); zero(g->col_b[i], W); }
		zero(g->col_w[p->n_col], 3 * W); zero(g->col_b[p->n_col], 3);
		if (p->lat_col) zero(g->lat_col, n_feet * p->lat_col);
	}
	if (!act_d && !act_c) {
		zero(g->trunk_w[0], (int64_t)W * K0); zero(g->trunk_b[0], W);
		for (int i = 1; i < p->n_trunk; ++i) { zero(g->trunk_w[i], (int64_t)W * W); zero(g->trunk_b[i], W); }
		zero_flush();
		return mrc;
	}
	zero_flush();
	if (mrc != FIND_OK) return mrc;

	// 1. transposed weights for the dX GEMMs (a frozen network's backward stops at the heads' first layers: only their later layers) --
	// where the kernel that will run the layer cannot read the model's weight itself: bf16x3 chains (split_w_kernel, FusedStep::wmode 1) and
	// gemm7 (Gemm2Args::w_tr) can, so a bf16x3 training step transposes nothing (round 6: this launch was 10 - 12 us in front of each of the
	// step's two MLP backward passes)
	const bool fused_b = use_fused(c, V, d.feet_t) && p->pe_size > 0;
	const bool small_chain = fused_b && !d.shared;                 // the whole dX chain is one fused launch (also the frozen network's head chains)
	const bool t_heads = !(small_chain ? chain_direct(c) : gemm7_direct(c, V, n_feet));                 // layers >= 1 of the heads
	const bool t_trunk = !(fused_b ? chain_direct(c) : gemm7_direct(c, V, d.feet_t));                   // trunk layers >= 1
	const bool t_first = !(fused_b && chain_direct(c));                                                  // the heads' first layers (two-operand sum: gemm3 outside a chain)
	struct WT { const float* w; int ld; int tr; };
	auto wt_of = [&](bool need_t, const float* orig, int ld_orig, const float* transposed) -> WT { return need_t ? WT{transposed, W, 0} : WT{orig, ld_orig, 1}; };
	auto wt_D = [&](int l) -> WT { return l == 0 ? wt_of(t_first, p->disp_w[0], ld_d0, b.Dt[0]) : wt_of(t_heads, p->disp_w[l], W, b.Dt[l]); };
	auto wt_C = [&](int l) -> WT { return l == 0 ? wt_of(t_first, p->col_w[0], ld_c0, b.Ct[0]) : wt_of(t_heads, p->col_w[l], W, b.Ct[l]); };
	auto wt_T = [&](int l) -> WT { return wt_of(t_trunk, p->trunk_w[l], W, b.Tt[l]); };
	{
		RepackArgs ra;
		memset(&ra, 0, sizeof(ra));
		int n = 0;
		if (!frozen) {
			if (t_trunk) for (int i = 1; i < p->n_trunk; ++i) ra.job[n++] = RepackJob{p->trunk_w[i], b.Tt[i], W, W, W, 0, W, 1, 0, 0};
			if (t_first) {
				ra.job[n++] = RepackJob{p->disp_w[0], b.Dt[0], W, W, ld_d0, 0, W, 1, 0, 0};
				ra.job[n++] = RepackJob{p->col_w[0], b.Ct[0], W, W, ld_c0, 0, W, 1, 0, 0};
			}
		}
		if (t_heads) {
			for (int i = 1; i < p->n_disp; ++i) ra.job[n++] = RepackJob{p->disp_w[i], b.Dt[i], W, W, W, 0, W, 1, 0, 0};
			for (int i = 1; i < p->n_col; ++i) ra.job[n++] = RepackJob{p->col_w[i], b.Ct[i], W, W, W, 0, W, 1, 0, 0};
		}
		ra.njobs = n;
		if (n > 0) hipLaunchKernelGGL(repack_kernel, dim3(64, n), dim3(256), 0, s, ra);
		FIND_LAUNCH_CHECK("repack_kernel(T)");
	}

	// 2. final layers: dz of the last hidden layer of each head + dW/db of the 3-wide layers
	const bool a16_rule = use_act16(c, c->f16, d.shared, n_feet, V);
	const int note16 = noted_act16(c, w.fbd, a16_rule, use_fold(c, a16_rule, d.shared, n_feet, V, p));
	const bool a16 = (note16 & 1) != 0;   // (the forward stored w.D / w.C as fp16: b.dzD / b.dzC follow)
	const bool fold = (note16 & 2) != 0;  // (... and did not store w.D[0] / w.C[0] at all: their readers form them from w.hp / w.hp2 and the bias rows)
	struct Virt { const float* P; const float* bias; int64_t bstride; };
	auto virt = [&](bool colour, int l) -> Virt {   // the input of head layer l (l >= 1): virtual for l == 1 under bcast_fold
		if (!fold || l != 1) return Virt{nullptr, nullptr, 0};
		if (colour) return (p->lat_col > 0) ? Virt{w.hp2, w.fbc, W} : Virt{w.hp2, p->col_b[0], 0};
		return (p->lat_disp > 0) ? Virt{w.hp, w.fbd, W} : Virt{w.hp, p->disp_b[0], 0};
	};
	int cd = 0, cc = 0;  // current dZ buffer per head
	{
		HeadOutBwdArgs h;
		memset(&h, 0, sizeof(h));
		h.y[0] = w.D[p->n_disp - 1]; h.y[1] = w.C[p->n_col - 1];
		h.w[0] = p->disp_w[p->n_disp]; h.w[1] = p->col_w[p->n_col];
		h.z[0] = w.zd; h.z[1] = w.zc;
		h.gout[0] = d_disp; h.gout[1] = d_col;
		h.dy[0] = b.dzD[cd]; h.dy[1] = b.dzC[cc];
		h.pw[0] = b.pwo[0]; h.pw[1] = b.pwo[1];
		h.pb[0] = b.pbo[0]; h.pb[1] = b.pbo[1];
		h.rows = d.rows_h;
		h.half = a16 ? 1 : 0;
		hipLaunchKernelGGL(head_out_bwd_kernel, dim3((unsigned)b.nblk_out, 2), dim3(256), 0, s, h);
		HeadOutReduceArgs hr;
		memset(&hr, 0, sizeof(hr));
		hr.pw[0] = b.pwo[0]; hr.pw[1] = b.pwo[1]; hr.pb[0] = b.pbo[0]; hr.pb[1] = b.pbo[1];
		hr.dw[0] = act_d ? g->disp_w[p->n_disp] : nullptr; hr.db[0] = g->disp_b[p->n_disp];
		hr.dw[1] = act_c ? g->col_w[p->n_col] : nullptr; hr.db[1] = g->col_b[p->n_col];
		hr.nblk = b.nblk_out;
		// (the 3-wide layers' dW / db feed nothing of the dX chain: their reduce goes to a side stream like every other weight gradient --
		// it used to sit between head_out_bwd and the first dX GEMM of both backward passes of a step, 10 - 14 us each)
		if (!frozen) {
			fk.fork_to(T2);
			hipLaunchKernelGGL(head_out_reduce_kernel, dim3(12, 2), dim3(1024), 0, fk.stream(T2), hr);
		}
		FIND_LAUNCH_CHECK("head_out_bwd");
	}

	const float* hl = w.H[p->n_trunk - 1];
	const int64_t hl_stride = d.shared ? 0 : V * W;

	const bool fused = use_fused(c, V, d.feet_t) && p->pe_size > 0;
	// The trunk's dX chain of a call whose heads are large (the shared template of a batch: launched far below, behind the heads' GEMMs) is
	// put together HERE, so that its weights are split (chain_prepare) before the large kernels fill the chip.
	Chain tch;
	if (fused && !frozen && d.shared) {
		WT Wt[2]; int nb = 0;
		if (act_d) Wt[nb++] = wt_D(0);
		if (act_c) Wt[nb++] = wt_C(0);
		for (int i = 0; i < nb; ++i) {
			FusedStep& st = tch.gemm(Wt[i].w, Wt[i].ld, W / KC, hl /* patched below: the head's foot-summed first-layer dZ */);
			st.wmode = (unsigned char)Wt[i].tr;
			st.accum = i > 0;
			if (i + 1 < nb) { st.keep = 1; continue; }
			st.mask = 1; st.aux = hl; st.dst = b.dzT[0]; st.to_lds = 1;
		}
		for (int l = p->n_trunk - 1; l >= 1; --l) {
			const WT t = wt_T(l);
			FusedStep& st = tch.gemm(t.w, t.ld, W / KC);
			st.wmode = (unsigned char)t.tr;
			st.mask = 1; st.aux = w.H[l - 1]; st.dst = b.dzT[p->n_trunk - l]; st.to_lds = 1;
		}
		FIND_TRY(chain_prepare(c, tch, V, d.feet_t, s, b.w6, chain_w6_bytes(p)));
	}
	if (frozen) {
		// ---- latents only.  d loss / d latent[foot] = (sum_v dZ0[foot, v]) . W0[:, 256:]: the heads' dX chains down to their first layers,
		// per-foot column sums, one small product per head.  Nothing reaches the trunk, no weight gradient is formed: at the reference's
		// batch size this backward is 4 of the 9 layer-steps of the full chain and none of its thirteen weight-gradient jobs.
		const bool want_d = act_d && p->lat_disp > 0 && g->lat_disp != nullptr, want_c = act_c && p->lat_col > 0 && g->lat_col != nullptr;
		if (!want_d && !want_c) return FIND_OK;
		if (fused && !d.shared) {
			Chain ch;
			int nsteps = 0;
			auto head_chain = [&](int nl, float* const* act, float* const* dzbuf, bool colour, int& cur) {
				for (int l = nl - 1; l >= 1; --l) {
					const WT t = colour ? wt_C(l) : wt_D(l);
					FusedStep& st = ch.gemm(t.w, t.ld, W / KC, cur == 0 ? dzbuf[0] : nullptr);
					st.wmode = (unsigned char)t.tr;
					st.mask = 1; st.aux = act[l - 1]; st.dst = dzbuf[cur + 1]; st.to_lds = 1;
					cur += 1; nsteps += 1;
				}
			};
			if (want_c) head_chain(p->n_col, w.C, b.dzC, true, cc);
			if (want_d) head_chain(p->n_disp, w.D, b.dzD, false, cd);
			if (nsteps > 0) FIND_TRY(launch_chain(c, ch, V, d.feet_t, s, b.w6, chain_w6_bytes(p)));
		} else {
			if (want_d) for (int l = p->n_disp - 1; l >= 1; --l) { const WT t = wt_D(l); const Virt v = virt(false, l); FIND_TRY(linear_bwd_dx(c, b.dzD[cd], t.w, t.ld, t.tr, v.P ? v.P : w.D[l - 1], b.dzD[cd + 1], V, n_feet, s, a16, v.bias, v.bstride)); cd += 1; }
			if (want_c) for (int l = p->n_col - 1; l >= 1; --l) { const WT t = wt_C(l); const Virt v = virt(true, l); FIND_TRY(linear_bwd_dx(c, b.dzC[cc], t.w, t.ld, t.tr, v.P ? v.P : w.C[l - 1], b.dzC[cc + 1], V, n_feet, s, a16, v.bias, v.bstride)); cc += 1; }
		}
		struct Job { const float* dz; float* ps; float* S; const float* w0; int ld0; const float* lat; int L; float* glat; };
		const Job jobs[2] = {{b.dzD[cd], b.pS, b.Sd, p->disp_w[0], ld_d0, lat_disp, p->lat_disp, g->lat_disp},
							 {b.dzC[cc], b.pS2, b.Sc, p->col_w[0], ld_c0, lat_col, p->lat_col, g->lat_col}};
		for (int h = 0; h < 2; ++h) {
			if (!(h == 0 ? want_d : want_c)) continue;
			const Job& j = jobs[h];
			hipLaunchKernelGGL(footsum_kernel, dim3((unsigned)b.nblk_fs * 4), dim3(256), 0, s, j.dz, (int)n_feet, (int)V, (float*)nullptr, j.ps, a16 ? 1 : 0);
			hipLaunchKernelGGL(footsum_reduce_kernel, dim3((unsigned)n_feet, 4), dim3(1024), 0, s, j.ps, b.nblk_fs, (int)n_feet, j.S);
			hipLaunchKernelGGL(latent_grad_kernel, dim3((unsigned)n_feet), dim3(256), 0, s, j.w0, j.ld0, j.lat, j.L, j.S, (int)n_feet, j.glat,
							   (float*)nullptr, (float*)nullptr);
			FIND_LAUNCH_CHECK("latent gradients of a frozen network");
		}
		return FIND_OK;
	}
	if (fused && !d.shared) {
		// ---- small call (every layer has few rows: the reference's batch 1, the texture samples): the whole dX chain of both heads and
		// the trunk is ONE fused launch (mlp_fused.h); every layer's dZ lands in its own buffer, and the weight gradients follow on the
		// side streams (round-robin, one slab set per stream)
		Chain ch;
		int cd2 = 0, cc2 = 0;
		auto head_chain = [&](int nl, float* const* act, float* const* dzbuf, bool colour, int& cur) {
			for (int l = nl - 1; l >= 1; --l) {
				const WT t = colour ? wt_C(l) : wt_D(l);
				FusedStep& st = ch.gemm(t.w, t.ld, W / KC, cur == 0 ? dzbuf[0] : nullptr);
				st.wmode = (unsigned char)t.tr;
				st.mask = 1; st.aux = act[l - 1]; st.dst = dzbuf[cur + 1]; st.to_lds = 1;
				cur += 1;
			}
		};
		if (act_c) head_chain(p->n_col, w.C, b.dzC, true, cc2);
		if (act_d) head_chain(p->n_disp, w.D, b.dzD, false, cd2);
		{
			// gradient wrt the trunk output: both heads summed in one accumulator, then masked by the trunk's last activation
			const float* A[2]; WT Wt[2]; int nb = 0;
			if (act_d) { A[nb] = b.dzD[cd2]; Wt[nb] = wt_D(0); ++nb; }
			if (act_c) { A[nb] = b.dzC[cc2]; Wt[nb] = wt_C(0); ++nb; }
			for (int i = 0; i < nb; ++i) {
				FusedStep& st = ch.gemm(Wt[i].w, Wt[i].ld, W / KC, A[i]);
				st.wmode = (unsigned char)Wt[i].tr;
				st.accum = i > 0;
				if (i + 1 < nb) { st.keep = 1; continue; }
				st.mask = 1; st.aux = hl; st.dst = b.dzT[0]; st.to_lds = 1;
			}
		}
		int ct2 = 0;
		for (int l = p->n_trunk - 1; l >= 1; --l) {
			const WT t = wt_T(l);
			FusedStep& st = ch.gemm(t.w, t.ld, W / KC);
			st.wmode = (unsigned char)t.tr;
			st.mask = 1; st.aux = w.H[l - 1]; st.dst = b.dzT[ct2 + 1]; st.to_lds = 1;
			ct2 += 1;
		}
		FIND_TRY(launch_chain(c, ch, V, d.feet_t, s, b.w6, chain_w6_bytes(p)));
		// weight gradients: all inputs exist now.  fp32: every 256 x 256 layer in ONE grouped launch (+ one grouped slab reduce) on a side
		// stream, the Fourier layer on another; the opt-in fp16 mode keeps its per-layer dw3 launches.
		const bool grouped = !c->f16 && b.grp_pw != nullptr;
		WgradGroup G;
		G.live_jobs = (act_d ? p->n_disp : 0) + (act_c ? p->n_col : 0) + (p->n_trunk - 1);
		int rr = 0;
		auto next_side = [&](BwdWs* bk) -> int {
			const int k = fk.on ? 1 + rr % 2 : 0;
			rr += 1;
			*bk = b;
			bk->pw = b.pw_t[k]; bk->pb = b.pb_t[k];
			fk.fork_to(k);
			return k;
		};
		auto one = [&](const float* dz, const float* x, int64_t xs, int64_t feet, float* dw, int ld_out, float* db, float* S) -> int {
			if (grouped) return wgrad_group_add(c, G, b, dz, x, xs, feet, V, dw, ld_out, db, S);
			BwdWs bk; const int k = next_side(&bk);
			return weight_grad(c, &fk, dz, x, xs, nullptr, 0, p, 1, feet, V, bk, dw, ld_out, W, 0, db, S, fk.stream(k));
		};
		struct Lat { const float* w0full; int ld0; const float* lat; int L; float* S; float* glat; float* gw0; };
		Lat lats[2]; int nlat = 0;
		auto head_wgrads = [&](int nl, float* const* act, float* const* dzbuf, int cur_last, float* const* gw, float* const* gb, const float* w0full, int ld0,
							   const float* lat, int L, float* S, float* glat) -> int {
			for (int l = nl - 1; l >= 1; --l) FIND_TRY(one(dzbuf[nl - 1 - l], act[l - 1], V * W, n_feet, gw[l], W, gb[l], nullptr));
			if (defer && L > 0) {
				// the weight gradients stay behind on the side streams: the latent gradients -- an OUTPUT autograd hands to whatever comes
				// next -- are formed on the caller's stream, from per-foot column sums of their own (no slab reduce to wait for)
				float* ps = (gw == g->disp_w) ? b.pS : b.pS2;
				hipLaunchKernelGGL(footsum_kernel, dim3((unsigned)b.nblk_fs * 4), dim3(256), 0, s, dzbuf[cur_last], (int)n_feet, (int)V, (float*)nullptr, ps, 0);
				hipLaunchKernelGGL(footsum_reduce_kernel, dim3((unsigned)n_feet, 4), dim3(1024), 0, s, ps, b.nblk_fs, (int)n_feet, S);
				hipLaunchKernelGGL(latent_grad_kernel, dim3((unsigned)(n_feet + W)), dim3(256), 0, s, w0full, ld0, lat, L, S, (int)n_feet, glat, gw[0], (float*)nullptr);
				FIND_LAUNCH_CHECK("latent gradients (deferred join)");
			}
			FIND_TRY(one(dzbuf[cur_last], hl, hl_stride, n_feet, gw[0], ld0, gb[0], (L > 0 && !defer) ? S : nullptr));
			if (L > 0 && !defer) lats[nlat++] = Lat{w0full, ld0, lat, L, S, glat, gw[0]};
			return FIND_OK;
		};
		if (act_d) FIND_TRY(head_wgrads(p->n_disp, w.D, b.dzD, cd2, g->disp_w, g->disp_b, p->disp_w[0], ld_d0, lat_disp, p->lat_disp, b.Sd, g->lat_disp));
		if (act_c) FIND_TRY(head_wgrads(p->n_col, w.C, b.dzC, cc2, g->col_w, g->col_b, p->col_w[0], ld_c0, lat_col, p->lat_col, b.Sc, g->lat_col));
		for (int l = p->n_trunk - 1; l >= 1; --l)
			FIND_TRY(one(b.dzT[p->n_trunk - 1 - l], w.H[l - 1], V * W, d.feet_t, g->trunk_w[l], W, g->trunk_b[l], nullptr));
		// the latent gradients read the per-foot column sums S (written by the slab reduce): same stream, behind it
		hipStream_t sl = s;
		if (grouped) {
			fk.fork_to(1);
			sl = fk.stream(1);
			FIND_TRY(wgrad_group_launch(c, G, sl));
		} else if (fk.on) {
			// (per-layer launches went to streams 1 / 2 round-robin: wait for both before the latent gradients on stream 1)
			fk.chain(2, 1);
			sl = fk.stream(1);
		}
		for (int i = 0; i < nlat; ++i) {
			hipLaunchKernelGGL(latent_grad_kernel, dim3((unsigned)(n_feet + W)), dim3(256), 0, sl, lats[i].w0full, lats[i].ld0, lats[i].lat, lats[i].L, lats[i].S,
							   (int)n_feet, lats[i].glat, lats[i].gw0, (float*)nullptr);
			FIND_LAUNCH_CHECK("latent_grad_kernel");
		}
		{
			const int k = fk.on ? 2 : 0;
			fk.fork_to(k);
			FIND_TRY(weight_grad(c, &fk, b.dzT[ct2], nullptr, 0, pos, V * 3, p, d.nkt0, d.feet_t, V, b, g->trunk_w[0], K0, 0, 1, g->trunk_b[0], nullptr, fk.stream(k)));
		}
		FIND_LAUNCH_CHECK("find_mlp_bwd");
		return FIND_OK;
	}

	// 3. heads, last hidden layer down to the first.  Weight-gradient work (dW / db / latent gradients only feed the outputs, never
	// the dX chain) goes to the side streams; every layer's dZ has its own buffer, so the dX chain on the caller's stream never waits.
	int big_toggle = 0;
	hipEvent_t set_free[2] = {nullptr, nullptr};   // fires when the slab set's previous reduce (on R) has read it
	// The V-row weight gradients of a shared template's backward -- the trunk's layers and (round 6) the heads' first layers over the
	// foot-summed dZ, which used to be a 50-70-us fp32-MFMA launch of its own per head -- travel in ONE grouped launch behind the trunk's dX chain.
	WgradGroup G;
	const bool grouped_v = fused && d.shared && !c->f16 && b.grp_pw != nullptr;
	G.live_jobs = (p->n_trunk - 1) + (c->group_head0 ? (act_d ? 1 : 0) + (act_c ? 1 : 0) : 0);
	auto head_bwd = [&](int nl, float* const* act, float* const* dzbuf, int& cur, bool colour, float* const* gw, float* const* gb,
						const float* w0full, int ld0, const float* lat, int L, float* S, float* glat, float* zs, float* ps, int side) -> int {
		int fsum_pairs = 0;
		for (int l = nl - 1; l >= 1; --l) {
			const Virt v = virt(colour, l);
			const float* const xin = v.P ? v.P : act[l - 1];
			const int64_t xin_stride = v.P ? 0 : V * W;
			fk.fork_to(Q);
			// the large layers alternate between two slab sets and hand their slab reduce to stream R: the reduce (LDS-using, so
			// it only gets a CU when a ring kernel's workgroup retires) no longer sits between two dw2 launches on Q
			// (Not under stream capture: there the streams only express dependencies and the graph executor schedules the branches;
			// hipStreamEndCapture of ROCm 7.0 / 7.2 faults on this R <-> Q event pattern, and captures cleanly without it.)
			if (fk.on && c->reduce_stream && !fk.capturing) {
				const int si = big_toggle & 1;
				big_toggle += 1;
				BwdWs bk = b;
				bk.pw = b.pw_t[si ? 3 : 0]; bk.pb = b.pb_t[si ? 3 : 0];
				fk.wait(Q, set_free[si]);
				FIND_TRY(weight_grad(c, &fk, dzbuf[cur], xin, xin_stride, nullptr, 0, p, 1, n_feet, V, bk, gw[l], W, W, 0, gb[l], nullptr, fk.stream(Q), Q, R, a16, v.bias, v.bstride));
				set_free[si] = fk.mark(R);
			} else {
				FIND_TRY(weight_grad(c, &fk, dzbuf[cur], xin, xin_stride, nullptr, 0, p, 1, n_feet, V, b, gw[l], W, W, 0, gb[l], nullptr, fk.stream(Q), -1, -1, a16, v.bias, v.bstride));
			}
			const WT t = colour ? wt_C(l) : wt_D(l);
			// footsum_fold: this dX GEMM's output is the broadcast layer's dZ, of which only sums are read (below): formed inside the GEMM
			fsum_pairs = (l == 1 && v.P && !a16 && c->footsum_fold && zs != nullptr && gemm7_direct_units(c, V, n_feet)) ? gemm7_fsum_pairs(c, V, n_feet) : 0;
			if (fsum_pairs > 0) {
				float* const fs1 = colour ? b.fs1C : b.fs1D;
				FIND_TRY(linear_bwd_dx(c, dzbuf[cur], t.w, t.ld, t.tr, xin, dzbuf[cur + 1], V, n_feet, s, a16, v.bias, v.bstride, zs, fs1 - zs, ps));
			} else {
				FIND_TRY(linear_bwd_dx(c, dzbuf[cur], t.w, t.ld, t.tr, xin, dzbuf[cur + 1], V, n_feet, s, a16, v.bias, v.bstride));
			}
			cur += 1;
		}
		float* db_late = nullptr;
		hipStream_t q0;
		if (d.shared && fsum_pairs > 0) {
			// (the two partial foot sums -> the foot sum; the per-foot column sums wait in `ps`, one block per workgroup pair of the GEMM)
			const int64_t n4 = V * W / 4;
			hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, s, zs, colour ? b.fs1C : b.fs1D, n4);
			fk.fork_to(side);
			q0 = fk.stream(side);
			hipLaunchKernelGGL(footsum_reduce_kernel, dim3((unsigned)n_feet, 4), dim3(1024), 0, q0, ps, fsum_pairs, (int)n_feet, S);
			if (L > 0) db_late = gb[0];
			else hipLaunchKernelGGL(colsum_small_kernel, dim3(1), dim3(256), 0, q0, S, (int)n_feet, gb[0]);
			FIND_LAUNCH_CHECK("footsum (folded)");
			BwdWs bk = b;
			bk.pw = b.pw_t[side]; bk.pb = b.pb_t[side];
			if (grouped_v && c->group_head0) FIND_TRY(wgrad_group_add(c, G, b, zs, hl, 0, 1, V, gw[0], ld0, nullptr, nullptr));
			else FIND_TRY(weight_grad(c, &fk, zs, hl, 0, nullptr, 0, p, 1, 1, V, bk, gw[0], ld0, W, 0, nullptr, nullptr, q0));
		} else if (d.shared) {
			// every foot multiplies the same trunk rows: reduce dZ0 over feet first (one pass), then M = V GEMMs
			// (Measured and dropped: the foot sum on the head's side stream, the caller's stream going straight on to the other head's dX
			// chain and waiting for the sums in step 4 -- 3.49 against 3.38 ms per train_3d step: the HBM-bound pass beside the dX GEMMs
			// costs them more than the wait it removes.)
			hipLaunchKernelGGL(footsum_kernel, dim3((unsigned)b.nblk_fs * 4), dim3(256), 0, s, dzbuf[cur], (int)n_feet, (int)V, zs, ps, a16 ? 1 : 0);
			// the foot-summed first layer is a small launch: its own side stream and slab set, so that it does not queue behind the
			// large weight-gradient launches on Q.  The per-foot column sums go there too: only the latent / bias gradients read them.
			fk.fork_to(side);
			q0 = fk.stream(side);
			BwdWs bk = b;
			bk.pw = b.pw_t[side]; bk.pb = b.pb_t[side];
			hipLaunchKernelGGL(footsum_reduce_kernel, dim3((unsigned)n_feet, 4), dim3(1024), 0, q0, ps, b.nblk_fs, (int)n_feet, S);
			// (the bias gradient -- S summed over feet -- rides along with the latent-gradient launch when there is one)
			if (L > 0) db_late = gb[0];
			else hipLaunchKernelGGL(colsum_small_kernel, dim3(1), dim3(256), 0, q0, S, (int)n_feet, gb[0]);
			FIND_LAUNCH_CHECK("footsum");
			if (grouped_v && c->group_head0) FIND_TRY(wgrad_group_add(c, G, b, zs, hl, 0, 1, V, gw[0], ld0, nullptr, nullptr));
			else FIND_TRY(weight_grad(c, &fk, zs, hl, 0, nullptr, 0, p, 1, 1, V, bk, gw[0], ld0, W, 0, nullptr, nullptr, q0));
		} else {
			fk.fork_to(Q);
			q0 = fk.stream(Q);
			fk.wait(Q, set_free[0]);   // set 0 may still be read by a reduce on R
			FIND_TRY(weight_grad(c, &fk, dzbuf[cur], hl, hl_stride, nullptr, 0, p, 1, n_feet, V, b, gw[0], ld0, W, 0, gb[0], (L > 0) ? S : nullptr, q0));
		}
		if (L > 0) {
			hipLaunchKernelGGL(latent_grad_kernel, dim3((unsigned)(n_feet + W + (db_late ? 1 : 0))), dim3(256), 0, q0, w0full, ld0, lat, L, S, (int)n_feet, glat,
							   gw[0], db_late);
			FIND_LAUNCH_CHECK("latent_grad_kernel");
		}
		return FIND_OK;
	};
	if (act_d) FIND_TRY(head_bwd(p->n_disp, w.D, b.dzD, cd, false, g->disp_w, g->disp_b, p->disp_w[0], ld_d0, lat_disp, p->lat_disp, b.Sd, g->lat_disp, b.zsD, b.pS, T1));
	if (act_c) FIND_TRY(head_bwd(p->n_col, w.C, b.dzC, cc, true, g->col_w, g->col_b, p->col_w[0], ld_c0, lat_col, p->lat_col, b.Sc, g->lat_col, b.zsC, b.pS2, T2));

	// 4 + 5.  gradient wrt the trunk output -- both heads (and, for a shared trunk, every foot) summed in the K loop -- and the trunk's
	// dX chain.  With few trunk rows (the shared template) all of it is one fused launch; the weight gradients follow on T1 / T2.
	int ct = 0;
	if (fused) {
		Chain& ch = tch;   // (built, and its weights split, right behind the transposes: see there)
		for (int i = 0, k = 0; i < ch.a.n_steps && k < 2; ++i)   // the first steps read the heads' (foot-summed) first-layer dZ
			if (ch.a.step[i].src_kind == FS_SRC_GLOBAL) { ch.a.step[i].src = (k == 0 && act_d) ? (d.shared ? b.zsD : b.dzD[cd]) : (d.shared ? b.zsC : b.dzC[cc]); ++k; }
		FIND_TRY(launch_chain(c, ch, V, d.feet_t, s, b.w6, chain_w6_bytes(p)));
		// the trunk's weight gradients: all their inputs exist now -- one grouped launch + one grouped reduce (fp32), as in the small-call path
		const bool grouped = grouped_v;
		for (int l = p->n_trunk - 1; l >= 1; --l) {
			if (grouped) {
				FIND_TRY(wgrad_group_add(c, G, b, b.dzT[ct], w.H[l - 1], V * W, d.feet_t, V, g->trunk_w[l], W, g->trunk_b[l], nullptr));
			} else {
				const int k = fk.on ? 1 + (l & 1) : 0;
				BwdWs bk = b;
				bk.pw = b.pw_t[k]; bk.pb = b.pb_t[k];
				fk.fork_to(k);
				FIND_TRY(weight_grad(c, &fk, b.dzT[ct], w.H[l - 1], V * W, nullptr, 0, p, 1, d.feet_t, V, bk, g->trunk_w[l], W, W, 0, g->trunk_b[l], nullptr, fk.stream(k)));
			}
			ct += 1;
		}
		if (grouped) {
			fk.fork_to(T1);
			FIND_TRY(wgrad_group_launch(c, G, fk.stream(T1)));
		}
	} else {
	// 4. gradient wrt the trunk output: both heads (and, for a shared trunk, every foot) summed in the K loop
		{
			GemmArgs a = gemm_args_zero();
			const float* A[2]; const float* Wt[2]; int nb = 0;
			if (act_d) { A[nb] = d.shared ? b.zsD : b.dzD[cd]; Wt[nb] = b.Dt[0]; ++nb; }
			if (act_c) { A[nb] = d.shared ? b.zsC : b.dzC[cc]; Wt[nb] = b.Ct[0]; ++nb; }
			a.nbase = nb; a.a0 = A[0]; a.w0 = Wt[0];
			if (nb > 1) { a.a1 = A[1]; a.w1 = Wt[1]; }
			a.lda = W; a.ldw = W; a.nchunk = W / KC;
			a.a_foot_stride = d.shared ? 0 : V * W;  // shared: the foot-summed (V,256) matrices
			a.mask = hl; a.mask_foot_stride = V * W;
			a.y = b.dzT[ct]; a.y_foot_stride = V * W; a.ldy = W; a.V = (int)V;
			FIND_TRY(launch_gemm(c, AMODE_MAT, EPI_MASK, a, d.feet_t, s));
			FIND_LAUNCH_CHECK("trunk-out dX gemm");
		}

	// 5. trunk: the dX chain runs back to back on the caller's stream, the layers' weight gradients -- independent of each other,
		// each filling a fraction of the chip -- alternate between T1 and T2 (own slab set each); Q / set 0 keeps the Fourier layer
		for (int l = p->n_trunk - 1; l >= 1; --l) {
			const int k = fk.on ? 1 + (l & 1) : 0;
			BwdWs bk = b;
			bk.pw = b.pw_t[k]; bk.pb = b.pb_t[k];
			fk.fork_to(k);
			FIND_TRY(weight_grad(c, &fk, b.dzT[ct], w.H[l - 1], V * W, nullptr, 0, p, 1, d.feet_t, V, bk, g->trunk_w[l], W, W, 0, g->trunk_b[l], nullptr, fk.stream(k)));
			const WT t = wt_T(l);
			FIND_TRY(linear_bwd_dx(c, b.dzT[ct], t.w, t.ld, t.tr, w.H[l - 1], b.dzT[ct + 1], V, d.feet_t, s));
			ct += 1;
		}
	}
	if (fused && d.shared && fk.on && c->pe_on_t2) {
		// (round 6: on Q the Fourier layer's weight gradient queued behind the last large layer's dw6 AND its slab reduce -- 60 us after the
		// trunk's dX chain had produced its input, at the very end of the step; T2 is idle by then, and its slab set is the Fourier layer's size)
		fk.fork_to(T2);
		BwdWs bk = b;
		bk.pw = b.pw_t[T2]; bk.pb = b.pb_t[T2];
		FIND_TRY(weight_grad(c, &fk, b.dzT[ct], nullptr, 0, pos, V * 3, p, d.nkt0, d.feet_t, V, bk, g->trunk_w[0], K0, 0, 1, g->trunk_b[0], nullptr, fk.stream(T2)));
	} else {
		fk.fork_to(Q);
		fk.wait(Q, set_free[0]);
		FIND_TRY(weight_grad(c, &fk, b.dzT[ct], nullptr, 0, pos, V * 3, p, d.nkt0, d.feet_t, V, b, g->trunk_w[0], K0, 0, 1, g->trunk_b[0], nullptr, fk.stream(Q)));
	}
	FIND_LAUNCH_CHECK("find_mlp_bwd");
	return FIND_OK;
}

}  // namespace mlp
}  // namespace find

// Slabs of one weight-gradient launch: up to max(512, feet) + feet + 16 partial 256x256 tiles and as many 256-float bias rows.
static int64_t wgrad_slabs(int64_t n_feet) { return std::max<int64_t>(512, n_feet) + n_feet + 16; }

extern "C" int64_t find_linear_wgrad_scratch_bytes(int64_t n_feet) {
	if (n_feet < 1) return -1;
	return wgrad_slabs(n_feet) * ((int64_t)W * W + W) * (int64_t)sizeof(float);
}

extern "C" int find_linear_wgrad(find_ctx* c, const float* dz, const float* x, int64_t n_feet, int64_t n_pts, float* dw, float* db,
								 void* scratch, int64_t scratch_bytes, void* stream) {
	FIND_TRY(check_ctx(c, "find_linear_wgrad"));
	FIND_REQUIRE(dz && x && dw && scratch, "find_linear_wgrad: NULL argument");
	FIND_REQUIRE(n_feet >= 1 && n_pts >= 1 && n_feet < (1 << 16), "find_linear_wgrad: bad sizes");
	FIND_REQUIRE(aligned16(dz) && aligned16(x) && aligned16(scratch), "find_linear_wgrad: dz, x and scratch must be 16-byte aligned");
	if (scratch_bytes < find_linear_wgrad_scratch_bytes(n_feet)) {
		set_error("find_linear_wgrad: scratch too small (%lld < %lld)", (long long)scratch_bytes, (long long)find_linear_wgrad_scratch_bytes(n_feet));
		return FIND_EWORKSPACE;
	}
	BwdWs b;
	memset(&b, 0, sizeof(b));
	b.pw = static_cast<float*>(scratch);
	b.pb = b.pw + wgrad_slabs(n_feet) * W * W;
	c->f16 = c->mlp_f16 == 1;
	c->x3 = c->mlp_f16 == 2;
	FIND_TRY(weight_grad(c, nullptr, dz, x, n_pts * W, nullptr, 0, nullptr, 1, n_feet, n_pts, b, dw, W, W, 0, db, nullptr, reinterpret_cast<hipStream_t>(stream)));
	FIND_LAUNCH_CHECK("find_linear_wgrad");
	return FIND_OK;
}

extern "C" int64_t find_mlp_bwd_scratch_bytes(const find_mlp_params* p, int64_t pos_batch, int64_t n_feet, int64_t n_pts) {
	Dims d;
	if (make_dims(p, pos_batch, n_feet, n_pts, &d) != FIND_OK) return -1;
	BwdWs b;
	carve_bwd(p, d, nullptr, &b);
	return b.bytes;
}

extern "C" int find_mlp_bwd(find_ctx* c, const find_mlp_params* p, const float* pos, int64_t pos_batch, int64_t n_feet, int64_t n_pts,
							const float* lat_disp, const float* lat_col, const float* d_disp, const float* d_col,
							const void* ws, int64_t ws_bytes, void* scratch, int64_t scratch_bytes,
							const find_mlp_grads* g, void* stream) {
	FIND_TRY(check_ctx(c, "find_mlp_bwd"));
	Dims d;
	FIND_TRY(make_dims(p, pos_batch, n_feet, n_pts, &d));
	FIND_TRY(check_weights(p));
	FIND_REQUIRE(pos && ws && scratch && g, "find_mlp_bwd: NULL argument");
	FIND_REQUIRE(p->lat_disp == 0 ? lat_disp == nullptr : (lat_disp != nullptr || d_disp == nullptr), "find_mlp_bwd: lat_disp pointer does not match params");
	FIND_REQUIRE(p->lat_col == 0 ? lat_col == nullptr : (lat_col != nullptr || d_col == nullptr), "find_mlp_bwd: lat_col pointer does not match params");
	FIND_REQUIRE(p->precision >= 0 && p->precision <= 3, "find_mlp_bwd: params.precision must be 0, 1, 2 or 3 (got %d)", p->precision);
	{
		// weight-gradient buffers: all of them (a head without an upstream gradient may leave its own out: nothing is written for it), or
		// none at all (frozen network: latent gradients only)
		const bool frozen = g->trunk_w[0] == nullptr;
		bool ok = true;
		for (int i = 0; i < p->n_trunk; ++i) ok = ok && ((g->trunk_w[i] == nullptr) == frozen) && ((g->trunk_b[i] == nullptr) == frozen);
		for (int i = 0; i <= p->n_disp; ++i) ok = ok && (frozen ? !g->disp_w[i] && !g->disp_b[i] : (d_disp == nullptr || (g->disp_w[i] && g->disp_b[i])));
		for (int i = 0; i <= p->n_col; ++i) ok = ok && (frozen ? !g->col_w[i] && !g->col_b[i] : (d_col == nullptr || (g->col_w[i] && g->col_b[i])));
		FIND_REQUIRE(ok, "find_mlp_bwd: grads must name every weight-gradient buffer of the evaluated heads and the trunk, or none at all (frozen network)");
	}
	FwdWs w;
	carve_fwd(p, d, true, const_cast<void*>(ws), &w);
	if (ws_bytes < w.bytes) {
		set_error("find_mlp_bwd: forward workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
		return FIND_EWORKSPACE;
	}
	BwdWs b;
	carve_bwd(p, d, scratch, &b);
	if (scratch_bytes < b.bytes) {
		set_error("find_mlp_bwd: scratch too small (%lld < %lld)", (long long)scratch_bytes, (long long)b.bytes);
		return FIND_EWORKSPACE;
	}
	c->f16 = call_f16(c, p);
	c->x3 = call_x3(c, p);
	Fork fk(c, reinterpret_cast<hipStream_t>(stream), c->bwd_streams != 0);
	// "defer_join" (one call: the knob is taken down here): the weight gradients of a small per-foot call -- the texture pass of a
	// train_3d step, 0.28 ms of grouped and Fourier-layer weight gradients that nothing reads until the main pass adds its own -- keep
	// running on the side streams while the caller's stream goes on; find_ctx_join (or the join of the next call) waits for them.
	// The caller keeps scratch, workspace and gradient buffers alive until then.  Not under stream capture, not for the large-call paths.
	const bool defer = c->defer_join != 0 && fk.on && !fk.capturing && !d.shared && use_fused(c, d.V, d.feet_t) && p->pe_size > 0 && g->trunk_w[0] != nullptr;
	c->defer_join = 0;
	const int rc = mlp_bwd_body(c, fk, p, d, w, b, pos, lat_disp, lat_col, d_disp, d_col, g, defer);
	// join on EVERY path: the caller's stream continues only after every side stream this call touched has finished, so scratch,
	// workspace and gradient buffers may be freed (stream-ordered) as soon as the call returns -- also after an error
	const int rj = (defer && rc == FIND_OK) ? fk.defer() : fk.join();
	return rc != FIND_OK ? rc : rj;
}

// Make `stream` wait for the side-stream work an earlier find_mlp_bwd left running ("defer_join").  Returns 1 if there was any, 0 if not.
extern "C" int find_ctx_join(find_ctx* c, void* stream) {
	FIND_TRY(check_ctx(c, "find_ctx_join"));
	int any = 0;
	for (int k = 0; k < N_SIDE; ++k)
		if (c->pend[k]) {
			FIND_HIP_OK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->pend_ev[k], 0), "hipStreamWaitEvent");
			c->pend[k] = false;
			any = 1;
		}
	return any ? 1 : FIND_OK;
}

// ------------------------------------------------------------------------------------------- streams and hardware queues
// HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, four by default), in creation order; two streams on one queue
// run their launches in order.  Which of the context's side streams really run beside the caller's stream -- and beside each other --
// therefore depends on what the process created before: a probe launch tells.
namespace find {
namespace mlp {
__global__ void spin_kernel(long long ticks) {
	const long long t0 = wall_clock64();   // 100 MHz
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void nop_kernel() {}
}  // namespace mlp
}  // namespace find

// does a launch on b run while a is busy?  "Yes" cannot be wrong (on one queue b's launch cannot finish before a's spin); "no" can, when
// the host thread is held up between b's completion and the query for longer than the spin: a "no" is asked again, twice.
static int runs_beside(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb, bool* beside) {
	*beside = false;
	for (int attempt = 0; attempt < 3 && !*beside; ++attempt) {
		FIND_HIP_OK(hipStreamSynchronize(a), "hipStreamSynchronize");
		FIND_HIP_OK(hipStreamSynchronize(b), "hipStreamSynchronize");
		hipLaunchKernelGGL(find::mlp::spin_kernel, dim3(1), dim3(64), 0, a, 30000ll << attempt);   // ~0.3 ms, then 0.6, 1.2
		FIND_HIP_OK(hipEventRecord(ea, a), "hipEventRecord");
		hipLaunchKernelGGL(find::mlp::nop_kernel, dim3(1), dim3(64), 0, b);
		FIND_HIP_OK(hipEventRecord(eb, b), "hipEventRecord");
		FIND_HIP_OK(hipEventSynchronize(eb), "hipEventSynchronize");
		*beside = hipEventQuery(ea) == hipErrorNotReady;
		FIND_HIP_OK(hipStreamSynchronize(a), "hipStreamSynchronize");
	}
	return FIND_OK;
}

// groups[0] = 0 for the caller's stream, groups[1 + k] for side stream k: streams with the same number share a hardware queue
static int stream_groups(hipStream_t const* st, int n, hipEvent_t ea, hipEvent_t eb, int* groups) {
	int ngroups = 0;
	int rep[16];
	for (int i = 0; i < n; ++i) {
		groups[i] = -1;
		for (int g = 0; g < ngroups && groups[i] < 0; ++g) {
			bool beside = false;
			FIND_TRY(runs_beside(st[rep[g]], st[i], ea, eb, &beside));
			if (!beside) groups[i] = g;
		}
		if (groups[i] < 0) {
			if (ngroups == 16) { groups[i] = 15; continue; }
			rep[ngroups] = i;
			groups[i] = ngroups++;
		}
	}
	return FIND_OK;
}

extern "C" int find_ctx_stream_groups(find_ctx* c, void* caller_stream, int32_t* groups) {
	FIND_TRY(check_ctx(c, "find_ctx_stream_groups"));
	FIND_REQUIRE(groups != nullptr, "find_ctx_stream_groups: groups is NULL");
	hipStream_t st[1 + N_SIDE];
	st[0] = reinterpret_cast<hipStream_t>(caller_stream);
	for (int k = 0; k < N_SIDE; ++k) st[1 + k] = c->side[k];
	int g[1 + N_SIDE];
	FIND_TRY(stream_groups(st, 1 + N_SIDE, c->ev[0], c->ev[1], g));
	for (int k = 0; k < 1 + N_SIDE; ++k) groups[k] = g[k];
	return FIND_OK;
}

namespace find { namespace mlp { static int bind_side_streams(find_ctx* c, hipStream_t caller); } }

// A caller's second stream that fits the context's layout: the first of `cands` that runs BESIDE caller_stream and shares the hardware queue
// of side stream `role` (0 = Q: the large head layers' weight gradients -- busy only during the main pass's backward).  With four hardware
// queues a fifth stream always shares one; which one decides what its work waits behind (find_hip.h).  *index = -1: none of them does.
extern "C" int find_ctx_stream_beside(find_ctx* c, void* caller_stream, void* const* cands, int32_t n, int32_t role, int32_t* index) {
	FIND_TRY(check_ctx(c, "find_ctx_stream_beside"));
	FIND_REQUIRE(cands != nullptr && index != nullptr && n >= 0 && role >= 0 && role < N_SIDE, "find_ctx_stream_beside: bad arguments (role 0..%d)", N_SIDE - 1);
	hipStream_t caller = reinterpret_cast<hipStream_t>(caller_stream);
	if (!c->side_bound && c->bind_streams) (void)find::mlp::bind_side_streams(c, caller);
	*index = -1;
	for (int i = 0; i < n; ++i) {
		hipStream_t s = reinterpret_cast<hipStream_t>(cands[i]);
		bool beside_caller = false, beside_role = true;
		FIND_TRY(runs_beside(caller, s, c->ev[0], c->ev[1], &beside_caller));
		if (!beside_caller) continue;
		FIND_TRY(runs_beside(c->side[role], s, c->ev[0], c->ev[1], &beside_role));
		if (!beside_role) { *index = i; break; }
	}
	return FIND_OK;
}

// The layout the step was tuned with (and gets in a process that creates nothing else first): the large weight gradients (Q) and the two
// small-launch streams (T1, T2) each on a queue of their own, none of them the caller's, and the slab reduces (R) behind T2's queue.
// After torch.distributed has created RCCL's streams the same four hipStreamCreate calls put R on the CALLER's queue -- the reduces then
// sit between the dX GEMMs: 3.47 instead of 3.25 ms per train_3d step on every rank of a multi-GPU run.  So the first call that forks
// chooses its side streams among a dozen candidates by probing (once per context, ~20 ms).
namespace find {
namespace mlp {
// a side-stream candidate: non-blocking, or -- cu_reserve -- confined to the CUs the mask leaves (bit i of the mask = CU i / 8 of XCD i % 8,
// the driver's order on multi-XCD parts: clearing the low 8 R bits takes R CUs from every XCD)
static hipError_t create_side_stream(find_ctx* c, hipStream_t* out) {
	if (c->cu_reserve <= 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
	uint32_t mask[16] = {};
	const int ncu = std::min(c->num_cus, 512), lo = std::min(8 * c->cu_reserve, ncu - 8);
	for (int i = lo; i < ncu; ++i) mask[i >> 5] |= 1u << (i & 31);
	return hipExtStreamCreateWithCUMask(out, (uint32_t)((ncu + 31) / 32), mask);
}

static int bind_side_streams(find_ctx* c, hipStream_t caller) {
	c->side_bound = true;   // (one attempt: a failure below keeps the streams as created)
	constexpr int N_CAND = 12;
	hipStream_t st[1 + N_CAND];
	st[0] = caller;
	int n = 1;
	if (c->cu_reserve > 0) {
		// the streams find_ctx_create made carry no mask: replace them
		for (int k = 0; k < N_SIDE; ++k) {
			hipStream_t m = nullptr;
			if (create_side_stream(c, &m) != hipSuccess) { set_error("find_ctx: hipExtStreamCreateWithCUMask failed"); return FIND_ELAUNCH; }
			(void)hipStreamDestroy(c->side[k]);
			c->side[k] = m;
		}
		c->side_cus = c->num_cus - std::min(8 * c->cu_reserve, c->num_cus - 8);
	}
	for (int k = 0; k < N_SIDE; ++k) st[n++] = c->side[k];
	for (; n < 1 + N_CAND; ++n)
		if (create_side_stream(c, &st[n]) != hipSuccess) break;
	int g[1 + N_CAND];
	int rc = stream_groups(st, n, c->ev[0], c->ev[1], g);
	int pick[N_SIDE] = {-1, -1, -1, -1};
	if (rc == FIND_OK) {
		int ng = 0;
		for (int k = 0; k < 3; ++k)   // Q, T1, T2: first candidate on a queue that is neither the caller's nor an earlier pick's
			for (int i = 1; i < n && pick[k] < 0; ++i) {
				bool fresh = g[i] != 0;
				for (int j = 0; j < k; ++j) fresh = fresh && g[i] != g[pick[j]];
				if (fresh) { pick[k] = i; ++ng; }
			}
		if (ng == 3) {
			const int rq = pick[c->r_queue];
			for (int i = 1; i < n && pick[3] < 0; ++i)   // R: another stream on T2's queue (knob r_queue: Q's or T1's)
				if (i != pick[0] && i != pick[1] && i != pick[2] && g[i] == g[rq]) pick[3] = i;
			if (pick[3] < 0)   // (none: any stream that is not on the caller's queue and not a pick)
				for (int i = 1; i < n && pick[3] < 0; ++i)
					if (g[i] != 0 && i != pick[0] && i != pick[1] && i != pick[2]) pick[3] = i;
		}
	}
	const bool ok = rc == FIND_OK && pick[0] > 0 && pick[1] > 0 && pick[2] > 0 && pick[3] > 0;
	hipStream_t chosen[N_SIDE];
	for (int k = 0; k < N_SIDE; ++k) chosen[k] = ok ? st[pick[k]] : c->side[k];
	for (int i = 1; i < n; ++i) {
		bool keep = false;
		for (int k = 0; k < N_SIDE; ++k) keep = keep || st[i] == chosen[k];
		if (!keep) (void)hipStreamDestroy(st[i]);
	}
	for (int k = 0; k < N_SIDE; ++k) c->side[k] = chosen[k];
	return rc;
}
}  // namespace mlp
}  // namespace find

// ------------------------------------------------------------------------------------------- context
extern "C" int find_ctx_create(int device, find_ctx** out) {
	FIND_REQUIRE(out != nullptr, "find_ctx_create: out is NULL");
	*out = nullptr;
	int prev = 0, ndev = 0;
	FIND_HIP_OK(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
	FIND_REQUIRE(device >= 0 && device < ndev, "find_ctx_create: device %d out of range (%d visible)", device, ndev);
	FIND_HIP_OK(hipGetDevice(&prev), "hipGetDevice");
	FIND_HIP_OK(hipSetDevice(device), "hipSetDevice");
	find_ctx* c = new find_ctx();
	c->device = device;
	auto fail = [&](const char* what, hipError_t e) {
		set_error("find_ctx_create: %s: %s", what, hipGetErrorString(e));
		for (int i = 0; i < c->n_events; ++i) (void)hipEventDestroy(c->ev[i]);
		for (int i = 0; i < N_SIDE; ++i) if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
		delete c;
		(void)hipSetDevice(prev);
		return FIND_ELAUNCH;
	};
	hipError_t e;
	int v = 0;
	if ((e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return fail("CU count", e);
	c->num_cus = v > 0 ? v : 256;
	c->side_cus = c->num_cus;
	if ((e = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device)) != hipSuccess) return fail("LDS size", e);
	c->lds_bytes = v;
	for (int i = 0; i < N_SIDE; ++i)
		if ((e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreateWithFlags", e);
	for (c->n_events = 0; c->n_events < N_EVENTS; ++c->n_events)
		if ((e = hipEventCreateWithFlags(&c->ev[c->n_events], hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreateWithFlags", e);
	for (int i = 0; i < N_SIDE; ++i)
		if ((e = hipEventCreateWithFlags(&c->pend_ev[i], hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreateWithFlags", e);
	FIND_HIP_OK(hipSetDevice(prev), "hipSetDevice");
	*out = c;
	return FIND_OK;
}

extern "C" int find_ctx_destroy(find_ctx* c) {
	if (!c) return FIND_OK;
	int prev = 0;
	(void)hipGetDevice(&prev);
	(void)hipSetDevice(c->device);
	for (int i = 0; i < N_SIDE; ++i) if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
	for (int i = 0; i < c->n_events; ++i) (void)hipEventDestroy(c->ev[i]);
	(void)hipSetDevice(prev);
	delete c;
	return FIND_OK;
}

namespace {
struct Knob { const char* key; int find_ctx::*field; int64_t lo, hi; };
const Knob KNOBS[] = {
#ifdef FIND_DIAG
	{"gemm7", &find_ctx::gemm7, 0, 1}, {"x3_abl", &find_ctx::x3_abl, 0, 15},
#endif
	{"gemm4_small", &find_ctx::gemm4_small, 0, INT32_MAX},
	{"dw_pe_target", &find_ctx::dw_pe_target, 16, INT32_MAX}, {"dw_pe_lds_free", &find_ctx::dw_pe_lds_free, 0, 1}, {"dw2_min_cps", &find_ctx::dw2_min_cps, 1, INT32_MAX},
	{"bwd_streams", &find_ctx::bwd_streams, 0, 1}, {"fwd_streams", &find_ctx::fwd_streams, 0, 1}, {"reduce_stream", &find_ctx::reduce_stream, 0, 1},
	{"gemm5_min_units", &find_ctx::gemm5_min_units, 0, INT32_MAX}, {"fused_max_units", &find_ctx::fused_max_units, 0, 1024}, {"fused6", &find_ctx::fused6, 0, 1}, {"dw6_wgs", &find_ctx::dw6_wgs, 0, 512}, {"bind_streams", &find_ctx::bind_streams, 0, 1}, {"direct_w", &find_ctx::direct_w, 0, 1}, {"dw6_group", &find_ctx::dw6_group, 0, 1}, {"dwpe6", &find_ctx::dwpe6, 0, 1}, {"r_queue", &find_ctx::r_queue, 0, 2}, {"cu_reserve", &find_ctx::cu_reserve, 0, 16}, {"group_spf", &find_ctx::group_spf, 0, 4096}, {"dw_lds_free", &find_ctx::dw_lds_free, 0, FIND_DIAG_ON ? 3 : 1}, {"reduce_exclusive", &find_ctx::reduce_exclusive, 0, 2}, {"mlp_f16", &find_ctx::mlp_f16, 0, 2}, {"gemm6_min_units", &find_ctx::gemm6_min_units, 0, INT32_MAX}, {"lds_exclusive", &find_ctx::lds_exclusive, 0, 1}, {"defer_join", &find_ctx::defer_join, 0, 1}, {"act16", &find_ctx::act16, 0, 1}, {"bcast_fold", &find_ctx::bcast_fold, 0, 1}, {"group_head0", &find_ctx::group_head0, 0, 1}, {"footsum_fold", &find_ctx::footsum_fold, 0, 1}, {"pe_on_t2", &find_ctx::pe_on_t2, 0, 1},
};
}  // namespace

extern "C" int find_ctx_set(find_ctx* c, const char* key, int64_t value) {
	FIND_REQUIRE(c != nullptr && key != nullptr, "find_ctx_set: NULL argument");
#ifdef FIND_DIAG
	if (strcmp(key, "dbg") == 0) {  // device pointer to >= 4 * grid uint64 (profiling only)
		c->dbg = reinterpret_cast<unsigned long long*>(value);
		return FIND_OK;
	}
	if (strcmp(key, "dw2_verify") == 0) {  // device pointer to 8 + 64 * 8 uint64 (diagnosis only)
		c->dw2_verify = reinterpret_cast<unsigned long long*>(value);
		return FIND_OK;
	}
#endif
	if (strcmp(key, "ablate") == 0) {
		FIND_REQUIRE(value >= 0 && value <= INT32_MAX && (FIND_DIAG_ON || (value & ~(int64_t)find::MLP_SWITCHES) == 0),
		             "find_ctx_set: ablate = %lld has bits outside the result-preserving switches 0x%x (the others exist in libfind_hip_diag.so only)", (long long)value, find::MLP_SWITCHES);
		c->ablate = (int)value;
		return FIND_OK;
	}
	if (strcmp(key, "gemm4_min_units") == 0) {
		FIND_REQUIRE(value >= 0, "find_ctx_set: gemm4_min_units must be >= 0");
		c->gemm4_min_units = value;
		return FIND_OK;
	}
	for (const Knob& k : KNOBS)
		if (strcmp(key, k.key) == 0) {
			FIND_REQUIRE(value >= k.lo && value <= k.hi, "find_ctx_set: %s = %lld out of range [%lld, %lld]", key, (long long)value, (long long)k.lo, (long long)k.hi);
			c->*(k.field) = (int)value;
			return FIND_OK;
		}
	set_error("find_ctx_set: unknown key %s", key);
	return FIND_EINVAL;
}

extern "C" int find_ctx_get(const find_ctx* c, const char* key, int64_t* value) {
	FIND_REQUIRE(c != nullptr && key != nullptr && value != nullptr, "find_ctx_get: NULL argument");
	if (strcmp(key, "num_cus") == 0) { *value = c->num_cus; return FIND_OK; }
	if (strcmp(key, "side_cus") == 0) { *value = c->side_cus; return FIND_OK; }
	if (strcmp(key, "pending") == 0) { *value = (c->pend[0] || c->pend[1] || c->pend[2] || c->pend[3]) ? 1 : 0; return FIND_OK; }
	if (strcmp(key, "lds_bytes") == 0) { *value = c->lds_bytes; return FIND_OK; }
	if (strcmp(key, "device") == 0) { *value = c->device; return FIND_OK; }
	if (strcmp(key, "events_per_call_max") == 0) { *value = c->events_per_call_max; return FIND_OK; }
	if (strcmp(key, "gemm4_min_units") == 0) { *value = c->gemm4_min_units; return FIND_OK; }
	if (strcmp(key, "ablate") == 0) { *value = c->ablate; return FIND_OK; }
	if (strcmp(key, "diag") == 0) { *value = FIND_DIAG_ON; return FIND_OK; }
	for (const Knob& k : KNOBS)
		if (strcmp(key, k.key) == 0) { *value = c->*(k.field); return FIND_OK; }
	set_error("find_ctx_get: unknown key %s", key);
	return FIND_EINVAL;
}

extern "C" int find_render_switches(int64_t bits) {
	FIND_REQUIRE(bits >= 0 && (bits & ~(int64_t)find::RASTER_SWITCHES) == 0, "find_render_switches: bits 0x%llx outside the result-preserving switches 0x%x", (unsigned long long)bits, find::RASTER_SWITCHES);
	find::g_raster_ablate = (int)bits;
	return FIND_OK;
}

#ifdef FIND_DIAG
extern "C" int find_debug_raster_ablate(int64_t bits) {   // every bit, also those under which the render is wrong (include/find_hip_diag.h)
	find::g_raster_ablate = (int)bits;
	return FIND_OK;
}
#endif
