/* find_hip.h -- C ABI of libfind_hip.so: the MI355X (gfx950) hot path of FIND.
 *
 * The reference (OllieBoyne/FIND) is pure Python; its "FFI" for this path is the PyTorch / PyTorch3D
 * operator surface.  Each entry point below names the reference call site(s) it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (PyTorch-allocated); the library allocates no device
 *     memory.  Scratch is passed in as `ws` + `ws_bytes`; query sizes with *_ws_bytes().  The only persistent
 *     state is the opaque per-device context of find_ctx_create (internal HIP streams / events of the MLP entry
 *     points, per-kernel launch attributes, tuning knobs): nothing is process-global.
 *   - All tensors are contiguous row-major fp32 unless stated; indices are int32 or int64 as stated.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and performs no device synchronisation.
 *   - Return value: 0 on success, negative FIND_E* code otherwise; find_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.
 *   - Threading: a context is used by one thread at a time (the thread that owns the autograd graph); one
 *     context per device and process under data parallelism.  Entry points without a context are re-entrant.
 */
#ifndef FIND_HIP_H
#define FIND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIND_ABI_VERSION 2

#define FIND_OK 0
#define FIND_EINVAL (-1)   /* bad argument / unsupported configuration */
#define FIND_EWORKSPACE (-2) /* workspace too small */
#define FIND_ELAUNCH (-3)  /* HIP launch or runtime error */

#define FIND_MAX_LAYERS 8

int find_abi_version(void);
const char* find_last_error(void);
/* Name of the code-object architecture the library was built for ("gfx950"). */
const char* find_build_arch(void);

/* ------------------------------------------------------------------------------------------------
 * Context: per-device state of the MLP entry points -- the internal side streams and the event pool
 * that find_mlp_fwd / find_mlp_bwd fork work onto (always joined back into the caller's stream before the call
 * returns, also on an error return), the device's CU count and LDS size, per-kernel launch attributes and the
 * tuning knobs.  The reference has no counterpart (its operators keep their state inside PyTorch / PyTorch3D).
 * find_ctx_create makes `device` current for the calling thread while it builds the context and restores the
 * previous device.  A context must be destroyed only after the work launched through it has completed.
 * ---------------------------------------------------------------------------------------------- */
typedef struct find_ctx find_ctx;
int find_ctx_create(int device, find_ctx** out);
int find_ctx_destroy(find_ctx* ctx);
/* Which of the context's four side streams share a hardware queue with `caller_stream` or with each other (HIP maps streams
 * onto GPU_MAX_HW_QUEUES queues in creation order; two streams on one queue run in order).  groups[0] = 0 is the caller's stream,
 * groups[1 + k] side stream k; equal numbers = same queue.  Synchronises the streams it probes (~0.3 ms per probe). */
int find_ctx_stream_groups(find_ctx* ctx, void* caller_stream, int32_t* groups /* [5] */);
/* A second stream for the CALLER's own concurrent work (find_amd.model_with_loss: the Chamfer term beside the texture term's MLP pass, the
 * GT render beside the predicted one) that fits the context's stream layout: *index = the first of the n candidate streams that runs beside
 * `caller_stream` AND shares the hardware queue of side stream `role` (0 = Q, 1 = T1, 2 = T2, 3 = R), -1 if none does.  With four hardware
 * queues a fifth stream always shares one with somebody, and its work runs in order behind whatever that stream was given first: on T1's
 * queue the Chamfer backward waited ~0.2 ms behind the texture pass's weight gradients; Q is idle until the main pass's backward.
 * Binds the side streams first if that has not happened yet; synchronises the streams it probes. */
int find_ctx_stream_beside(find_ctx* ctx, void* caller_stream, void* const* candidates, int32_t n, int32_t role, int32_t* index);
/* Knobs (see the list at find_ctx_set below); find_ctx_get reads the current value, plus the read-only
 * "num_cus", "lds_bytes", "device", "events_per_call_max" (largest event count one MLP call has used so far). */
int find_ctx_set(find_ctx* ctx, const char* key, int64_t value);
int find_ctx_get(const find_ctx* ctx, const char* key, int64_t* value);

/* ------------------------------------------------------------------------------------------------
 * MLP: Fourier positional encoding + trunk + displacement head + colour head.
 * Replaces NeuralDisplacementField.forward  (src/model/model.py:393-453)
 *      and FourierFeatureTransform.forward  (src/utils/fourier_feature_transform.py:28-55),
 * plus their autograd backward (reached from src/train/trainer.py:121).
 *
 * Weight pointers use the reference state_dict tensors directly:
 *   trunk_w[i] = base.{2i}.weight   (width, in_i) ; in_0 = in_dim + 2*pe_size (or in_dim), else width
 *   disp_w[0]  = mlp_disp.0.weight  (width, width + lat_disp)  ... disp_w[n_disp] = final (3, width)
 *   col_w[0]   = mlp_col.0.weight   (width, width + lat_col)   ... col_w[n_col]   = final (3, width)
 * Only width == 256 and in_dim == 3 are supported (the reference's fixed setting, model.py:207).
 * ---------------------------------------------------------------------------------------------- */
typedef struct find_mlp_params {
	int32_t width;      /* hidden width (256) */
	int32_t in_dim;     /* 3 */
	int32_t pe_size;    /* Fourier mapping size (256); 0 = positional encoding disabled */
	int32_t n_trunk;    /* number of trunk Linear layers (depth + 1) */
	int32_t n_disp;     /* hidden Linear layers in the displacement head (dispdepth); final 3-wide layer is extra */
	int32_t n_col;      /* hidden Linear layers in the colour head (coldepth) */
	int32_t lat_disp;   /* latent columns appended to the disp-head input ([shapevec, posevec]) */
	int32_t lat_col;    /* latent columns appended to the col-head input (texvec) */
	const float* B;     /* (in_dim, pe_size) Fourier matrix (_B; not in the state_dict) */
	const float* trunk_w[FIND_MAX_LAYERS];
	const float* trunk_b[FIND_MAX_LAYERS];
	const float* disp_w[FIND_MAX_LAYERS];
	const float* disp_b[FIND_MAX_LAYERS];
	const float* col_w[FIND_MAX_LAYERS];
	const float* col_b[FIND_MAX_LAYERS];
	const float* avg_col; /* (3) added to the colour output when non-NULL (use_avg_colour, model.py:446-447) */
	int32_t precision;    /* arithmetic of the 256 -> 256 layers, per call (two models in one process may differ):
	                       *   0 = the context's "mlp_f16" knob;
	                       *   1 = fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's arithmetic;
	                       *   2 = fp16 MFMA operands (rounded to 11 bits) with fp32 accumulation: opt-in, BASELINE.json configs[4], no reference
	                       *       counterpart, ~1e-3 relative per layer;
	                       *   3 = bf16x3: every fp32 operand split EXACTLY into three bf16 pieces, the six products of relative size >= 2^-18
	                       *       on the bf16 matrix pipe, fp32 accumulation -- what is left out is <= 2^-26 of a product (a quarter of one fp32
	                       *       rounding), so results are as close to the float64 product as mode 1's (tests/test_gpu_mlp_bf16x3.py) at
	                       *       6/16 of its matrix-pipe time; the default of the Python surface (find_amd.functional.set_mlp_precision).
	                       *       Launches of >= "gemm6_min_units" 32-row units (gemm7_kernel / dw6_kernel) and the fused layer chains of small calls
	                       *       (fused6_kernel); what lies in between, the grouped weight gradients of small calls and the Fourier layer's run
	                       *       mode 1's kernels */
} find_mlp_params;

/* Gradient outputs, same shapes as the corresponding weights; every buffer given is OVERWRITTEN.
 * Which buffers may be NULL (what autograd's needs_input_grad asks for on the FIND path):
 *   - a head whose upstream gradient (d_disp / d_col) is NULL contributes nothing: its weight-gradient and latent-gradient pointers may be
 *     NULL (nothing is written); given, they receive exact zeros;
 *   - ALL weight-gradient pointers NULL = frozen network (requires_grad False on every weight; stage 3 of src/train/train.py:217-224 steps
 *     Adam(latent_params) only): the call stops at the heads' first layers and writes lat_disp / lat_col alone;
 *   - anything in between is refused (FIND_EINVAL). */
typedef struct find_mlp_grads {
	float* trunk_w[FIND_MAX_LAYERS];
	float* trunk_b[FIND_MAX_LAYERS];
	float* disp_w[FIND_MAX_LAYERS];
	float* disp_b[FIND_MAX_LAYERS];
	float* col_w[FIND_MAX_LAYERS];
	float* col_b[FIND_MAX_LAYERS];
	float* lat_disp; /* (n_feet, lat_disp) or NULL */
	float* lat_col;  /* (n_feet, lat_col) or NULL */
} find_mlp_grads;

/* Bytes of workspace find_mlp_fwd needs.  `pos_batch` is 1 (positions shared by every foot: the template,
 * model.py:404-406 / pytorch3d_tools.py:7-16) or n_feet.  With save_for_bwd the same buffer must be passed,
 * untouched, to find_mlp_bwd. */
int64_t find_mlp_ws_bytes(const find_mlp_params* p, int64_t pos_batch, int64_t n_feet, int64_t n_pts, int save_for_bwd);

/* pos (pos_batch, n_pts, 3); lat_disp (n_feet, lat_disp) or NULL; lat_col (n_feet, lat_col) or NULL (NULL when params say the head takes no
 * latents -- or when the call does not evaluate that head: disp / col NULL here, d_disp / d_col NULL in find_mlp_bwd);
 * out: disp (n_feet, n_pts, 3) = 0.1*tanh(.), col (n_feet, n_pts, 3) = 0.5*(1+tanh(.)) [+avg_col]. */
int find_mlp_fwd(find_ctx* ctx, const find_mlp_params* p, const float* pos, int64_t pos_batch, int64_t n_feet, int64_t n_pts,
				 const float* lat_disp, const float* lat_col, float* disp, float* col,
				 void* ws, int64_t ws_bytes, int save_for_bwd, void* stream);

/* Backward of find_mlp_fwd.  d_disp, d_col: (n_feet, n_pts, 3) upstream gradients (either may be NULL = zero).
 * `ws` is the forward workspace (save_for_bwd=1); `scratch` is extra space of find_mlp_bwd_scratch_bytes(). */
int64_t find_mlp_bwd_scratch_bytes(const find_mlp_params* p, int64_t pos_batch, int64_t n_feet, int64_t n_pts);
int find_mlp_bwd(find_ctx* ctx, const find_mlp_params* p, const float* pos, int64_t pos_batch, int64_t n_feet, int64_t n_pts,
				 const float* lat_disp, const float* lat_col, const float* d_disp, const float* d_col,
				 const void* ws, int64_t ws_bytes, void* scratch, int64_t scratch_bytes,
				 const find_mlp_grads* grads, void* stream);

/* Make `stream` wait for weight-gradient work a find_mlp_bwd left running under the "defer_join" knob.  Returns 1 if there was any,
 * 0 if not, a negative code on error. */
int find_ctx_join(find_ctx* ctx, void* stream);

/* One hidden layer  y = relu(x @ w^T + b)  with x (n_feet*n_pts, 256), w (256,256), b (256): the dominant kernel
 * of the path (nn.Linear + nn.ReLU pairs built at src/model/model.py:255-257, 353-356, 362-365).  Exposed so the
 * kernel can be timed and checked in isolation; find_mlp_fwd launches the same kernel.  w must be 16-byte aligned. */
int find_linear_relu_fwd(find_ctx* ctx, const float* x, const float* w, const float* b, int64_t n_feet, int64_t n_pts, float* y, void* stream);

/* Weight gradient of one 256 -> 256 layer:  dw[n][k] = sum_rows dz[row][n] * x[row][k]  (256 x 256, row-major) and, when db is not
 * NULL, db[n] = sum_rows dz[row][n], over rows = (foot, point) as above -- what autograd computes for the `weight` / `bias` of an
 * nn.Linear (src/model/model.py:255-257, 353-356, 362-365) from the layer's input x and the gradient dz of its output.  Exposed, like
 * find_linear_relu_fwd, so that the kernel (dw2_kernel; dw3_kernel in the fp16 mode) can be timed and checked in isolation;
 * find_mlp_bwd launches the same kernels.  `scratch` holds the partial tiles (find_linear_wgrad_scratch_bytes). */
int64_t find_linear_wgrad_scratch_bytes(int64_t n_feet);
int find_linear_wgrad(find_ctx* ctx, const float* dz, const float* x, int64_t n_feet, int64_t n_pts, float* dw, float* db,
					  void* scratch, int64_t scratch_bytes, void* stream);

/* Tuning knobs of a context (no reference counterpart).  Results do not depend on any knob except "mlp_f16" (the precision, below) and the
 * summation order of "reduce_exclusive" = 2, "footsum_fold", "group_head0" / "dw6_group" / "dwpe6" (which kernel sums a weight gradient).  The laboratory -- fault reproducers, superseded kernels kept for A/B runs, per-workgroup timers
 * and the ablation bits under which results are WRONG -- is not in this library: find_amd/build.py builds it from the same sources with
 * -DFIND_DIAG as libfind_hip_diag.so, whose additional keys and entry point include/find_hip_diag.h declares.  Read-only key "diag": 0 here,
 * 1 there.
 *   "gemm4_min_units" launches with at least this many 32-row x 128-column units use the W-resident kernel on column halves (default 1024)
 *   "gemm4_small"     ... and launches of at least this many 32-row units use it on column quarters (default 64; 0 = never);
 *                     anything smaller, and the two-segment trunk-output gradient, runs on the LDS-DMA ring kernel (gemm3)
 *   "dw2_min_cps", "dw_pe_target"   weight-gradient kernels: shortest row run per workgroup, workgroups of the Fourier layer's launch
 *   "dw_pe_lds_free"  Fourier layer's weight gradient: 1 (default) = dwpe_kernel (no LDS, features regenerated per lane, slabs of 2 pe + 32
 *                     columns; pe_size >= 32), 0 = the LDS-staged kernel of round 1 (A/B runs)
 *   "bwd_streams"     0 = backward on the caller's stream only, 1 = weight gradients on the context's side streams (default)
 *   "fwd_streams"     1 = the forward runs the colour head on a side stream beside the displacement head (default), 0 = one stream
 *   "reduce_stream"   1 = slab reduces of the large head layers on their own stream, two alternating slab sets; default 0 (behind their
 *                     weight-gradient launch: measured 0.6 - 0.9 % faster since the weight gradients use no LDS)
 *   "bind_streams"    1 (default) = the first call that forks picks the four side streams among a dozen candidates by probing which
 *                     hardware queue each one shares (see find_ctx_stream_groups); 0 = keep them as created
 *   "r_queue"         which side stream's hardware queue the slab-reduce stream shares: 0 = Q, 1 = T1, 2 = T2 (default, measured best)
 *   "defer_join"      1 = the NEXT find_mlp_bwd, if it is a small per-foot call (the fused-chain path: the texture pass of a train_3d step),
 *                     returns with its weight-gradient kernels still running on the context's side streams; its latent gradients are
 *                     complete on the caller's stream.  The weight-gradient buffers, `scratch` and `ws` of that call must stay untouched
 *                     until find_ctx_join() -- or the end of a later find_mlp_bwd on this context -- has made the reader's stream wait.
 *                     The knob resets itself with the call.  Ignored under stream capture and for the large-call paths.
 *   "mlp_f16"         default precision for calls whose find_mlp_params.precision is 0, and the precision of find_linear_relu_fwd /
 *                     find_linear_wgrad: 1 = the K = 256 Linear layers (forward, dX and dW) run on the fp16 matrix pipe: operands rounded to
 *                     fp16, fp32 accumulation, fp32 tensors (gemm5_kernel, dw3_kernel; BASELINE.json configs[4]).  Default 0: this knob DOES
 *                     change results (~1e-3 relative per layer); the Python surface is find_amd.functional.set_mlp_precision
 *                     2 = bf16x3 (find_mlp_params.precision 3: fp32-faithful, gemm7_kernel / dw6_kernel)
 *   "gemm5_min_units" in fp16 mode, launches of fewer 32-row units than this stay on the fp32 kernels (default 1024)
 *   "act16"           in fp16 mode only: 1 (default) = the heads' hidden activations and their gradients are STORED as fp16 inside the
 *                     call's workspace / scratch when a template is shared by more than one foot and the heads have >= "gemm5_min_units"
 *                     units (the matrix pipe rounds these values to fp16 anyway; BASELINE.json configs[4] is bound by their bytes); 0 = fp32
 *                     storage.  (A find_mlp_bwd follows what the find_mlp_fwd of its workspace did, also if a knob was turned in between.)
 *   "bcast_fold"      1 (default) = with a template shared by several feet, the output relu(P[v] + bias[foot]) of a head's broadcast first layer is
 *                     not stored: the second layer's forward GEMM, the ReLU mask of its dX GEMM and the x operand of its weight gradient form it
 *                     from the V x 256 product P and the bias rows (bf16x3: gemm7_kernel / dw6v_kernel; fp16 mode inside "act16": gemm5_kernel /
 *                     dw3_h16v_kernel).  Bit-identical to 0 (bias_relu_bcast_kernel materialises it).  The backward follows its forward's note
 *   "footsum_fold"    1 (default; bf16x3, needs "bcast_fold") = the dX GEMM that produces that layer's gradient forms the two sums the backward reads
 *                     of it -- over the feet, and per foot over the rows -- in its epilogue and stores no gradient tensor (gemm7_kernel<.., FSUM>);
 *                     0 = footsum_kernel over the stored tensor.  Same sums in another order (~2e-7 of a gradient's largest entry), deterministic
 *   "group_head0", "pe_on_t2", "direct_w", "dw6_group", "dwpe6"   scheduling / kernel-choice switches of the backward (A/B runs), all default 1:
 *                     the first head layers' weight gradients in the trunk's grouped launch; the Fourier layer's weight gradient on T2; kernels read
 *                     the model's weights without repacked copies; grouped bf16x3 weight gradients; the Fourier layer's on the bf16 pipe
 *   "gemm6_min_units" the same threshold for the bf16x3 kernels (default 1024)
 *   "fused_max_units" calls of at most this many 32-row units (0..1024, default 512) run whole layer chains -- the trunk, trunk + heads of a
 *                     per-foot pass, their dX chains -- in one launch of fused_chain_kernel, and the weight gradients of a chain as one grouped
 *                     launch + one grouped reduce; 0 = one launch per layer at every size
 *   "fused6"          bf16x3 calls run their chains on fused6_kernel (weights pre-split by split_w_kernel into the call's workspace; default 1);
 *                     0 = fused_chain_kernel (fp32 MFMA), as the other precisions
 *   "dw6_wgs"         workgroups (= 256 x 256 slabs) of a dw6_kernel launch: 0 (default) = one per CU, half that inside a backward whose side
 *                     streams are on (it runs beside the next layer's dX GEMM; half the slabs are half the reduce's traffic)
 *   "dw_lds_free"     kernel of the 256 x 256 weight gradients: 1 = dw4_kernel (operands straight from global memory, no LDS, <= 256
 *                     registers; default), 0 = dw2_kernel (LDS-DMA ring, the whole register file of its SIMDs claimed)
 *   "lds_exclusive"   1 = the LDS-DMA ring kernels reserve their CU's whole LDS: round 1's containment of that fault, which turned out
 *                     to be about registers; default 0
 *   "reduce_exclusive" the stress test's hook for that fault (tests/test_gpu_mlp.py): 1 = the slab-reduce kernels reserve their CU's whole LDS;
 *                     2 = they use no LDS and are slow, so that they stay resident beside later weight-gradient kernels; default 0
 *   "ablate"          switches that leave results unchanged (the tests compare them): 16 no s_setprio in gemm4 / gemm7, 32 every column block in
 *                     the Fourier layer's weight gradient, 128 fused chains always on 32-row tiles.  Any other bit is refused here.
 * The Python binding applies FIND_TUNING="key=value,..." from the environment to every context it creates. */

/* Process-wide switches of the rasteriser and of find_chamfer_fwd's neighbour search, all with unchanged results (tests/test_gpu_render.py,
 * tests/test_gpu_geom.py compare them): 8 no early exit of finished pixels / tiles, 16 tile lists left in face order (no depth-slab sort),
 * 256 a list pool of 512 entries per image (tiles without room scan the faces themselves); Chamfer: 512 all pairs at every size, 1024 the
 * uniform grid from 64 points per cloud on (geom.hip); 2048 / 4096 the band / the candidate-list rasteriser at every image size (default: the
 * band kernel from 384^2 pixels on, csrc/render_band.h -- the two agree in everything but the last bits of the alpha products).  Any other bit
 * is refused (FIND_EINVAL). */
int find_render_switches(int64_t bits);
/* ------------------------------------------------------------------------------------------------
 * Latent-table lookup.  Replaces LatentVector.__getitem__ with a tensor of indices (src/model/model.py:131-152;
 * call sites src/model/model.py:360-372 get_meshes_from_batch) and the index_put its autograd runs backward.
 * table (n_rows, dim) fp32; idx (n_idx) int64, negative values count from the end; an out-of-range index yields a
 * NaN row (no host synchronisation here).  The backward owns every table element by one thread and sums the
 * matching rows in index order: deterministic with duplicates, untouched rows come out zero.
 * ---------------------------------------------------------------------------------------------- */
int find_latent_gather_fwd(const float* table, int64_t n_rows, int64_t dim, const int64_t* idx, int64_t n_idx, float* out, void* stream);
int find_latent_gather_bwd(const float* d_out, const int64_t* idx, int64_t n_idx, int64_t n_rows, int64_t dim, float* d_table, void* stream);
/* The same for up to 8 tables in one launch each way -- trainer.sample_latent_vectors (src/train/trainer.py:29-46) looks up the shape,
 * pose, texture and registration rows of a batch, one LatentVector.__getitem__ each.  tables / idx / outs (and n_rows, dims) are HOST
 * arrays of n_tables entries; every lookup has n_idx indices.  Backward: d_outs[t] may be NULL (no gradient arrived: d_tables[t] = 0). */
int find_latent_gather_many_fwd(int64_t n_tables, const float* const* tables, const int64_t* n_rows, const int64_t* dims,
								const int64_t* const* idx, int64_t n_idx, float* const* outs, void* stream);
int find_latent_gather_many_bwd(int64_t n_tables, const float* const* d_outs, const int64_t* n_rows, const int64_t* dims,
								const int64_t* const* idx, int64_t n_idx, float* const* d_tables, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss weighting: scaled[i] = *terms[i] * weights[i], *total = sum_i scaled[i] in term order (ModelWithLoss.forward,
 * src/model/model.py:1157-1163: losses[k] = raw * opts.weight_k, loss = sum).  terms: HOST array of n <= 8 device scalars; weights:
 * host floats; scaled (n) and total: device.  Backward: d_terms[i] = weights[i] * (*g_total + g_scaled[i]); either upstream may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int find_weighted_terms_fwd(int64_t n, const float* const* terms, const float* weights, float* scaled, float* total, void* stream);
int find_weighted_terms_bwd(int64_t n, const float* weights, const float* g_total, const float* g_scaled, float* d_terms, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Similarity registration  X = ((v + disp) * S) @ R(euler 'XYZ') + t.
 * Replaces euler_angles_to_matrix + Transform3d().scale().rotate().translate().transform_points
 * in NeuralDisplacementField.get_meshes (src/model/model.py:481-491).
 * reg (n_feet, 9) = [t(3), euler(3), S(3)]; verts (verts_batch in {1, n_feet}, n_pts, 3).
 * ---------------------------------------------------------------------------------------------- */
int find_register_fwd(const float* verts, int64_t verts_batch, const float* disp, const float* reg,
					  int64_t n_feet, int64_t n_pts, float* out, void* stream);
/* d_out (n_feet,n_pts,3) -> d_disp (n_feet,n_pts,3), d_reg (n_feet,9).  ws: find_register_bwd_ws_bytes(). */
int64_t find_register_bwd_ws_bytes(int64_t n_feet, int64_t n_pts);
int find_register_bwd(const float* verts, int64_t verts_batch, const float* disp, const float* reg,
					  const float* d_out, int64_t n_feet, int64_t n_pts, float* d_disp, float* d_reg,
					  void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Surface sampling gather.  Replaces the gather/lerp half of pytorch3d.ops.sample_points_from_meshes
 * (call sites src/model/losses.py:39-41,63,67; src/eval/eval_3d.py:149-150); the random draws
 * (face index, u, v) are INPUTS so CPU and GPU runs see identical samples.
 * verts (n_meshes, n_verts, 3); faces (n_faces, 3) int32 shared topology when faces_batch == 1, else
 * (n_meshes, n_faces, 3); face_idx (n_meshes, n_samples) int32; uv (n_meshes, n_samples, 2).
 * out (n_meshes, n_samples, 3) = w0*v0 + w1*v1 + w2*v2,  (w0,w1,w2) = (1-sqrt(u), sqrt(u)(1-v), sqrt(u)v).
 * `attr` (optional, same layout as verts, n_attr channels=3) is sampled into attr_out with the same weights.
 * ---------------------------------------------------------------------------------------------- */
int find_sample_points_fwd(const float* verts, const int32_t* faces, int64_t faces_batch, const int32_t* face_idx,
						   const float* uv, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_samples,
						   float* out, const float* attr, float* attr_out, void* stream);
/* d_out (n_meshes,n_samples,3) scattered to d_verts (n_meshes,n_verts,3), which must be zero-initialised. */
int find_sample_points_bwd(const int32_t* faces, int64_t faces_batch, const int32_t* face_idx, const float* uv,
						   const float* d_out, int64_t n_meshes, int64_t n_verts, int64_t n_faces, int64_t n_samples,
						   float* d_verts, void* stream);
/* Face areas 0.5*|(v1-v0)x(v2-v0)| (n_meshes, n_faces): the multinomial weights of the sampler. */
int find_face_areas(const float* verts, const int32_t* faces, int64_t faces_batch, int64_t n_meshes, int64_t n_verts,
					int64_t n_faces, float* areas, void* stream);
/* The whole of sample_points_from_meshes with the face choice on the device (same call sites): faces ~ multinomial(area) with
 * replacement -- a block per mesh builds the running sum of the face areas (ws: find_sample_surface_ws_bytes), a thread per sample
 * searches it with its uniform draw -- then the gather above.  rnd (n_meshes, n_samples, 3) uniform in [0,1): [face draw, u, v];
 * only the DRAWS are inputs (torch's device generator, exactly where PyTorch3D would draw).  Outputs: face_idx (n_meshes, n_samples)
 * int32 and uv (n_meshes, n_samples, 2) -- what find_sample_points_bwd / find_uv_sample need --, out and optionally attr_out as
 * above.  Faces of zero area (and the -1 padding of ragged batches) are never chosen; a mesh of total area 0 yields face 0. */
int64_t find_sample_surface_ws_bytes(int64_t n_meshes, int64_t n_faces);
int find_sample_surface_fwd(const float* verts, const int32_t* faces, int64_t faces_batch, const float* rnd, int64_t n_meshes,
							int64_t n_verts, int64_t n_faces, int64_t n_samples, int32_t* face_idx, float* uv, float* out,
							const float* attr, float* attr_out, void* ws, int64_t ws_bytes, void* stream);
/* The same for a mesh whose running area sum is already in `ws`: left there by an earlier find_sample_surface_fwd on the SAME verts /
 * faces (a GT scan is sampled twice per training step and does not change between steps: losses.py:39,63).  Only the search + gather run. */
int find_sample_surface_again(const float* verts, const int32_t* faces, int64_t faces_batch, const float* rnd, int64_t n_meshes,
							  int64_t n_verts, int64_t n_faces, int64_t n_samples, int32_t* face_idx, float* uv, float* out,
							  const float* attr, float* attr_out, const void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Chamfer nearest neighbour (K=1, squared L2, brute force).
 * Replaces pytorch3d.ops.knn_points inside pytorch3d.loss.chamfer_distance
 * (call sites src/model/losses.py:77,85,88; src/eval/eval_3d.py:151,159).
 * x (n, p1_max, 3) with per-cloud lengths x_len (n) int32 (NULL = all p1_max); same for y.
 * Outputs for every valid x_i: dist (n,p1_max) = min_j |x_i-y_j|^2, idx (n,p1_max) int32 = argmin_j
 * (lowest j on ties); padding rows get dist 0 / idx -1.
 * ---------------------------------------------------------------------------------------------- */
int find_nn_fwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n,
				int64_t p1_max, int64_t p2_max, float* dist, int32_t* idx, void* stream);
/* Chamfer gradient for one direction: for valid i, g = 2*w[n,i]*(x_i - y_idx);  d_x[n,i] += g ; d_y[n,idx] -= g.
 * w (n,p1_max) is the upstream gradient of dist.  d_x / d_y must be zero-initialised (or hold the other
 * direction's contribution); either may be NULL. */
int find_nn_bwd(const float* x, const int32_t* x_len, const float* y, const int32_t* idx, const float* w, int64_t n,
				int64_t p1_max, int64_t p2_max, float* d_x, float* d_y, void* stream);
/* pytorch3d.loss.chamfer_distance with its defaults as ONE loss (the reference never uses anything else: losses.py:77,85,88;
 * eval_3d.py:151,159): both nearest-neighbour directions in one launch, then
 *   loss = ( sum_n sum_i min_j|x_i-y_j|^2 / max(x_len[n],1)  +  sum_n sum_j min_i|x_i-y_j|^2 / max(y_len[n],1) ) / n
 * as a 1-element device scalar (deterministic reduction).  ws (find_chamfer_ws_bytes) keeps the neighbours for the backward and must
 * be passed to it untouched.  find_chamfer_bwd: g_loss = device scalar, the upstream gradient; d_x (n,p1_max,3) / d_y (n,p2_max,3)
 * must be zero-initialised, either may be NULL; contributions are added with float atomics (as PyTorch3D's knn backward). */
int64_t find_chamfer_ws_bytes(int64_t n, int64_t p1_max, int64_t p2_max);
int find_chamfer_fwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n, int64_t p1_max,
					 int64_t p2_max, float* loss, void* ws, int64_t ws_bytes, void* stream);
int find_chamfer_bwd(const float* x, const int32_t* x_len, const float* y, const int32_t* y_len, int64_t n, int64_t p1_max,
					 int64_t p2_max, const float* g_loss, const void* ws, int64_t ws_bytes, float* d_x, float* d_y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Masked colour error of TextureLossGTSpace (src/model/losses.py:43-57): mask = any(target < 1) per point, loss = mean over all
 * n_pts * 3 elements of mask * (pred - target)^2 -- F.mse_loss(reduction='none') * mask, .mean() -- as a 1-element device scalar.
 * pred, target (n_pts, 3).  The backward overwrites d_pred (n_pts, 3) with g_loss * dloss/dpred (g_loss: device scalar).
 * ---------------------------------------------------------------------------------------------- */
int find_masked_mse_fwd(const float* pred, const float* target, int64_t n_pts, float* loss, void* stream);
int find_masked_mse_bwd(const float* pred, const float* target, int64_t n_pts, const float* g_loss, float* d_pred, void* stream);
/* Image losses in one pass each way: loss = mean over n_pix x channels of (a * a_mask - b * b_mask)^2, masks per pixel or NULL (= 1).
 * Replaces nn.MSELoss on image * mask.unsqueeze(-1) products (pixel loss, src/model/model.py:1101-1105: images compared inside their
 * silhouettes) and on the two masks (silhouette loss, model.py:1107-1108 / losses.py:122-128: channels = 1, no masks).  ws: find_image_mse_ws_bytes().
 * Backward: gradients w.r.t. a (d_a, same shape) and a_mask (d_a_mask, per pixel), either may be NULL; b / b_mask are the GT side. */
int64_t find_image_mse_ws_bytes(void);
int find_image_mse_fwd(const float* a, const float* a_mask, const float* b, const float* b_mask, int64_t n_pix, int64_t channels, float* loss,
					   void* ws, int64_t ws_bytes, void* stream);
int find_image_mse_bwd(const float* a, const float* a_mask, const float* b, const float* b_mask, int64_t n_pix, int64_t channels,
					   const float* g_loss, float* d_a, float* d_a_mask, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mesh smoothness: mesh_edge_loss(target 0) and mesh_laplacian_smoothing('cot').
 * Replaces pytorch3d.loss.mesh_edge_loss / mesh_laplacian_smoothing (call site src/model/losses.py:95-97).
 * Topology is shared by every mesh in the batch (one template) and static, so the host builds two CSR tables once:
 *   vf_off (n_verts+1), vf_items (3*n_faces): for every vertex the incident corners, item = face*3 + corner;
 *   nbr_off (n_verts+1), nbr_idx (2*n_edges): for every vertex its neighbours over the unique undirected edges.
 * faces (n_faces,3) int32; verts (n_meshes,n_verts,3).  All kernels are deterministic gathers (no float atomics).
 * Outputs: loss_edge, loss_lap: 1-element device scalars (batch means, as PyTorch3D).
 * ws: find_smooth_ws_bytes();  the forward leaves what the backward needs in ws.
 * ---------------------------------------------------------------------------------------------- */
int64_t find_smooth_ws_bytes(int64_t n_meshes, int64_t n_verts, int64_t n_faces);
int find_smooth_fwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
					const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
					int64_t n_edges, float* loss_edge, float* loss_lap, void* ws, int64_t ws_bytes, void* stream);
/* d_verts (n_meshes,n_verts,3) OVERWRITTEN with g_edge*dLedge/dV + g_lap*dLlap/dV; g_* are device scalars. */
int find_smooth_bwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
					const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
					int64_t n_edges, const float* g_edge, const float* g_lap, void* ws, int64_t ws_bytes, float* d_verts,
					void* stream);
/* MeshSmoothnessLoss as one scalar (src/model/losses.py:93-99: 0.1 * laplacian + 10 * edge): loss = w_edge * loss_edge + w_lap * loss_lap
 * written by the same three kernels; the backward takes the upstream gradient of that scalar (device pointer) and the two weights. */
int find_smooth_loss_fwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
						 const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
						 int64_t n_edges, float w_edge, float w_lap, float* loss, void* ws, int64_t ws_bytes, void* stream);
int find_smooth_loss_bwd(const float* verts, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_items,
						 const int32_t* nbr_off, const int32_t* nbr_idx, int64_t n_meshes, int64_t n_verts, int64_t n_faces,
						 int64_t n_edges, float w_edge, float w_lap, const float* g_loss, void* ws, int64_t ws_bytes,
						 float* d_verts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Differentiable mesh render: world->view->NDC transform, rasterisation, fused shading.
 * Replaces FootRenderer.forward / rasterize (src/model/renderer.py:208-245, 247-383), i.e. PyTorch3D's
 * FoVPerspectiveCameras + rasterize_meshes (naive and binned) + SoftSilhouetteShader + SoftPhongShader +
 * softmax_rgb_blend (in-repo mirror src/model/renderer.py:23-72), and their backward.
 *
 * Image index = mesh*n_views + view (renderer.py:271-277).  verts (n_meshes, n_verts, 3) world space;
 * faces (n_faces,3) int32 shared when faces_batch==1 else (n_meshes,n_faces,3); R (n_views,3,3), T (n_views,3)
 * in PyTorch3D row-vector convention (p_view = p_world @ R + T).
 * ---------------------------------------------------------------------------------------------- */
typedef struct find_render_params {
	int32_t image_h, image_w;
	float fov_deg;        /* 60 */
	float znear, zfar;    /* 0.02, 100 (renderer.py:274; PyTorch3D default zfar) */
	float sil_blur_radius;/* log(1/1e-4 - 1) * 1e-4 (renderer.py:127) */
	float sil_sigma;      /* 1e-4 (renderer.py:124) */
	int32_t sil_faces_per_pixel; /* 100 (renderer.py:128); the soft mask uses the K nearest-in-depth faces */
	float rgb_sigma, rgb_gamma;  /* 1e-4, 1e-4 BlendParams defaults */
	float background[3];  /* (1,1,1) renderer.py:109,118 */
	float light_pos[3];   /* (0,0,100) renderer.py:114 */
	float ambient, diffuse, specular, shininess; /* 0.5, 0.3, 0.2, 64 (PyTorch3D PointLights/Materials defaults) */
	float z_clip;         /* znear/2 (renderer.py:231-234) */
} find_render_params;

int64_t find_render_ws_bytes(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t n_verts,
							 int64_t n_faces);
/* Outputs (any may be NULL to skip): mask (n_meshes,n_views,H,W) soft silhouette; image (n_meshes,n_views,H,W,3)
 * Phong RGB; pix_to_face (n_meshes,n_views,H,W) int32 nearest face (K=1, blur 0) packed id mesh*n_faces+f or -1;
 * zbuf (n_meshes,n_views,H,W) depth of that face (-1 where empty).  vert_colors (n_meshes,n_verts,3) needed for image. */
int find_render_fwd(const find_render_params* rp, const float* verts, const int32_t* faces, int64_t faces_batch,
					const float* vert_colors, const float* R, const float* T, int64_t n_meshes, int64_t n_views,
					int64_t n_verts, int64_t n_faces, float* mask, float* image, int32_t* pix_to_face, float* zbuf,
					void* ws, int64_t ws_bytes, void* stream);
/* d_mask / d_image upstream grads (either NULL); `mask` is the forward's soft silhouette (required with d_mask; the kernel reads the
 * alpha product the forward saved in ws beside it -- 1 - mask is zero wherever the mask has rounded to 1).
 * d_verts (n_meshes,n_verts,3) and d_vert_colors (same, may be NULL) are OVERWRITTEN.  ws must be the forward workspace,
 * untouched since find_render_fwd (it holds the projected faces and the nearest-fragment buffers). */
int find_render_bwd(const find_render_params* rp, const float* verts, const int32_t* faces, int64_t faces_batch,
					const float* vert_colors, const float* R, const float* T, int64_t n_meshes, int64_t n_views,
					int64_t n_verts, int64_t n_faces, const float* mask, const float* d_mask, const float* d_image,
					float* d_verts, float* d_vert_colors, void* ws, int64_t ws_bytes, void* stream);
/* Diagnostics of the last forward that used `ws` (synchronises the stream): out2[0] = faces straddling the z-clip
 * plane (PyTorch3D would clip them; they are rasterised whole here -- none exists on FIND's camera set-up),
 * out2[1] = pixels left UNRESOLVED by the K-nearest rule: a pixel with more silhouette candidates than
 * sil_faces_per_pixel keeps the K nearest in depth (ties to the earlier face, as PyTorch3D's per-pixel K-buffer);
 * only a pixel with more than 4096 candidates is not resolved -- all of its candidates stay blended. */
int find_render_flags(const void* ws, int32_t* out2, void* stream);


/* ------------------------------------------------------------------------------------------------
 * UV textures.  Replaces pytorch3d TexturesUV.sample_textures as the reference uses it for GT scans
 * (src/data/dataset.py:263-271 builds TexturesUV(img, faces_uvs, verts_uvs); losses.py:39-43 samples GT surface colours;
 * renderer.py:329-346 renders GT images): colour at a surface point = bilinear read of the vertically flipped map at the
 * barycentric mix of the face's three UV vertices (grid_sample align_corners=True, padding_mode='border').
 * maps (n_maps,H,W,3); verts_uvs (n_maps,Vt,2); faces_uvs (1|n_maps, F, 3) int32; face_idx / bary (n_rows,P[,3]) with
 * n_rows a multiple of n_maps (rows of one map consecutive: feet x views); face_idx < 0 -> zeros.
 * find_render_frags copies out the nearest-fragment buffers (local face id, barycentrics) of the forward that used `ws`.
 * ---------------------------------------------------------------------------------------------- */
int find_uv_sample(const float* maps, int64_t n_maps, int64_t map_h, int64_t map_w, const float* verts_uvs, int64_t n_uv_verts,
				   const int32_t* faces_uvs, int64_t faces_batch, int64_t n_faces, const int32_t* face_idx, const float* bary,
				   int64_t n_rows, int64_t n_points, float* out, void* stream);
int find_render_frags(const find_render_params* rp, int64_t n_meshes, int64_t n_views, int64_t n_verts, int64_t n_faces, const void* ws,
					  int32_t* face_local, float* bary, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor optimiser steps.  Replace torch.optim.Adam / torch.optim.SGD(momentum=0.9)
 * as constructed by the reference (src/train/train.py:161-168) and stepped once per batch
 * (src/train/trainer.py:121-123).  `param`, `grad`, moment arrays: HOST arrays of n_tensors DEVICE pointers (fp32,
 * contiguous); `numel`: host array.  Dense updates, torch's single-tensor arithmetic operation for operation;
 * amsgrad / maximize / foreach-only options are not provided.
 * find_adam_step: `step` is the 1-based count INCLUDING this update (torch's state['step'] after it).
 * find_sgd_step: `first_step` != 0 initialises the momentum buffers to the gradient (torch clones it).
 * ---------------------------------------------------------------------------------------------- */
int find_adam_step(int64_t n_tensors, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
				   const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);
 /* find_adam_step_dev: the same update with the step count on the DEVICE -- `step_dev` points to one fp32 holding the 1-based count
 * including this update (the caller increments it on `stream` first), and the bias corrections are formed on the device in fp32, as
 * torch.optim.Adam(capturable=True) does: nothing step-dependent is baked into the launch, so the call can be captured in a HIP graph
 * and replayed (find_amd/graph.py). */
int find_adam_step_dev(int64_t n_tensors, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
					   const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay, const float* step_dev, void* stream);
int find_sgd_step(int64_t n_tensors, float* const* param, const float* const* grad, float* const* momentum_buf, const int64_t* numel,
				  float lr, float momentum, float dampening, float weight_decay, int nesterov, int first_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FIND_HIP_H */
