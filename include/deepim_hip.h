/*
 * deepim_hip.h — C ABI of libdeepim_hip.so, the MI355X (gfx950) implementation of
 * mx-DeepIM's render-and-compare inner loop.
 *
 * Boundary rules
 *   - plain C linkage, plain pointers and sizes, no torch / MXNet types;
 *   - every `deepim_*` entry returns 0 on success, non-zero on failure
 *     (`deepim_last_error()` gives the HIP error text); nothing prints-and-continues
 *     the way the reference's CUDA wrapper does (lib/flow_c/gpu_flow_kernel.cu:18-25);
 *   - "d_" / unprefixed tensor pointers are DEVICE pointers obtained from
 *     deepim_malloc(); tensors are dense NCHW float32 unless stated;
 *   - all work is enqueued on the context's HIP stream and is asynchronous;
 *     deepim_sync() / the d2h copy are the synchronisation points.
 *
 * Each entry cites the reference interface it replaces (path:line under the
 * reference checkout).
 */
#ifndef DEEPIM_HIP_H_
#define DEEPIM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct deepim_ctx deepim_ctx;

/* ---------------------------------------------------------------- runtime -- */
/* replaces `mx.gpu(i)` contexts + MXNet NDArray storage (deepim/core/tester.py:27-47) */
int deepim_create(int device_id, deepim_ctx** out);
int deepim_destroy(deepim_ctx* ctx);
const char* deepim_last_error(void);
int deepim_device_count(int* n);
int deepim_malloc(deepim_ctx* ctx, size_t bytes, void** dptr);
int deepim_free(deepim_ctx* ctx, void* dptr);
int deepim_memset(deepim_ctx* ctx, void* dptr, int value, size_t bytes);
int deepim_h2d(deepim_ctx* ctx, void* dst, const void* src, size_t bytes);   /* sync */
int deepim_d2h(deepim_ctx* ctx, void* dst, const void* src, size_t bytes);   /* sync */
int deepim_d2d(deepim_ctx* ctx, void* dst, const void* src, size_t bytes);   /* async */
int deepim_sync(deepim_ctx* ctx);
/* y[i] += alpha * x[i] — CustomOp.assign(dst, 'add', src) (mx.operator.CustomOp.assign) */
int deepim_axpy(deepim_ctx* ctx, float* y, const float* x, float alpha, size_t n);
/* strided channel-slice copy: dst[b, dst_coff:dst_coff+C, :] = src[b, :, :] (MXNet Concat, async) */
int deepim_copy_channels(deepim_ctx* ctx, float* dst, int dst_ctotal, int dst_coff,
                         const float* src, int C, int B, size_t hw);
void* deepim_stream(deepim_ctx* ctx);           /* hipStream_t, for interop */
/* integer tuning knobs. "conv_max_split": 0 = auto split-K (default), 1 = never split AND canonical order
 * (conv/deconv results are then a single (ci,ky,kx)-ordered fmaf chain, bit-identical to the oracle; with NC8 input —
 * deepim_conv2d_forward_ex — still a single chain, in that kernel's own (c/8,ky,kx,s,h) order), n = cap.
 * "conv_direct": 1 (default) = 128x128-tiled convs with even Cin run on the LDS-free register-fed kernel, whose
 * fmaf chain runs over (ci/2,ky,kx,ci%2) (differs from the canonical order in the last bits; the oracle has the
 * matching order switch); it steps aside when conv_max_split == 1; 2 = use it even then; 0 = never.
 * "conv_tail_split": 1 = the autotuner may cut only the tiles of the under-filled last round of a launch into K
 * slices (default 0: a pair's summation order would depend on its position in the batch); "conv_tail_slots": round
 * size in 128x128 blocks (default 1024 = 256 CUs x 4); "conv_force_plan": dev knob, n > 0 uniform split-K,
 * n < 0 tail split with -n slices, 0 = heuristic/autotuner.
 * "conv_xcd_swizzle": 1 (default) = XCD-aware tile order, 0 = plain block order.
 * "conv_autotune": 0 (default) = the split-K factor of an under-filled conv grid comes from a deterministic cost
 * model (same geometry → same summation order in every run, process and rank); 1 = on the first call of a conv
 * geometry, time a few split-K factors and keep the fastest (plans then depend on timing noise). "conv_tile256": 1 = 256x128 tiles on 512-thread blocks when
 * Cout % 256 == 0 (default 0: measured no faster than 128x128). "wgrad_lds": 1 (default) = LDS-staged weight-gradient kernel
 * (and the few-filter stream kernel for Cout <= 4), 0 = the round-2 register-fed kernel; "dgrad_group": 1 (default) = the four
 * parity classes of deepim_conv2d_dgrad_s2 share one launch, 0 = class by class (A/B measurements). "wino_two_wave": 0 (default) =
 * the Winograd layers on the one-wave-per-SIMD kernel (16 positions per wave), 1 = the two-wave form (8 positions per wave, LDS
 * hand-over; measured slower on the big layers); "wino_s2d_skip": 1 (default) = deepim_conv2d_wino_forward_s2d skips the positions
 * whose transformed weights are identically zero, 0 = runs all 16 (A/B measurements). "wino_shared": 1 (default) = Winograd layers with
 * Cout % 64 == 0 run on the shared-transform kernel (8- or 4-wave blocks, the input transform computed once per block and handed over
 * through LDS), 0 = the round-4 one-wave kernel. "wino_wide": block shape of that kernel — 1 (default) = per layer by the work per CU,
 * 0 = 64 channels x 64 tiles, 3 = 128 x 32 (Cout % 128 == 0), 2 = 64 x 32 on four waves, two blocks per CU. "wino_split": its split
 * over the input channels where the grid does not fill the chip — 0 (default) = the deterministic plan of the geometry, 1 = never
 * (one block walks all of Cin: the 3x3 layers are then bit-identical to the one-wave kernel), n = at most n slices. "wino_persistent": 1
 * (default) = its grid is one block per resident slot (256 of 8 waves, 512 of 4), each walking its share of the tile blocks, 0 = one
 * block per tile block (same results). "wino_streamk": 1 (default) = where that walk would end in a partly filled round the persistent
 * blocks share that round granule by granule, a cut tile block is finished by whichever of its pieces arrives last (a fixed order of adds:
 * deterministic; rounding as for a K split; like "conv_tail_split" it makes the last bits of a pair's result depend on the pair's place in
 * the batch — which tile blocks are cut does —, never its 1e-5-of-range bound; on by default since it is worth +1 % of the loop where the
 * direct kernels' tail split was not; off with wino_split = 1 or wino_persistent = 0; 2 = wherever it applies, whatever the cost model
 * says; 0 = never: permuting the pairs of a batch then permutes the results bit for bit). "wino_fin": 1 = a layer split over
 * the input channels is finished inside the kernel — the block whose slice of a tile block arrives last adds the raw copies in slice order,
 * the bias and the activation: the second pass's sums bit for bit, without its launch —, 0 (default) = wino_reduce_kernel as a second pass
 * (measured on MI355X, profiles/r06_b4_share.md: the lone finishing block's read-back is a serial tail that costs more than the parallel
 * pass it replaces — conv6_1 at B = 4: 48 -> 129 us). The arrival counters of both in-kernel finishes (64 KB) are allocated by
 * deepim_create: a layer's plan depends on geometry and options only, under graph capture as in eager runs. "conv_fewout_quad": 1 (default) = the 3x3
 * stride-1 pad-1 convolutions with Cout <= 4 and W % 4 == 0 (flow / mask predictors) compute four pixels per lane, 0 = one;
 * "conv_fewout_blocks": the grid those few-filter convolutions slice their input channels for — 0 (default) = ~1024 blocks where the pixels
 * alone give >= 32 blocks, ~512 below (measured, profiles/r06_heads.md), n = about n blocks; "conv_fewout_minc": the smallest channel slice
 * (default 32). Unknown names fail. */
int deepim_set_option(deepim_ctx* ctx, const char* name, int value);
/* *value = the current setting of an option deepim_set_option knows (host code that has to follow the context's kernel selection —
 * e.g. which weight-gradient layout the training graph registers — reads it here). Unknown names fail. */
int deepim_get_option(deepim_ctx* ctx, const char* name, int* value);
/* Device-side ordering between two contexts (two streams) of ONE GPU: work queued on `waiter` after this call starts only after
 * everything queued on `ctx` before it has finished; the host does not block. Lets independent kernels of one graph (the weight
 * and the data gradient of a layer) share the chip: each context has its own stream and scratch. */
int deepim_stream_wait(deepim_ctx* waiter, deepim_ctx* ctx);
/* HIP-event stopwatch on the context stream (bench.py's per-kernel timing) */
int deepim_timer_create(deepim_ctx* ctx, int* timer_id);
int deepim_timer_start(deepim_ctx* ctx, int timer_id);
int deepim_timer_stop(deepim_ctx* ctx, int timer_id);
int deepim_timer_elapsed_ms(deepim_ctx* ctx, int timer_id, float* ms);  /* syncs on stop event */
/* hipGraph capture of a launch sequence on the context stream */
int deepim_graph_begin(deepim_ctx* ctx);
int deepim_graph_end(deepim_ctx* ctx, int* graph_id);
int deepim_graph_launch(deepim_ctx* ctx, int graph_id);

/* ------------------------------------------------ multi-GPU: pose all-gather over RCCL/xGMI -- */
/* pairs shard across GPUs with no data-path collective except ONE all-gather of the refined poses per
 * refinement iteration, so that every rank/host holds all poses for the next render/update — the role of the per-device
 * executor group's merged outputs in the reference (deepim/core/DataParallelExecutorGroup.py:364-388, deepim/test.py:135).
 * One process per GPU; librccl.so is dlopen'ed on first use; no PyTorch. Bootstrap: rank 0 makes the id, ships the bytes
 * out of band (mx_deepim_amd/parallel.py does it over a TCP rendezvous), every rank calls deepim_comm_init. */
#define DEEPIM_COMM_ID_BYTES 128
int deepim_comm_unique_id(void* id_bytes /* DEEPIM_COMM_ID_BYTES, host */);
int deepim_comm_init(deepim_ctx* ctx, int rank, int world, const void* id_bytes);
int deepim_comm_destroy(deepim_ctx* ctx);
/* all_poses (world*B,3,4) device <- poses (B,3,4) device of every rank, rank-major; one asynchronous enqueue on the
 * context stream (a plain copy when no communicator is set) */
int deepim_allgather_poses(deepim_ctx* ctx, float* all_poses, const float* poses, int B);
/* in-place all-reduce of n device doubles on the context stream: op 0 = max, 1 = sum (bench.py's max-over-ranks time) */
int deepim_comm_allreduce_f64(deepim_ctx* ctx, double* buf, int n, int op);
/* What this process bound, as text: "backend=rccl|none;rccl_ranks=N;rccl_version=V;librccl_path=…;libamdhip64_path=…" — the
 * rank count is ncclCommCount of the context's communicator (0 without one), the paths are the files the ncclAllGather / hipMalloc
 * symbols in use live in (dladdr). Reporting only (bench.py's `comm` block); no reference counterpart. */
int deepim_comm_info(deepim_ctx* ctx, char* buf, int n);

/* ------------------------------------------------ F-group: depth warp / flow -- */
/* B2 drop-in. Same symbol, argument list and host-pointer contract as
 * lib/flow_c/gpu_flow.hpp:1-3 (`_flow`, defined gpu_flow_kernel.cu:82-148):
 * all pointers are HOST float32, caller-owned, synchronous. Unlike the
 * reference it aborts via deepim_last_error + stderr + nonzero `deepim_flow_status()`
 * when HIP fails. */
void _flow(float* flow, float* valid, float* depth_src, float* depth_tgt,
           float* KT, float* Kinv, int batch_size, int height, int width,
           int device_id);
/* The one body behind `_flow`.  gpu_flow.hpp:1-3 has C++ linkage (no extern "C") and the reference's Cython
 * extension is built language="c++" (gpu_flow.pyx:13-16, setup_linux.py:116-125), so it links the mangled
 * _Z5_flowPfS_S_S_S_S_iiii: the library exports that symbol too (csrc/flow_cxx.cpp; not declarable in this C header),
 * forwarding here like the C-linkage `_flow` above. */
void deepim_flow_host(float* flow, float* valid, float* depth_src, float* depth_tgt,
                      float* KT, float* Kinv, int batch_size, int height, int width,
                      int device_id);
int deepim_flow_status(void);
/* device-pointer variant of the same kernel (gpu_flow_kernel.cu:32-69) */
int deepim_flow_forward(deepim_ctx* ctx, float* flow, float* valid,
                        const float* depth_src, const float* depth_tgt,
                        const float* KT /*B,3,4 device*/, const float* Kinv_host /*3,3 HOST*/,
                        int B, int H, int W);
/* lib/pair_matching/flow.py:12-63 `calc_flow` semantics (un-rounded bounds test
 * is on rounded coords, valid needs depth_src != 0 and |d_tgt| > 1e-10);
 * KT = K·se3 (B,3,4), Kinv (3,3); flow channel order [dh,dw] unless standard_rep */
int deepim_calc_flow_forward(deepim_ctx* ctx, float* flow /*B,H,W,2*/, float* visible /*B,H,W*/,
                             const float* depth_src, const float* depth_tgt,
                             const float* KT, const float* Kinv_host, float thresh,
                             int standard_rep, int B, int H, int W);
/* Loader-side flow labels (lib/utils/image.py:402-450 get_pair_flow, called from get_data_pair_train_batch,
 * lib/pair_matching/data_pair.py:170-176): calc_flow(depth_rendered, pose_rendered, pose_observed, K, depth_observed)
 * per pair, laid out (B,2,H,W) as the loader hands it to the network, with flow_weights by TRAIN.FLOW_WEIGHT_TYPE:
 * weight_type 0 'all', 1 'viz', 2 'valid', tiled over the two channels. Poses (B,3,4) device, K host. */
int deepim_pair_flow_labels(deepim_ctx* ctx, float* flow /*B,2,H,W*/, float* flow_weights /*B,2,H,W*/,
                            const float* depth_rendered, const float* depth_observed, const float* pose_rendered,
                            const float* pose_observed, const float* K_host, float thresh, int standard_rep,
                            int weight_type, int B, int H, int W);
/* image.py:381-387: mask_rendered = depth_rendered with values > thresh set to 1 (others kept) */
int deepim_depth_clip_mask(deepim_ctx* ctx, float* mask, const float* depth, float thresh, size_t n);
/* deepim/operator_py/flow_updater.py:42-102 `FlowUpdater` forward: integer flow
 * from rounded+clamped coords; pose_src/pose_tgt (B,3,4); K, Kinv host 3x3 */
int deepim_flow_updater_forward(deepim_ctx* ctx, float* flow /*B,2,H,W*/, float* flow_weights /*B,2,H,W*/,
                                const float* depth_src, const float* depth_tgt,
                                const float* pose_src, const float* pose_tgt,
                                const float* K_host, float thresh, int wh_rep,
                                int B, int H, int W);
/* batch_updater_py_multi.py:255-265: KT = K·(pose_tgt ∘ pose_src⁻¹) and
 * mask_rendered = depth_rendered > 0.2 */
int deepim_calc_KT(deepim_ctx* ctx, float* KT /*B,3,4*/, const float* pose_src,
                   const float* pose_tgt, const float* K_host, int B);
int deepim_depth_to_mask(deepim_ctx* ctx, float* mask, const float* depth, float thresh, size_t n);
/* "box_rendered" / "box_observed" rectangle of a mask (lib/pair_matching/data_pair.py:94-116, the INIT_MASK twin at
 * lib/utils/image.py:355-374): 1 inside [y_start:y_end, x_start:x_end] with start/end = first/last row and column holding a
 * non-zero — numpy slices, so the last row and column stay 0. An empty mask gives zeros and sets bit 2 (value 4) of the status word
 * (deepim_zoom_status) where the reference raises. mask, box: (B,1,H,W) device. */
int deepim_mask_box_forward(deepim_ctx* ctx, float* box, const float* mask, int B, int H, int W);

/* --------------------------------------------------- Z-group: zoom / warp -- */
/* zoom_mask.py:29-112 `ZoomMaskOperator.forward`.
 * K_host: 3x3 float32 (host). Outputs are rounded resampled masks + zoom_factor (B,4). */
int deepim_zoom_mask_forward(deepim_ctx* ctx,
                             const float* mask_observed, const float* mask_gt_observed,
                             const float* mask_rendered, const float* src_pose /*B,3,4*/,
                             const float* K_host,
                             float* zoom_mask_observed, float* zoom_mask_gt_observed,
                             float* zoom_mask_rendered, float* zoom_factor /*B,4*/,
                             int B, int H, int W);
/* zoom_image.py:26-107 `ZoomImageOperator.forward` (no-mask variant). pixel_means_host: 3 floats
 * in the channel order of the image tensor. */
int deepim_zoom_image_forward(deepim_ctx* ctx,
                              const float* image_observed, const float* image_rendered,
                              const float* src_pose, const float* K_host,
                              const float* pixel_means_host,
                              float* zoom_image_observed, float* zoom_image_rendered,
                              float* zoom_factor, int B, int H, int W);
/* zoom_image_with_factor.py:31-65 */
int deepim_zoom_image_with_factor_forward(deepim_ctx* ctx, const float* zoom_factor,
                                          const float* image_observed, const float* image_rendered,
                                          const float* pixel_means_host, int high_light_center,
                                          float* zoom_image_observed, float* zoom_image_rendered,
                                          int B, int H, int W);
/* zoom_depth.py:24-44 */
int deepim_zoom_depth_forward(deepim_ctx* ctx, const float* zoom_factor,
                              const float* depth_observed, const float* depth_rendered,
                              float* zoom_depth_observed, float* zoom_depth_rendered,
                              int B, int H, int W);
/* zoom_flow.py:28-71; flow_weights/zoom_flow_weights may be NULL when b_inv_zoom */
int deepim_zoom_flow_forward(deepim_ctx* ctx, const float* zoom_factor, const float* flow,
                             const float* flow_weights, float* zoom_flow, float* zoom_flow_weights,
                             int b_inv_zoom, int B, int H, int W);
/* zoom_mask_with_factor.py:29-64 */
int deepim_zoom_mask_with_factor_forward(deepim_ctx* ctx, const float* zoom_factor, const float* mask,
                                         float* zoom_mask, int b_inv_zoom, int B, int H, int W);
/* zoom_trans.py:22-46 forward, :48-74 backward */
int deepim_zoom_trans_forward(deepim_ctx* ctx, const float* zoom_factor, const float* trans_delta,
                              float* zoom_trans_delta, int b_inv_zoom, int B);
int deepim_zoom_trans_backward(deepim_ctx* ctx, const float* zoom_factor, const float* out_grad,
                               float* in_grad, int b_inv_zoom, int b_zoom_grad, int B);
/* Fused front end of the test graph (deepIM_flownet.py:33-62 + :563-622): ZoomMask +
 * ZoomImageWithFactor [+ ZoomDepth] + `/255` + Concat written straight into the
 * conv1 input (B,C,H,W), C = 8 (+2 with depth). Also returns zoom_factor. depth_* may be NULL.
 * With both masks NULL (INPUT_MASK=False, deepIM_flownet.py:594-605) the factor comes from ZoomImage's
 * non-black-pixel boxes and C = 6 (+2 with depth). */
int deepim_zoom_concat_forward(deepim_ctx* ctx,
                               const float* image_observed, const float* image_rendered,
                               const float* mask_observed, const float* mask_rendered,
                               const float* depth_observed, const float* depth_rendered,
                               const float* src_pose, const float* K_host,
                               const float* pixel_means_host,
                               float* net_input, float* zoom_factor,
                               int B, int H, int W);
/* the same front end for the shipped 8-channel input (masks, no depth) writing channel-blocked records: net_input_nc8 is
 * (B,H,W,8) = the "NC8" layout [n][C/8][h][w][8] with C = 8, which deepim_conv2d_forward_ex(in_nc8 = 1) reads for conv1;
 * element values identical to deepim_zoom_concat_forward's (B,8,H,W) */
int deepim_zoom_concat_forward_nc8(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                   const float* mask_observed, const float* mask_rendered, const float* src_pose,
                                   const float* K_host, const float* pixel_means_host, float* net_input_nc8,
                                   float* zoom_factor, int B, int H, int W);
/* the front end of the fp16 conv path (config 5): the same values rounded once to fp16 and written as the pixel records
 * deepim_conv1_f16_h16_forward reads — main8 (B,H,W,8 halves) = net-input channels 0-7; with depth_* set (INPUT_DEPTH, C = 10)
 * extra2 (B,H,W,2 halves) = channels 8-9 (the masks). depth_observed, depth_rendered and extra2 are all NULL or all set. */
int deepim_zoom_concat_forward_h16(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                   const float* mask_observed, const float* mask_rendered, const float* depth_observed,
                                   const float* depth_rendered, const float* src_pose, const float* K_host,
                                   const float* pixel_means_host, void* main8, void* extra2, float* zoom_factor, int B,
                                   int H, int W);
/* the same front end in the TRAINING graph (deepIM_flownet.py:392-412): the zoom region comes from mask_gt_observed
 * (B,1,H,W; NULL = mask_observed, i.e. the test graph) */
int deepim_zoom_concat_train_forward(deepim_ctx* ctx, const float* image_observed, const float* image_rendered,
                                     const float* mask_observed, const float* mask_gt_observed, const float* mask_rendered,
                                     const float* depth_observed, const float* depth_rendered, const float* src_pose,
                                     const float* K_host, const float* pixel_means_host, float* net_input,
                                     float* zoom_factor, int B, int H, int W);
/* parity hook: runs all 2^32 float bit patterns through the kernel's 5-op replacement of `v / 255.0f`
 * (deepIM_flownet.py:35-36 divides the zoomed images by 255) against the IEEE division on the device and returns the
 * number of patterns whose results differ in any bit (NaN results compare equal). Must be 0. */
int deepim_selfcheck_div255(deepim_ctx* ctx, unsigned long long* mismatches_host);
/* debug/parity hook: the int32 source indices (x0,y0 = floor of the sampling
 * position) the resampler uses for every output pixel: idx (B,2,H,W) int32 */
int deepim_zoom_indices(deepim_ctx* ctx, const float* zoom_factor, int32_t* idx, int B, int H, int W);
/* debug/parity hook: the affine (wx,wy,tx,ty) the b_inv_zoom ops sample with, i.e. zoom_flow.py:36-44 /
 * zoom_mask_with_factor.py:43-52 applied to a stored factor (NumPy-1.x promotion: float64 chain, rounded once).
 * zoom_factor, inv_factor: (B,4) device */
int deepim_zoom_inverse_factor(deepim_ctx* ctx, const float* zoom_factor, float* inv_factor, int B, int H, int W);
/* sticky status word since the last read (reading clears it):
 * bit0 (1) = a zoom-factor computation saw an observed mask/image with no valid pixel — the reference raises ValueError
 *            there (np.min of an empty array, zoom_mask.py:55); the zoom factor is NaN for that sample
 * bit1 (2) = GroupPicker index out of range
 * bit2 (4) = deepim_mask_box_forward / deepim_render_update_forward saw an empty mask (data_pair.py:98 raises)
 * bit3 (8) = split-fp16 conv path: a value left fp16's range after scaling (|v·scale| > 60000 or NaN) and was clamped — the
 *            results of that call are not fp32-grade; lower the activation scale or use the fp32 path */
int deepim_zoom_status(deepim_ctx* ctx, int* status);

/* ------------------------------------------ N-group: matching network ops -- */
/* MXNet Convolution (+bias) [+LeakyReLU slope] (deepIM_flownet.py:63-107,123,145,176,317).
 * `packed_w` comes from deepim_conv_pack_weights (one-time re-layout of the
 * (Cout,Cin,kh,kw) tensor into MFMA tile order; size from deepim_conv_packed_size).
 * slope == 1.0f means "no activation". out may have more channels than Cout
 * (out_ctotal, written at channel offset out_coff) so Concat is free. */
size_t deepim_conv_packed_size(int Cout, int Cin, int kh, int kw);
int deepim_conv_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w /*Cout,Cin,kh,kw dev*/,
                             int Cout, int Cin, int kh, int kw);
/* the same with a choice of operand orders: bits 1 = LDS-kernel order, 2 = NCHW register-fed order, 4 = NC8 order. A caller
 * that runs NCHW activations only (the training graph and its data gradients) passes 3 and saves a third of the re-pack
 * after every SGD step; a left-out order must not be used by the forward call (NC8 input needs bit 4). */
int deepim_conv_pack_weights_ex(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin, int kh, int kw, int orders);
/* packed weights of a DATA GRADIENT straight from the layer's raw tensor w_layer (Co_l,Ci_l,kh_l,kw_l): the packed convolution has
 * Cout = Ci_l, Cin = Co_l, an nky x nkx kernel with tap (a,b) = layer tap (ky0 + st·(nky-1-a), kx0 + st·(nkx-1-b)). st = 1 with
 * the whole kernel = transposed + flipped weights (replaces deepim_conv_flip_weights + pack); st = 2 = one output parity class of
 * a stride-2 layer (replaces deepim_conv_subkernel_flip + pack). orders: bits 1 | 2. */
int deepim_conv_pack_dgrad(deepim_ctx* ctx, float* packed_w, const float* w_layer, int Co_l, int Ci_l, int kh_l, int kw_l, int ky0,
                           int kx0, int st, int nky, int nkx, int orders);
/* the one order (1 or 2) deepim_conv2d_forward will read for this geometry under the context's current options */
int deepim_conv_weight_order(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad);
int deepim_conv2d_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                          const float* bias, int B, int Cin, int H, int W, int Cout,
                          int kh, int kw, int stride, int pad, float slope,
                          int out_ctotal, int out_coff);
/* Same convolution with channel-blocked ("NC8") activations between layers: a tensor (B,C,H,W) stored as
 * [n][C/8][h][w][8] (C % 8 == 0). in_nc8 / out_nc8 select the layout of `in` / `out` (0 = NCHW). NC8 input runs on the
 * LDS-free kernel whose fmaf chain per output runs over (c/8, ky, kx, s, h) with channel 8(c/8) + s + 4h — one 16-byte
 * load per lane feeds four MFMA k-steps; it needs Cout > 64. The encoder uses NCHW -> NC8 for conv1, NC8 -> NC8 up to
 * conv6, NC8 -> NCHW for conv6_1 (fc6 wants MXNet's flatten order). deepim_relayout_nc8 converts a tensor either way. */
int deepim_conv2d_forward_ex(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                             const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                             int stride, int pad, float slope, int out_ctotal, int out_coff, int in_nc8, int out_nc8);
int deepim_relayout_nc8(deepim_ctx* ctx, float* dst, const float* src, int B, int C, size_t hw, int to_nc8);
/* The same conversion (NC8 -> NCHW) written into channels [dst_coff, dst_coff + C) of a dst_ctotal-channel NCHW tensor: the encoder
 * skip connections of the refinement decoder (Concat, deepim/symbols/deepIM_flownet.py:128-131, :143-146) in one pass. */
int deepim_relayout_nc8_slice(deepim_ctx* ctx, float* dst, int dst_ctotal, int dst_coff, const float* src_nc8, int B, int C, size_t hw);
/* The encoder's 3x3 stride-1 pad-1 layers (conv3_1 / conv4_1 / conv5_1 / conv6_1, deepIM_flownet.py:69-101) as fp32 Winograd
 * F(2x2,3x3): the same fp32 arithmetic with 2.25x fewer multiplies, a different summation (NOT the direct kernels' fmaf chain:
 * within 1e-5 of the layer's range, tests/test_gpu_wino.py). `in` is NC8; `out` NC8 (out_nc8 = 1) or channels [out_coff,
 * out_coff + Cout) of an NCHW tensor with out_ctotal channels (0 = Cout). Cout % 32 == 0, Cin % 8 == 0. packed_w: U = G g G^T
 * from deepim_conv_wino_pack_weights (16 floats per weight tap set: Cout*Cin*64 bytes). */
size_t deepim_conv_wino_packed_size(int Cout, int Cin);
/* 1 when the layer should take the Winograd path: Cout % 64 == 0 (shared-transform kernel, splits the input channels on small grids)
 * from 64 tiles on; other channel counts (one-wave kernel, no split) from 128 blocks of 32 channels x 128 tiles on; always 0 on a
 * context in the canonical-summation-order configuration ("conv_max_split" = 1). ctx may be NULL (the defaults). Which layers
 * qualify — and the split plan — depend on B: so does the rounding of a sample's output, never its 1e-5-of-range bound. */
int deepim_conv_wino_preferred(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout);
/* the same for a 5x5 stride-2 pad-2 layer (B, Cin, H, W) run over its space-to-depth form (one-wave kernel: from 256 blocks on) */
int deepim_conv_wino_preferred_s2d(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout);
/* The launch plan the shared-transform kernel would use for a layer under the context's options (no launch; tests and docs read it):
 * arguments as deepim_conv2d_wino_forward sees them (s2d = 1: Cin, H, W of the space-to-depth problem). plan[0] block shape (0 = 64
 * channels x 64 tiles, 1 = 128 x 32, 2 = 64 x 32 on four waves), [1] grid, [2] K slices S, [3] K steps per slice, [4] stream-K granules
 * per tile block (0 = off: whole tile blocks per block), [5] granules of the last round per persistent block, [6] whole tile blocks
 * per persistent block before them, [7] tile blocks of the layer (incl. the padding of the XCD deal), [8] the first so many blocks of an
 * XCD take one granule more. All -1 where another kernel runs the layer. Host arithmetic only: ctx = NULL asks for the plan under
 * the default options (no device needed). */
int deepim_conv_wino_plan(deepim_ctx* ctx, int B, int Cin, int H, int W, int Cout, int out_nc8, int s2d, int* plan /*9, host*/);
int deepim_conv_wino_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w /*Cout,Cin,3,3 dev*/, int Cout, int Cin);
/* The 5x5 stride-2 pad-2 layers (conv2 / conv3, deepIM_flownet.py:65-68) on the same kernel: a stride-2 convolution is a stride-1
 * convolution over the four input phases (x[2m + py][2l + px] as channel (py*2+px)*Cin + c of a (4 Cin, H/2, W/2) tensor) with the
 * taps w[2a+py][2b+px] — a 3x3 kernel, zero where the index reaches 5. The producer writes that "space-to-depth" NC8 tensor itself
 * (out_nc8 = 3 of deepim_conv2d_forward_ex / deepim_conv2d_wino_forward; deepim_relayout_nc8_s2d converts for tests), this packs
 * the layer's own (Cout, Cin, 5, 5) weights for it (size: deepim_conv_wino_packed_size(Cout, 4*Cin)), and the layer runs as
 * deepim_conv2d_wino_forward(ctx, out, in_s2d, packed, bias, B, 4*Cin, H/2, W/2, Cout, ...). 1.56x fewer multiplies than direct. */
int deepim_conv_wino_pack_weights_s2d(deepim_ctx* ctx, float* packed_w, const float* w /*Cout,Cin,5,5 dev*/, int Cout, int Cin);
/* the stride-2 layer in one call: (B, Cin, H, W) input given in its space-to-depth NC8 form, output (B, Cout, H/2, W/2); the positions
 * whose transformed weights are identically zero (third kernel row / column of the odd input phases) are skipped: 49 of 64 MFMAs.
 * The shared-transform kernel (Cout % 64 == 0, 4 Cin % 64 == 0) walks the four input phases interleaved — two 8-channel blocks of each
 * per loop body — so that the skipped positions are a compile-time property of every step: the same products, the channel sum in
 * that order (2e-6 of range from the one-wave kernel's natural order). */
int deepim_conv2d_wino_forward_s2d(deepim_ctx* ctx, float* out, const float* in_s2d, const float* packed_w, const float* bias,
                                   int B, int Cin, int H, int W, int Cout, float slope, int out_nc8, int out_ctotal, int out_coff);
int deepim_relayout_nc8_s2d(deepim_ctx* ctx, float* dst, const float* src, int B, int C, int H, int W, int to_s2d);
int deepim_conv2d_wino_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias,
                               int B, int Cin, int H, int W, int Cout, float slope, int out_nc8, int out_ctotal, int out_coff);
/* The same 3x3 stride-1 pad-1 Convolution + bias + LeakyReLU layers (deepIM_flownet.py:77-101) as fp32 Winograd F(4,3) x F(2,3): 4-row x
 * 2-column output tiles, 24 positions, 3 multiply-adds per output, input and output channel instead of 4 (csrc/wino42.hip; Cout % 64 == 0,
 * Cin % 8 == 0; channel-blocked input, channel-blocked (out_nc8 = 1) or NCHW-slice (0) output). ~3e-6 of the layer's range from the direct
 * sum (bar 1e-5). Packed size: 24 floats per (output, input channel) pair. Measurements and the reason for this tile: profiles/r06_wino44.md. */
size_t deepim_conv_wino42_packed_size(int Cout, int Cin);
int deepim_conv_wino42_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int Cout, int Cin);
int deepim_conv2d_wino42_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias,
                                 int B, int Cin, int H, int W, int Cout, float slope, int out_nc8, int out_ctotal, int out_coff);
/* fp16 conv path (BASELINE config 5): NHWC fp16 activations, fp16 weights (packed once), fp16 matrix cores
 * with fp32 accumulation, bias + LeakyReLU in fp32, NHWC fp16 output. Same layer semantics as
 * deepim_conv2d_forward (deepIM_flownet.py:63-107); tolerance documented in DESIGN.md (fp16 cannot meet 1e-4).
 * Cin_pad = Cin rounded up to a multiple of 8 (zero channels); Cout % 4 == 0. */
int deepim_nchw_f32_to_nhwc_f16(deepim_ctx* ctx, void* out_f16, const float* in, int B, int C, int H, int W, int Cpad);
int deepim_nhwc_f16_to_nchw_f32(deepim_ctx* ctx, float* out, const void* in_f16, int B, int C, int H, int W);
size_t deepim_conv_f16_packed_size(int Cout, int Cin_pad, int kh, int kw);
int deepim_conv_f16_pack_weights(deepim_ctx* ctx, void* packed, const float* w /*Cout,Cin,kh,kw dev f32*/,
                                 int Cout, int Cin, int Cin_pad, int kh, int kw);
int deepim_conv2d_f16_forward(deepim_ctx* ctx, void* out_nhwc_f16, const void* in_nhwc_f16, const void* packed_w,
                              const float* bias, int B, int Cin_pad, int H, int W, int Cout, int kh, int kw,
                              int stride, int pad, float slope);
/* Split-fp16 ("x3") convolution: the same Convolution + bias + LeakyReLU (deepIM_flownet.py:63-107) at fp32-grade accuracy on
 * the fp16 matrix cores. A value v travels as the fp16 pair hi = f16(v·s), lo = f16(v·s − hi) (22 significand bits, s a power
 * of two), a product is hi·hi + hi·lo + lo·hi (v_mfma_f32_32x32x16_f16, fp32 accumulation; the dropped lo·lo term is 2^-22
 * relative). Tensors are "split16" NHWC: per pixel, per 16 channels, 32 halves [hi 0..15 | lo 0..15].
 * acc_scale = 1 / (s_in · s_w) returns the accumulator to real units before bias; out_scale = s of the stored output.
 * Needs Cin % 32 == 0, Cout % 128 == 0 (conv2 … conv6_1 of the encoder; conv1: deepim_conv1_x3_forward, or the fp32 conv with a
 * split16 epilogue for other channel counts). Inputs beyond 2 GiB run as sub-batches. A value that leaves fp16's range after
 * scaling is clamped and reported through bit 3 of the status word (deepim_zoom_status). Not bit-exact against the fp32 path:
 * every layer within 1e-5 of the float64-accumulating oracle (tests/test_gpu_x3.py). */
int deepim_nchw_f32_to_split16(deepim_ctx* ctx, void* out_split16, const float* in, int B, int C, int H, int W, float scale);
int deepim_split16_to_nchw_f32(deepim_ctx* ctx, float* out, const void* in_split16, int B, int C, int H, int W, float inv_scale);
size_t deepim_conv_x3_packed_size(int Cout, int Cin, int kh, int kw);
int deepim_conv_x3_pack_weights(deepim_ctx* ctx, void* packed, const float* w /*Cout,Cin,kh,kw dev f32*/, int Cout, int Cin,
                                int kh, int kw, float w_scale);
/* conv1 of that encoder for channel counts other than 8: the fp32 MFMA convolution (NCHW fp32 in, exact products) writing
 * split16 from its epilogue; Cout % 16 == 0, even Cin */
int deepim_conv2d_forward_split16(deepim_ctx* ctx, void* out_split16, const float* in, const float* packed_w, const float* bias,
                                  int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, float slope,
                                  float out_scale);
/* conv1 (Cin 8, 7x7, stride 2, pad 3, Cout 64; deepIM_flownet.py:63-67) in split fp16 straight from the NCHW fp32 net input:
 * persistent LDS-patch kernel, two taps per 16-wide MFMA k-step. in_scale: scale applied to the input before the split;
 * acc_scale = 1 / (in_scale · w_scale). Output split16 (B,Ho,Wo,64 ch). */
size_t deepim_conv1_x3_packed_size(void);
int deepim_conv1_x3_pack_weights(deepim_ctx* ctx, void* packed, const float* w /*64,8,7,7*/, float w_scale);
int deepim_conv1_x3_forward(deepim_ctx* ctx, void* out_split16, const float* in /*B,8,H,W*/, const void* packed_w,
                            const float* bias, int B, int H, int W, float slope, float in_scale, float acc_scale,
                            float out_scale);
/* the same kernel with plain fp16 operands = conv1 of the fp16 path (config 5): NCHW fp32 in → NHWC fp16 (B,Ho,Wo,64) out;
 * packed_w = the first half (hi parts) of deepim_conv1_x3_pack_weights(..., w_scale = 1) */
int deepim_conv1_f16_forward(deepim_ctx* ctx, void* out_nhwc_f16, const float* in /*B,8,H,W*/, const void* packed_w,
                             const float* bias, int B, int H, int W, float slope);
/* conv1 of the fp16 path for BASELINE config 5's RGB-D net input (INPUT_DEPTH: Cin = 10, deepIM_flownet.py:33-62): channels
 * 0-7 as above, channels 8-9 as 14 pseudo taps of 4 consecutive columns x 2 channels (7 extra k-steps instead of the 24 of a
 * 16-channel padding).  w (64,10,7,7) fp32 → packed (deepim_conv1_f16_c10_packed_size bytes). */
size_t deepim_conv1_f16_c10_packed_size(void);
int deepim_conv1_f16_c10_pack_weights(deepim_ctx* ctx, void* packed, const float* w /*64,10,7,7*/);
int deepim_conv1_f16_c10_forward(deepim_ctx* ctx, void* out_nhwc_f16, const float* in /*B,10,H,W*/, const void* packed_w,
                                 const float* bias, int B, int H, int W, float slope);
/* conv1 of the fp16 path from fp16 pixel records (deepim_zoom_concat_forward_h16): extra2 == NULL → 8-channel input, packed_w as
 * for deepim_conv1_f16_forward; extra2 set → the 10-channel RGB-D input, packed_w from deepim_conv1_f16_c10_pack_weights */
int deepim_conv1_f16_h16_forward(deepim_ctx* ctx, void* out_nhwc_f16, const void* main8, const void* extra2,
                                 const void* packed_w, const float* bias, int B, int H, int W, float slope);
int deepim_conv2d_x3_forward(deepim_ctx* ctx, void* out_split16, const void* in_split16, const void* packed_w, const float* bias,
                             int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, float slope,
                             float acc_scale, float out_scale);
/* MXNet Deconvolution k4 s2 p0 (+bias) + Crop(offset 1,1 → Ho,Wo) [+LeakyReLU]
 * (deepIM_flownet.py:127-143,149-165). w layout (Cin,Cout,4,4). */
size_t deepim_deconv_packed_size(int Cin, int Cout);
int deepim_deconv_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int Cin, int Cout);
int deepim_deconv4x4s2_crop_forward(deepim_ctx* ctx, float* out, const float* in, const float* packed_w,
                                    const float* bias, int B, int Cin, int H, int W, int Cout,
                                    int Ho, int Wo, int crop_y, int crop_x, float slope,
                                    int out_ctotal, int out_coff);
/* grouped (depthwise) Deconvolution k32 s16 no-bias + Crop(offset) to (Ho,Wo), times `scale`
 * (deepIM_flownet.py:185-200,326-340,636-648,687-702). w (C,1,32,32). */
int deepim_upsample16_crop_forward(deepim_ctx* ctx, float* out, const float* in, const float* w,
                                   int B, int C, int H, int W, int Ho, int Wo,
                                   int crop_y, int crop_x, float scale);
/* FullyConnected y = x·Wᵀ + b [+LeakyReLU] (deepIM_flownet.py:112-116,211-215) */
int deepim_fc_forward(deepim_ctx* ctx, float* out, const float* in, const float* w /*O,I*/,
                      const float* bias, int B, int I, int O, float slope);
/* fc6 on the fp32 matrix cores, one pass over the weights for up to 32 batch rows (v_mfma_f32_32x32x2_f32 split-K GEMM
 * with a fixed-order reduction): the (O,I) MXNet weight is re-packed once into MFMA operand order, then
 * out (B,O) = lrelu(in (B,I) · wᵀ + bias). Built for O == 256 (fc6, fc7), I % 8 == 0. */
size_t deepim_fc_packed_size(int O, int I);
int deepim_fc_pack_weights(deepim_ctx* ctx, float* packed_w, const float* w, int O, int I);
int deepim_fc_forward_packed(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, const float* bias,
                             int B, int I, int O, float slope);
/* fc7-input → rot(4), trans(3) FCs + ZoomTrans(inverse) + Concat → se3 (B,7)
 * (deepIM_flownet.py:715-726) */
int deepim_pose_head_forward(deepim_ctx* ctx, float* se3 /*B,7*/, const float* feat /*B,F*/,
                             const float* w_rot, const float* b_rot,
                             const float* w_trans, const float* b_trans,
                             const float* zoom_factor, int B, int F);

/* --------------------------------------------- S-group: SE(3) pose update -- */
/* RT_transform (lib/pair_matching/RT_transform.py:127-151) batched: pose_est (B,3,4) f32 from
 * pose_src (B,3,4), se3 (B,7) = [quat(4) | trans(3)]. Computed in float64 like the reference,
 * stored float32. rot_coord: 0 MODEL, 1 CAMERA, 2 CAMERA_NEW, 3 NAIVE. pose_est64 optional (B,3,4) f64. */
int deepim_rt_transform(deepim_ctx* ctx, float* pose_est, double* pose_est64, const float* pose_src,
                        const float* se3, const float* T_means_host, const float* T_stds_host,
                        int rot_coord, int B);
/* calc_RT_delta (lib/pair_matching/RT_transform.py:16-44, rot_type QUAT) batched: the ground-truth
 * (rot (B,4) w>=0, trans (B,3)) labels between a source and a target pose — R_inv_transform (:64-71),
 * T_inv_transform (:105-124), mat2quat (:432-509, largest eigenvector of the 4x4 K matrix; Jacobi sweeps in
 * float64 here). Used by the loader (data_pair.py:188-195) and the batch updater (:239-246). */
int deepim_calc_rt_delta(deepim_ctx* ctx, float* rot /*B,4*/, float* trans /*B,3*/, const float* pose_src,
                         const float* pose_tgt, const float* T_means_host, const float* T_stds_host,
                         int rot_coord, int B);
/* The tail of one test-graph refinement iteration in ONE launch: fc7 (FullyConnected 256 → 256 + LeakyReLU,
 * deepIM_flownet.py:114-116) → rot / trans FullyConnected + inverse ZoomTrans → se3 (:715-726, zoom_trans.py:22-46) →
 * RT_transform (lib/pair_matching/RT_transform.py:127-151, quaternion form) — the same sums in the same order as
 * deepim_fc_forward + deepim_pose_head_forward + deepim_rt_transform, hence bit-identical to them. fc7_out (B,256), se3 (B,7),
 * pose_est (B,3,4; may be pose_src: in-place update), fc6 (B,256); feat must be 256. */
int deepim_pose_tail_forward(deepim_ctx* ctx, float* fc7_out, float* se3, float* pose_est, const float* fc6,
                             const float* w_fc7, const float* b_fc7, const float* w_rot, const float* b_rot,
                             const float* w_trans, const float* b_trans, const float* zoom_factor,
                             const float* pose_src, const float* T_means_host, const float* T_stds_host, int rot_coord,
                             int B, int feat, float slope);
/* RT_transform with an Euler-angle rotation (ROT_TYPE EULER: r.shape[0] == 3 → euler2mat(r0, r1, r2), static xyz axes,
 * RT_transform.py:130-131, :240-307): euler_trans (B,6) = [euler(3) | trans(3)]. */
int deepim_rt_transform_euler(deepim_ctx* ctx, float* pose_est, double* pose_est64, const float* pose_src,
                              const float* euler_trans, const float* T_means_host, const float* T_stds_host,
                              int rot_coord, int B);
/* calc_RT_delta with the reference's rot_type switch (RT_transform.py:34-41): 0 QUAT → rot (B,4), 1 EULER (mat2euler
 * sxyz, :310-373) → rot (B,3), 2 MATRIX → rot (B,9). */
int deepim_calc_rt_delta_ex(deepim_ctx* ctx, float* rot, float* trans, const float* pose_src, const float* pose_tgt,
                            const float* T_means_host, const float* T_stds_host, int rot_coord, int rot_type, int B);
/* the rotation converters of RT_transform.py as batched ops, float64 results like the reference's:
 * op 0 quat2mat (:383-429; in (B,4) → out (B,9)), 1 mat2quat (:432-509; (B,9) → (B,4), w >= 0),
 * 2 euler2mat sxyz (:240-307; (B,3) → (B,9)), 3 mat2euler sxyz (:310-373; (B,9) → (B,3)) */
int deepim_rot_convert(deepim_ctx* ctx, double* out, const float* in, int op, int B);
/* get_point_cloud_observed (lib/pair_matching/data_pair.py): out (B,3,N) = R·points + T for pose (B,3,4) */
int deepim_points_transform(deepim_ctx* ctx, float* out, const float* points, const float* pose, int B, int N);
/* calc_se3 (RT_transform.py:176-187): se3_mul(pose_tgt, se3_inverse(pose_src)) in float32 (lib/utils/projection.py:12-43)
 * → rotm (B,3,3), t (B,3) */
int deepim_calc_se3(deepim_ctx* ctx, float* rotm, float* t, const float* pose_src, const float* pose_tgt, int B);
/* Pose-error metrics of lib/utils/pose_error.py (used by the evaluate_pose functions of LM6D_REFINE.py:278-512 and by
 * tester.py:401): per pair out[b] = { re (:118-124, geodesic angle in degrees; == calc_rt_dist_m's rd_deg),
 * te (:127-145, ||t_gt - t_est||), add (:55-69), adi (:72-88, nearest-neighbour mean, brute force instead of a
 * KD-tree), arp_2d (:36-52, mean reprojection distance in pixels) }. points: (B,3,N) model points or, with
 * points_shared != 0, one (3,N) set for all pairs. float64 accumulation, float32 results (B,5). */
int deepim_pose_error(deepim_ctx* ctx, float* out /*B,5*/, const float* pose_est, const float* pose_gt,
                      const float* points, int points_shared, const float* K_host, int B, int N);
/* Transform3D forward/backward (deepim/operator_py/transform3d.py:34-151) */
int deepim_transform3d_forward(deepim_ctx* ctx, float* out /*B,3,N*/, const float* points /*B,3,N*/,
                               const float* rotation /*B,4*/, const float* translation /*B,3*/,
                               const float* pose_src, const float* T_means_host,
                               const float* T_stds_host, int rot_coord, int B, int N);
int deepim_transform3d_backward(deepim_ctx* ctx, float* d_rotation /*B,4*/, float* d_translation /*B,3*/,
                                const float* out_grad /*B,3,N*/, const float* points,
                                const float* rotation, const float* translation,
                                const float* pose_src, const float* T_means_host,
                                const float* T_stds_host, int rot_coord, int B, int N);

/* ---------------------------------------------------- H-group: heads/losses -- */
/* point-matching loss (deepIM_flownet.py:265-312): per-element
 * loss = weights · f((est − gt)/normalize), f ∈ {0: L1 |x|, 1: L2 x², 2: smooth-L1(σ)};
 * writes loss elements (B,3,N), the batch sum and d loss/d est (scaled by grad_scale) */
int deepim_point_matching_loss(deepim_ctx* ctx, float* loss /*B,3,N*/, float* loss_sum /*1*/,
                               float* d_est /*B,3,N or NULL*/, const float* est, const float* gt,
                               const float* weights /*B,3,N or NULL*/, float normalize,
                               int loss_type, float sigma, float grad_scale, int B, int N);
/* flow loss (deepIM_flownet.py:201-207): loss = w·(est − gt/normalize_flow)² */
int deepim_flow_loss(deepim_ctx* ctx, float* loss /*n*/, float* loss_sum, float* d_est,
                     const float* est, const float* gt, const float* weights,
                     float normalize_flow, float grad_scale, size_t n);
/* MXNet L2Normalization, instance mode (deepIM_flownet.py:217 `normalize_quat`): out = x / sqrt(Σx² + eps) */
int deepim_l2_normalize_forward(deepim_ctx* ctx, float* out, const float* in, int B, int D, float eps);
int deepim_l2_normalize_backward(deepim_ctx* ctx, float* d_in, const float* d_out, const float* in, int B, int D,
                                 float eps);
/* rotation distance loss (deepIM_flownet.py:238-248): loss_b = 1 − (q_gt·q_est)², d_q_est = −2(q_gt·q_est)·q_gt·grad_scale.
 * The translation distance loss (:250-262) is deepim_point_matching_loss with N = 1, normalize = 1, no weights. */
int deepim_rot_dist_loss(deepim_ctx* ctx, float* loss /*B*/, float* d_q_est /*B,4 or NULL*/, const float* q_gt,
                         const float* q_est, float grad_scale, int B);
/* mask head test path (deepIM_flownet.py:647-666): sigmoid → ZoomMaskWithFactor(inverse)
 * → round. `logits` is the cropped upsampled mask_conv3 output (B,1,H,W). prob optional. */
int deepim_mask_head_forward(deepim_ctx* ctx, float* mask_pred /*B,1,H,W*/, float* prob /*or NULL*/,
                             const float* logits, const float* zoom_factor, int B, int H, int W);
/* mask loss (deepIM_flownet.py:342-361): LogisticRegressionOutput forward = sigmoid,
 * backward = (p − y)·grad_scale */
int deepim_mask_logistic(deepim_ctx* ctx, float* prob, float* d_logits /*or NULL*/,
                         const float* logits, const float* label, float grad_scale, size_t n);
/* GroupPicker (deepim/operator_py/group_picker.py:22-56) */
int deepim_group_picker_forward(deepim_ctx* ctx, float* out /*B,C/G*/, const float* in /*B,C*/,
                                const float* group_idx /*B*/, int group_num, int B, int C);
int deepim_group_picker_backward(deepim_ctx* ctx, float* in_grad /*B,C*/, const float* out_grad,
                                 const float* group_idx, int group_num, int B, int C);

/* ------------------------------------ T-group: backward of the network + SGD -- */
/* what module.backward and the "sgd" optimizer do for the conv stack and the FC head in the reference's
 * training loop (deepim/core/module.py:1131-1137, deepim/train.py:295-338; layers at deepIM_flownet.py:63-116,211-215).
 * All tensors NCHW fp32 device. Reductions are deterministic (fixed order, no float atomics). */
/* LeakyReLU gradient from the saved OUTPUT y: dz = y > 0 ? dy : slope*dy */
int deepim_lrelu_backward(deepim_ctx* ctx, float* dz, const float* dy, const float* y, float slope, size_t n);
/* db[c] = sum over n, pixels of dz (B,C,hw) */
int deepim_bias_grad(deepim_ctx* ctx, float* db, const float* dz, int B, int C, size_t hw);
/* the two in one walk, with the gradient arriving over a skip connection folded in: dz = lrelu'(y)*(dy [+ add]) (B,C,hw; dz may
 * be dy; add may be NULL), db[c] = sum of dz — what the training graph needs per encoder layer */
int deepim_lrelu_bias_backward(deepim_ctx* ctx, float* dz, float* db, const float* dy, const float* add, const float* y,
                               float slope, int B, int C, size_t hw);
/* wt (Cin,Cout,kh,kw) = w (Cout,Cin,kh,kw) transposed and flipped: the weights with which the data gradient of a
 * convolution is itself a stride-1 convolution (pad kh-1-p) — run on deepim_conv2d_forward after deepim_conv_pack_weights */
int deepim_conv_flip_weights(deepim_ctx* ctx, float* wt, const float* w, int Cout, int Cin, int kh, int kw);
/* Data gradient of a stride-2 convolution as four stride-1 convolutions of the UN-dilated gradient, one per output parity
 * class (py, px): the class only meets the taps ky = ky0 + 2a, kx = kx0 + 2b with ky0 = (py + pad) % 2, kx0 = (px + pad) % 2.
 * wt (Cin,Cout,nky,nkx) = that sub-kernel of w (Cout,Cin,kh,kw), transposed and flipped → deepim_conv_pack_weights →
 * deepim_conv2d_forward(stride 1, pad P) → deepim_interleave2d puts the window [cy, cy + hq) x [cx, cx + wq) of the result on
 * dx[.., 2i + py, 2j + px] (the training loop's module.backward, deepim/core/module.py:1131-1137, for conv2/3/4/5/6). */
int deepim_conv_subkernel_flip(deepim_ctx* ctx, float* wt, const float* w, int Cout, int Cin, int kh, int kw, int ky0, int kx0,
                               int nky, int nkx);
int deepim_interleave2d(deepim_ctx* ctx, float* dx /*BC,H,W*/, const float* src /*BC,Hs,Ws*/, int BC, int Hs, int Ws, int cy,
                        int cx, int H, int W, int py, int px);
/* the class convolution and the interleave in one: the final stores of the conv kernels (and of their split-K second pass)
 * put the window of the result on out (B,Cout,Hd,Wd)[.., 2i + py, 2j + px] — no class buffer */
int deepim_conv2d_forward_remap(deepim_ctx* ctx, float* out, const float* in, const float* packed_w, int B, int Cin, int H, int W,
                                int Cout, int kh, int kw, int pad, int cy, int cx, int Hd, int Wd, int py, int px);
/* The whole data gradient of a stride-2 convolution (kernel k x k, pad): dx (B,Ci_l,Hd,Wd) from dz (B,Co_l,Ho,Wo), Ho = (Hd + 2 pad
 * - k)/2 + 1, and the layer's RAW weights w_layer (Co_l,Ci_l,k,k). The four parity classes above are packed, convolved and
 * reduced TOGETHER — one launch each, the blocks of the convolution launch shared out over the classes — when the register-fed
 * kernel takes the geometry (even Co_l; 64 x 256 tiles for Ci_l <= 64, else 128 x 128); class by class otherwise or with option
 * "dgrad_group" = 0.
 * packed_ws: device workspace of deepim_conv_dgrad_s2_packed_size bytes. */
size_t deepim_conv_dgrad_s2_packed_size(int Co_l, int Ci_l, int k, int pad);
int deepim_conv2d_dgrad_s2(deepim_ctx* ctx, float* dx, const float* dz, const float* w_layer, float* packed_ws, int B, int Ci_l,
                           int Hd, int Wd, int Co_l, int k, int pad);
/* Data gradient of a Convolution layer (weights (Co_l,Ci_l,k,k), stride 1 or 2, pad) from its RAW weights, optionally already
 * multiplied by the activation gradient of the layer below: dx (B,Ci_l,Hd,Wd) = lrelu'(act_y) * (dgrad(dz) [+ add]) — act_y =
 * that layer's saved output, add = the gradient reaching it over a skip connection, both laid out like dx; NULL act_y = plain
 * data gradient (add must then be NULL). Stride 1 = deepim_conv_pack_dgrad + the forward kernel, stride 2 =
 * deepim_conv2d_dgrad_s2; the activation gradient rides in the final stores of the register-fed kernels and their split-K
 * second passes (no pass of its own over dx), other kernel families get it as one extra pass. What module.backward does per
 * encoder layer (deepim/core/module.py:1131-1137). packed_ws: deepim_conv_dgrad_packed_size bytes. */
size_t deepim_conv_dgrad_packed_size(int Co_l, int Ci_l, int k, int stride, int pad);
int deepim_conv2d_dgrad(deepim_ctx* ctx, float* dx, const float* dz, const float* w_layer, float* packed_ws, int B, int Ci_l,
                        int Hd, int Wd, int Co_l, int k, int stride, int pad, const float* act_y, const float* add, float slope);
/* out (BC,Hd,Wd) = in (BC,Ho,Wo) with stride-1 zeros between the samples (data gradient of a strided convolution) */
int deepim_dilate2d(deepim_ctx* ctx, float* out, const float* in, int BC, int Ho, int Wo, int Hd, int Wd, int stride);
/* the same with an offset: out[bc][off_y + stride*y][off_x + stride*x] = in[bc][y][x], zeros elsewhere — also the backward of
 * Crop (stride 1): the gradient of a cropped Deconvolution output put back into the full frame */
int deepim_scatter2d(deepim_ctx* ctx, float* out, const float* in, int BC, int Ho, int Wo, int Hd, int Wd, int stride,
                     int off_y, int off_x);
/* data gradient of deepim_upsample16_crop_forward (depthwise k32 s16 transposed conv with fixed bilinear weights, lr_mult 0:
 * deepIM_flownet.py:185-200,326-340): d_in (B,C,H,W) from dy (B,C,Ho,Wo) */
int deepim_upsample16_crop_backward(deepim_ctx* ctx, float* d_in, const float* dy, const float* w, int B, int C, int H, int W,
                                    int Ho, int Wo, int crop_y, int crop_x, float scale);
/* backward of [Concat slice -> LeakyReLU -> Crop] in front of a Deconvolution, in one walk: out (B,C,hf,wf) = channels
 * [coff, coff+C) of dcat (B,ctotal,ho,wo) times lrelu'(same slice of ycat; NULL: no activation) at offset (off_y, off_x), zeros
 * around; db[c] = its sum (NULL: skipped). deepIM_flownet.py:120-167 */
int deepim_slice_lrelu_bias_scatter(deepim_ctx* ctx, float* out, float* db, const float* dcat, const float* ycat, int B, int ctotal,
                                    int coff, int C, int ho, int wo, int hf, int wf, int off_y, int off_x, float slope);
/* backward of Concat: dst (B,C,hw) = channels [src_coff, src_coff+C) of src (B,src_ctotal,hw) */
int deepim_extract_channels(deepim_ctx* ctx, float* dst, const float* src, int src_ctotal, int src_coff, int C, int B,
                            size_t hw);
/* dw (Cout,Cin,kh,kw) = sum over n, output pixels of dz (B,Cout,Ho,Wo) x the matching taps of x (B,Cin,H,W): MFMA GEMM
 * with the pixels as the reduction dimension; Ho*Wo must be a multiple of 4 */
int deepim_conv2d_wgrad(deepim_ctx* ctx, float* dw, const float* x, const float* dz, int B, int Cin, int H, int W, int Cout,
                        int kh, int kw, int stride, int pad);
/* the weight gradient in TAP-MAJOR layout dw_tm (Cout, kh*kw, Cin) — what the LDS-staged kernel produces fastest (one range
 * check and one address per eight channels of a tap). Same sums, same order as deepim_conv2d_wgrad; needs Cin % 8 == 0, Cout > 4,
 * option "wgrad_lds" on. deepim_sgd_mom_update_multi reads it in place; deepim_weight_grad_to_natural gives (Cout,Cin,kh,kw). */
int deepim_conv2d_wgrad_tm(deepim_ctx* ctx, float* dw_tm, const float* x, const float* dz, int B, int Cin, int H, int W, int Cout,
                           int kh, int kw, int stride, int pad);
int deepim_weight_grad_to_natural(deepim_ctx* ctx, float* dw, const float* dw_tm, int Cout, int Cin, int khw);
/* weight AND bias gradient of a layer (db[c] = sum of dz): one launch for the few-filter layers (Cout <= 4, 3x3 / 4x4: the
 * prediction heads), deepim_bias_grad + deepim_conv2d_wgrad otherwise */
int deepim_conv2d_wgrad_bias(deepim_ctx* ctx, float* dw, float* db, const float* x, const float* dz, int B, int Cin, int H, int W,
                             int Cout, int kh, int kw, int stride, int pad);
/* FullyConnected backward: dx (B,I) = dy·w, dw (O,I) = dyT·x, db (O) = sum_b dy; any of dx/dw/db may be NULL */
int deepim_fc_backward(deepim_ctx* ctx, float* dx, float* dw, float* db, const float* dy, const float* x, const float* w,
                       int B, int I, int O);
/* MXNet sgd_mom_update (train.py:296-303): mom = momentum*mom - lr*(rescale*g [clipped to +-clip if clip > 0] + wd*w); w += mom */
int deepim_sgd_mom_update(deepim_ctx* ctx, float* w, float* mom, const float* g, float lr, float wd, float momentum,
                          float rescale, float clip, size_t n);

/* the same update of every parameter in one launch. table (device): `rows` rows of six 64-bit words {w, mom, g (device
 * addresses), n, (bits of float wd) | (first block of the row << 32), layout of g: 0 = like w, else Cin | (kh*kw << 32) = the
 * tap-major gradient of deepim_conv2d_wgrad_tm}; a row owns ceil(n/1024) blocks (four parameters per thread: w, mom and a natural g
 * must be 16-byte aligned), rows in block order, total_blocks = their sum.
 * Results are bit-identical to per-tensor deepim_sgd_mom_update calls on natural gradients. */
int deepim_sgd_mom_update_multi(deepim_ctx* ctx, const unsigned long long* table, int rows, int total_blocks, float lr,
                                float momentum, float rescale, float clip);

/* ------------------------------------- R-group: re-render between iterations -- */
/* Replaces Render_Py.render (lib/render_glumpy/render_py_multi.py:101-129: OpenGL draw + glReadPixels +
 * depth linearisation) and the tensor packing that follows it in the batch updater
 * (lib/pair_matching/batch_updater_py_multi.py:117-133) for a whole batch of poses of ONE mesh, on the device.
 * Camera model of :132-147 (a pixel is covered when its index lies inside the triangle projected with K),
 * GL_LESS depth test, no culling, fragments outside (zNear, zFar) dropped; triangles with a vertex at or behind
 * zNear are dropped whole instead of clipped.
 *   image  (B,3,H,W) device: RGB − pixel_means (pixel_means_host in tensor channel order, NULL = 0); background = −means
 *   depth  (B,1,H,W) device: metric z, 0 = background
 *   vertices (V,3) device, model frame; faces (F,3) int32 device; poses (B,3,4) device [R|t]
 *   texture != NULL: (tex_h,tex_w,3) device, values on the 0..255 scale, row 0 ↔ v = 0, sampled GL_LINEAR/clamp;
 *                    vertex_attr = (V,2) uv.   texture == NULL: vertex_attr = (V,3) RGB on the 0..255 scale.
 *   K_host: 9 floats, row-major intrinsics. */
int deepim_render_forward(deepim_ctx* ctx, float* image, float* depth, const float* vertices,
                          const float* vertex_attr, const int32_t* faces, const float* texture,
                          int tex_h, int tex_w, const float* poses, const float* K_host,
                          const float* pixel_means_host, int V, int F, int B, int H, int W, float znear,
                          float zfar);
/* The same draw fused with the rest of the test-loop update (deepim/core/tester.py:437-449,
 * lib/pair_matching/data_pair.py:94-105): also writes mask_rendered = depth > mask_thresh (B,1,H,W) and, when mask_box
 * is not NULL, the box_rendered rectangle of that mask (see deepim_mask_box_forward) — one pass over the frame instead
 * of three. */
int deepim_render_update_forward(deepim_ctx* ctx, float* image, float* depth, float* mask_rendered,
                                 float* mask_box /*or NULL*/, float mask_thresh, const float* vertices,
                                 const float* vertex_attr, const int32_t* faces, const float* texture,
                                 int tex_h, int tex_w, const float* poses, const float* K_host,
                                 const float* pixel_means_host, int V, int F, int B, int H, int W,
                                 float znear, float zfar);
/* The draw of the ModelNet loops (BASELINE config 5): lib/render_glumpy/render_py_light_modelnet_multi.py:36-80,170-188 — the
 * same rasterisation with the per-fragment diffuse term of its shader, in OpenGL camera coordinates (y, z of the pose flipped):
 *   brightness = clamp(n·(L − p) / (|n|·|L − p|), 0, 1),  colour = texture·((1 − r) + r·brightness)·intensity,
 *   read back as round(clamp(colour, 0, 1)·255) (the reference reads uint8, :162-166);  p, n = the fragment's interpolated
 *   position / normal under the pose;  L = light_offset + (t_x, −t_y, −t_z) of the sample's pose, as the render closures of
 *   deepim/core/tester.py:146-172 and lib/pair_matching/batch_updater_py_multi.py:185-228 set it (offset 0.5·(0,1,1)).
 * normals (V,3) device; light_offset_host 3 floats; light_intensity (B,3) device or NULL (= 1,1,1; the reference draws
 * U(0.9,1.1) per sample); brightness_ratio r (0.7 there). mask_rendered / mask_box may be NULL (see deepim_render_update_forward).
 * OpenGL itself cannot run here: restated, PARITY UNPINNED like the unlit draw (tests: oracle/render.py). */
int deepim_render_lit_forward(deepim_ctx* ctx, float* image, float* depth, float* mask_rendered /*or NULL*/,
                              float* mask_box /*or NULL*/, float mask_thresh, const float* vertices,
                              const float* vertex_attr, const float* normals, const int32_t* faces, const float* texture,
                              int tex_h, int tex_w, const float* poses, const float* K_host, const float* pixel_means_host,
                              const float* light_offset_host, const float* light_intensity /*device (B,3) or NULL*/,
                              float brightness_ratio, int V, int F, int B, int H, int W, float znear, float zfar);

#ifdef __cplusplus
}
#endif
#endif  /* DEEPIM_HIP_H_ */
