/*
 * gd4d.h - C ABI of libgd4d.so: the MI355X (gfx950) decoder hot path of Graph-DETR4D.
 *
 * The reference has NO native code of its own: its hot path calls third-party
 * kernels (mmcv MultiScaleDeformableAttnFunction, ATen grid_sampler / nn.MultiheadAttention /
 * addmm) from Python.  Each entry point below therefore names the reference call site
 * (file:line under projects/mmdet3d_plugin/models/utils/) whose tensor-level work it replaces;
 * the Python-side binding a maintainer would add is in INTEGRATION.md and is what
 * graph-detr4d_amd/_lib.py does with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in any signature
 *     (`stream` is a hipStream_t passed as void*, NULL = the null stream);
 *   - every pointer is DEVICE memory unless its comment says "host";
 *   - the caller owns and allocates every buffer (incl. workspaces, sized by *_workspace_bytes);
 *     the library never allocates device memory, never synchronises, and launches only on `stream`;
 *   - tensors are dense, row-major, innermost dimension last, in the order written in the comment;
 *   - return value: GD4D_OK (0) or a negative GD4D_E* code (see gd4d_error_string);
 *   - dtype codes: GD4D_F32 / GD4D_BF16 describe the STORAGE type of feature/value tensors;
 *     query-side tensors, weights, accumulation and outputs are always fp32.
 */
#ifndef GD4D_H_
#define GD4D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD4D_ABI_VERSION 55

enum { GD4D_F32 = 0, GD4D_BF16 = 1 };

/* Layout of the projected value tensor (output of gd4d_value_proj_*, input of gd4d_cross_attn_fwd):
 *   PIXEL_MAJOR (B*N, S, Hh, Dh) - what mmcv's MultiScaleDeformableAttnFunction takes (deform3d_cross_attn.py:280)
 *   HEAD_MAJOR  (B*N, Hh, S, Dh) - one contiguous S x Dh plane per (camera, head): the two x-adjacent
 *                bilinear corners of a head are one contiguous 2*Dh run (better DRAM locality). */
enum { GD4D_LAYOUT_PIXEL_MAJOR = 0, GD4D_LAYOUT_HEAD_MAJOR = 1 };

/* Arithmetic of gd4d_value_proj_*: F32 = split-bf16 x3 MFMA, fp32-class results (<= ~2^-17 relative per
 * product); BF16 = one bf16 product per MAC with fp32 accumulation (bf16-class, only with bf16 output). */
enum { GD4D_VP_PRECISION_F32 = 0, GD4D_VP_PRECISION_BF16 = 1 };

enum {
  GD4D_OK = 0,
  GD4D_EINVAL = -1,       /* NULL pointer / non-positive size */
  GD4D_EUNSUPPORTED = -2, /* shape outside what the kernels are built for (see each function) */
  GD4D_EALIGN = -3,       /* pointer not 16-byte aligned where the kernel does 16-byte accesses */
  GD4D_ELAUNCH = -4,      /* hipLaunch / hipGetLastError failed */
  GD4D_EWORKSPACE = -5    /* workspace too small */
};

#define GD4D_MAX_LEVELS 8
#define GD4D_MAX_LAYERS 8
#define GD4D_CA_RAW_CAM_WEIGHTS 1
#define GD4D_CA_PLAN_ITEMS 2      /* gd4d_cross_attn_plan_fwd: write the ITEMS form of the plan (gd4d_cross_attn_agg_items_fwd) */
#define GD4D_CA_PLAN_BOTH 16      /* pairs AND items: plan_bytes >= 2 x gd4d_cross_attn_plan_bytes(); [0, bytes) is the pairs plan (what the
                                     training backward kernels read), plan + bytes an items plan for gd4d_cross_attn_agg_items_fwd */

int gd4d_abi_version(void);
const char* gd4d_error_string(int code);
/* Text of the last HIP error seen by this thread inside the library ("" if none). */
const char* gd4d_last_hip_error(void);
/* dev: device-side timeline.  buffer = device uint64 array {entries written, capacity, then (id, 100-MHz time) pairs}
 * zeroed by the caller with [1] = capacity; block 0 of the row-chain, attention-core, channels-last and aggregate kernels
 * stamps its entry (and exit) time while a buffer is set.  NULL disables.  Synchronises the device (hipMemcpyToSymbol). */
int gd4d_trace_enable(void* buffer);

/* --------------------------------------------------------------------------------------------
 * gd4d_cross_attn_fwd - fused 3D->2D projection + visibility mask + masked softmax weights +
 * multi-camera / multi-level bilinear gather + per-camera sigmoid-weighted reduction.
 *
 * Replaces, in Deform3DCrossAttn.forward (deform3d_cross_attn.py):
 *   :220-258  de-normalise reference points, add metre offsets, lidar2img matmul, depth/border mask
 *   :274,281-284  softmax over L*P, multiplied by the mask without renormalisation
 *   :301-304  MultiScaleDeformableAttnFunction.apply (third-party mmcv CUDA kernel)
 *   :320-324  sigmoid camera weights (raw-view "scramble" of :211-212 applied here) and sum over cameras
 *
 *   value        (B*N, S, Hh, Dh) or (B*N, Hh, S, Dh) per `value_layout`; S = sum_l H_l*W_l; output of value_proj
 *   level_hw     host, L x 2 int32 (H_l, W_l)       (reference: spatial_shapes, :271)
 *   ref          (B, Q, 3) fp32 in [0,1]            (reference_points)
 *   offsets      (B, Q, Hh, P, 3) fp32 metres       (deform_sampling_offsets(query), :227)
 *   attn_logits  (B, Q, Hh, L, P) fp32              (attention_weights(query), :281)
 *   cam_logits   (B, Q, N) fp32, the UN-scrambled Linear output (:211); weight of camera n for
 *                query q is sigmoid(flat[b][n*Q + q])
 *   lidar2img    (B, N, 4, 4) fp32, rows act on [x y z 1]^T
 *   pc_range     host, 6 doubles [x0 y0 z0 x1 y1 z1] ((hi-lo) is formed in double like the
 *                reference's Python floats, :222-224)
 *   img_h, img_w img_metas[0]['img_shape'][0][:2]   (:242-243)
 *   out          (B, Q, Hh*Dh) fp32 = sum_n sigmoid(cam) * MSDA_n   (input of output_proj, :326)
 *   mask_out     optional (B, N, Q, Hh, P) uint8 visibility mask (bit-exact vs the reference's CPU
 *                arithmetic at this boundary); NULL to skip
 *   uv_out       optional (B, N, Q, Hh, P, 2) fp32 normalised image coordinates; NULL to skip
 *   flags        GD4D_CA_RAW_CAM_WEIGHTS: weight of camera n is the raw flat[b][n*Q + q] instead of its sigmoid
 *                (the neighbour pass of Deform3DCrossAttnMP, deform3d_cross_attn_multi_point.py:424-430)
 *   query_order  optional int32 permutation of [0, B*Q) from gd4d_query_order_fwd, or NULL.  Scheduling only: with
 *                it each XCD (private L2) processes queries that look at the same camera region; the result is
 *                bit-identical for any permutation.
 *
 * Attention logits for value row i = b*N+n are taken from batch (i % B): this is what the
 * reference's `query.repeat(num_cams,1,1)` (:277) pairs them with; identity for B = 1.
 *
 * Supported: Hh*Dh == 256 and Dh % 4 == 0 (fp32) / Dh % 8 == 0 (bf16), P == 4 (or P == 1: one point per level, the
 * neighbour pass of Deform3DCrossAttnMP), 1 <= L <= 8,
 * N <= 64.  `value` 16-byte aligned.
 */
int gd4d_cross_attn_fwd(const void* value, const int32_t* level_hw, const float* ref,
                        const float* offsets, const float* attn_logits, const float* cam_logits,
                        const float* lidar2img, const double* pc_range, float img_h, float img_w,
                        float* out, uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh,
                        int Dh, int L, int P, int value_dtype, int value_layout, int flags,
                        const int32_t* query_order, void* stream);

/* --------------------------------------------------------------------------------------------
 * Aggregate-then-project form of the same path (inference; gd4d_cross_attn_late.hip).  value_proj is linear and the
 * gather is a weighted sum of value rows, so per head h
 *     out[q, h*Dh + d] = ( W_h * sum_i w_i x_i )[d] + b_h[d] * sum_i w_i
 * over the in-bounds bilinear corners i of the head's visible samples, x_i = the RAW C-channel feature of the pixel:
 * the per-layer projected value tensors (deform3d_cross_attn.py:264-280: N * S * 256 fp32 per decoder layer) are never
 * written.  Three entry points:
 *
 * gd4d_pyramid_channels_last_fwd - the reference's flatten(3) / transpose / cat of the FPN levels (:264-276), once
 *   per sample instead of once per layer:  feats host array of L device pointers, level l = (R, C, H_l, W_l) fp32
 *   (R = B*N camera rows);  out (R, S, C) fp32 or bf16 (out_dtype; bf16 = round to nearest even: the reduced-precision
 *   storage mode of `value_dtype='bf16'`, fp32 accumulation downstream), S = sum_l H_l*W_l, level l at pixel offset
 *   sum_{l'<l} H_l'*W_l'.
 *   max_cus: 0 = one workgroup per tile over the whole device; > 0 = one persistent workgroup on each of max_cus compute
 *   units, which it fills (the others stay free for kernels of another stream: the first layer's query side).
 *   Supported: C == 256, L <= 8, fp32.
 *
 * gd4d_cross_attn_agg_fwd - projection + mask + softmax / camera weights + gather of raw features.  Arguments as
 *   gd4d_cross_attn_fwd except: feats_cl = the channels-last pyramid above instead of a projected value tensor;
 *   agg (B*Q, Hh, C) fp32 out: sum_i w_i x_i per head;  wsum (B*Q, Hh) fp32 out: sum_i w_i (in-bounds corners only -
 *   mmcv's zero padding drops the bias with the value).  Same visibility mask / uv (bit-exact) as gd4d_cross_attn_fwd.
 *   vp_weight (C, C), vp_bias (C) or NULL, out (B*Q, C): optional - value_proj applied to the aggregates in the kernel's
 *   epilogue (exact fp32 FMAs), out = what gd4d_value_proj_heads_fwd would return; agg / wsum may then be NULL.
 *   Supported: B == 1 (for B > 1 the reference pairs value rows with the logits of batch (row % B), :277 - use
 *   gd4d_cross_attn_fwd), C == 256, P == 4, L <= 4, N <= 64, Hh in {4, 8, 16}, B*N*S < 2^31, features fp32 or
 *   bf16 (feats_dtype).
 *
 * gd4d_value_proj_heads_fwd - value_proj applied to the aggregates: out (M, Hh*Dh) with
 *   out[r, h*Dh + d] = sum_c weight[h*Dh + d][c] * agg[r][h][c] + bias[h*Dh + d] * wsum[r][h]
 *   (exact fp32 products on v_mfma_f32_16x16x4_f32); out is what gd4d_cross_attn_fwd returns (input of output_proj,
 *   :326).  weight (C, C) = value_proj.weight, bias (C) or NULL.  Supported: C == 256, Hh in {4, 8, 16}. */
int gd4d_pyramid_channels_last_fwd(const void* const* feats, const int32_t* level_hw, void* out, int R, int C, int L,
                                   int in_dtype, int out_dtype, int max_cus, void* stream);
int gd4d_cross_attn_agg_fwd(const void* feats_cl, const int32_t* level_hw, const float* ref, const float* offsets,
                            const float* attn_logits, const float* cam_logits, const float* lidar2img,
                            const double* pc_range, float img_h, float img_w, float* agg, float* wsum, uint8_t* mask_out,
                            float* uv_out, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype, int flags,
                            const int32_t* query_order, const float* vp_weight, const float* vp_bias, float* out,
                            void* stream);
int gd4d_value_proj_heads_fwd(const float* agg, const float* wsum, const float* weight, const float* bias, float* out,
                              int M, int Hh, int C, void* stream);

/* --------------------------------------------------------------------------------------------
 * Channel-sliced form of the aggregate-then-project gather (gd4d_cross_attn_sliced.hip; the default of the inference
 * step).  Same arithmetic and same reference lines as gd4d_cross_attn_agg_fwd (deform3d_cross_attn.py:220-258, :277,
 * :281-284, :301-304, :320-324), another shape of the work: the 256 channels are cut into 8 slices of 32 (one 128-byte
 * line per bilinear corner), a workgroup is one (query, slice), and the workgroups of an XCD run slice-major - at any
 * time an XCD gathers ONE slice of all its queries, 1/8 of the footprint its 4-MB L2 has to hold.
 *
 * gd4d_pyramid_slice_planar_fwd - as gd4d_pyramid_channels_last_fwd (the reference's flatten / transpose / cat,
 *   :264-276) but out = (8, R, S, 32): slice s of every pixel in one contiguous plane.
 *
 * gd4d_cross_attn_plan_fwd - per (sample, query), once per decoder layer: everything that does not depend on the slice.
 *   Projection + visibility mask (bit-exact, the routine of gd4d_cross_attn_fwd) + softmax over L*P + camera weights +
 *   the bilinear corner of every visible sample at every level: writes `plan` (gd4d_cross_attn_plan_bytes(B, N, Q, Hh, P)
 *   bytes, 16-byte aligned) - per head the number of visible (camera, point) items and, per pass of 4 items, 64 pairs
 *   {byte offset inside the level, weight} in the order the gather's lanes consume them - and wsum (B*Q, Hh) fp32 =
 *   sum_i w_i over the in-bounds corners (mmcv's zero padding drops the bias with the value).  The byte offsets are
 *       r * cam_stride_bytes[l] + (y*W_l + x) * pix_stride_bytes       (camera row r = b*N + n, level l = level_hw[l])
 *   of the pyramid the gather will read.  Value row i = b*N + n takes the logits of batch (i % B) (:277).  mask_out /
 *   uv_out as gd4d_cross_attn_fwd.  query_order (or NULL): the plan is stored by POSITION in that order - hand the same
 *   order to gd4d_cross_attn_agg_sliced_fwd.  Supported: P == 4, L <= 4, N <= 64, B <= 16, Hh in {4, 8, 16}, every
 *   level's bytes < 2^32.
 *
 * gd4d_cross_attn_agg_sliced_fwd - the gather: agg (B*Q, Hh, C) fp32 = sum_i w_i x_i, from a pyramid addressed as
 *     address(level l, camera row r, pixel y*W_l + x, slice s, channel k of the slice)
 *         = level_ptrs[l] + r * cam_stride_bytes[l] + (y*W_l + x) * pix_stride_bytes + s * slice_stride_bytes + k * elem
 *   level_ptrs: host array of L device pointers.  This covers the slice-planar copy (pix 128, slice R*S*128), the
 *   pixel-major copy of gd4d_pyramid_channels_last_fwd (pix 1024, slice 128) and caller-owned channels-last (NHWC) levels
 *   read in place without any copy (per-level pointers, pix 1024, slice 128; bf16: half of each).  slice_lo / slice_n:
 *   the slices of this launch (0, 8 = all).  query_order: the one the plan was made with (scheduling only: results are
 *   bit-identical for any permutation).  Supported: C == 256, P == 4, L <= 4, N <= 64, B <= 16, Hh in {4, 8, 16}.
 *
 * The ITEMS form of the plan (flags & GD4D_CA_PLAN_ITEMS; what the inference step uses).  The eight slices of a query
 *   read their plan a phase apart - too far for the L2 - so the plan's bytes are paid eight times over the fabric.  In
 *   this form gd4d_cross_attn_plan_fwd stores 32 bytes per visible (camera, point) item - {u, v, camera row, count}
 *   {softmax x camera weight of level 0..3} - instead of 4 levels x 4 corners x {offset, weight} = 128 bytes, does not
 *   touch `wsum`, and gd4d_cross_attn_agg_items_fwd does the corner arithmetic itself (lane = (item, level), 16 items per
 *   step, the same operations in the same order: agg and wsum are bit-identical to the pairs form) from the geometry
 *   arguments, which must be the ones the plan was made with.  wsum (B*Q, Hh) is written by the launch that contains
 *   slice 0 (NULL: not wanted).  A level may span up to 64 GiB (offsets in units of 16 bytes once a level reaches 4 GiB -
 *   e.g. VoVNet-99 level 0 stored channels-last with B >= 2; all strides must then be multiples of 16); the pairs form is
 *   limited to 4 GiB per level.  The training backward kernels read the pairs form only.
 * */
size_t gd4d_cross_attn_plan_bytes(int B, int N, int Q, int Hh, int P);
int gd4d_cross_attn_agg_items_fwd(const void* const* level_ptrs, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                  int64_t pix_stride_bytes, int64_t slice_stride_bytes, const void* plan, float* agg,
                                  float* wsum, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                  const int32_t* query_order, int slice_lo, int slice_n, void* stream);

/* gd4d_cross_attn_agg_items_coarse_fwd - the same gather (deform3d_cross_attn.py:264-280, :301-304, :320-324) with the two COARSE
 * levels taken from rows value_proj has already been applied to.  value_proj is linear and the gather a weighted sum:
 *     out[q, h] = W_h (sum over fine-level corners w_i x_i) + b_h (sum of those w_i)  +  sum over coarse-level corners w_i (W_h x_i + b_h)
 * A raw corner costs 1 KB through the L1s whatever its level (8 slices x 128 B), a projected corner of head h its own 128 B; levels 2-3
 * are 6 % of the pixels (43 800 rows at 24 cameras: gd4d_value_proj_fwd over them is ~ 6 % of the whole pyramid's) and were a third of
 * the launch.  L == 4 and Hh == 8 (a head = 32 channels = one 128-byte line) only.
 *   level_ptrs .. slice_stride_bytes: the raw pyramid as gd4d_cross_attn_agg_items_fwd takes it (levels 0, 1 are read; entries 2, 3 of
 *   level_hw are the coarse levels' shapes); proj_ptrs[0 .. 1]: levels 2, 3 projected - (R, H_l W_l, 256) fp32 rows, pixel-major, the
 *   bias included (gd4d_value_proj_fwd on those two levels, GD4D_LAYOUT_PIXEL_MAJOR), camera row r of level l at
 *   proj_ptrs[l - 2] + r * proj_cam_stride_bytes[l - 2]; plan: the ITEMS form.
 *   agg (B*Q, 8, 256), wsum (B*Q, 8): the FINE levels' aggregates and weight sums; pagg (B*Q, 256): the coarse levels' part, already
 *   projected, columns head-major - GD4D_CHAIN_HEADGEMM / gd4d_value_proj_heads_fwd of (agg, wsum) plus pagg is the layer's
 *   sampled value (what output_proj takes).  One launch: a ninth workgroup per query (the walk's last phase) gathers the projected rows. */
int gd4d_cross_attn_agg_items_coarse_fwd(const void* const* level_ptrs, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                         int64_t pix_stride_bytes, int64_t slice_stride_bytes, const void* const* proj_ptrs,
                                         const int64_t* proj_cam_stride_bytes, const void* plan, float* agg, float* wsum,
                                         float* pagg, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                         const int32_t* query_order, void* stream);

/* gd4d_cross_attn_agg_items_count_fwd - a TRAINING step's forward gather and the first step of the pyramid gradient's bookkeeping
 * in one launch: gd4d_cross_attn_agg_items_fwd (all slices, every level) on plan_items and gd4d_pyramid_grad_count on plan_pairs (the
 * two forms of one plan: GD4D_CA_PLAN_BOTH) - the same results as the two launches (the slots, as there, in an order the atomics
 * decide).  Both only read the plan; the count lives on L2 atomic round trips, the gather on the fabric: every ninth group of
 * eight workgroups counts, the others gather (each keeps its XCD).  8 heads, 4 levels, fp32 features, narrow offsets
 * (GD4D_EUNSUPPORTED otherwise: launch the two). */
int gd4d_cross_attn_agg_items_count_fwd(const void* const* level_ptrs, const int32_t* level_hw, const int64_t* cam_stride_bytes,
                                        int64_t pix_stride_bytes, int64_t slice_stride_bytes, const void* plan_items, float* agg,
                                        float* wsum, int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                        const int32_t* query_order, const void* plan_pairs, int32_t* count, void* slots,
                                        size_t slots_bytes, void* stream);
int gd4d_cross_attn_plan_fwd(const float* ref, const float* offsets, const float* attn_logits, const float* cam_logits,
                             const float* lidar2img, const double* pc_range, float img_h, float img_w,
                             const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes, void* plan,
                             size_t plan_bytes, float* wsum, uint8_t* mask_out, float* uv_out, int B, int N, int Q, int Hh,
                             int L, int P, int flags, const int32_t* query_order, void* stream);
int gd4d_cross_attn_agg_sliced_fwd(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan, float* agg,
                                   int B, int N, int Q, int Hh, int C, int L, int P, int feats_dtype,
                                   const int32_t* query_order, int slice_lo, int slice_n, void* stream);
int gd4d_pyramid_slice_planar_fwd(const void* const* feats, const int32_t* level_hw, void* out, int R, int C, int L,
                                  int in_dtype, int out_dtype, int max_cus, void* stream);

/* --------------------------------------------------------------------------------------------
 * Training backward of the channel-sliced path (gd4d_cross_attn_sliced_bwd.hip): the pyramid side of a training step
 * without a projected value tensor.  Replaces, for Deform3DCrossAttn, autograd over deform3d_cross_attn.py:264-324
 * (value_proj over every pixel, mmcv's ms_deformable_col2im with its atomicAdd scatter, the elementwise chain).
 * Forward = gd4d_cross_attn_plan_fwd + gd4d_cross_attn_agg_sliced_fwd + gd4d_value_proj_heads_fwd:
 *     out[q, h] = W_h A[q, h] + b_h s[q, h],  A = sum_r w_r x_r,  s = sum_r w_r   (r: in-bounds corners, x_r: raw pixel).
 *
 * gd4d_value_proj_heads_bwd - grad_agg (M, Hh, C) = W_h^T grad_out[m, h], beta (M, Hh) = <b_h, grad_out[m, h]> (bias NULL:
 *   zeros; beta NULL: not written).
 * gd4d_value_proj_heads_bwd_weight - value_proj's own gradients from the forward's aggregates: grad_weight (C, C) with
 *   dW_h = sum_m grad_out[m, h] (x) agg[m, h], grad_bias (C) (or NULL) with db_h = sum_m grad_out[m, h] wsum[m, h]: a
 *   contraction over the M = B*Q rows instead of gd4d_value_proj_bwd_weight's over every pixel; fixed summation order
 *   (workspace: gd4d_value_proj_heads_bwd_weight_workspace_bytes, 16-byte aligned; accumulate != 0: added to grad_weight /
 *   grad_bias).
 * gd4d_cross_attn_dot_sliced - D[pair] = <grad_agg[q, h], x_pair> for every pair of the plan, as 8 per-slice partials:
 *   dpart = (8, B*Q*Hh*cap_t*64) fp32, gd4d_cross_attn_dot_bytes(B, N, Q, Hh, P) bytes; same pyramid addressing, plan and
 *   query_order as gd4d_cross_attn_agg_sliced_fwd (fp32 or bf16 pyramids; products and sums in fp32).
 * gd4d_cross_attn_plan_bwd - the query-side gradients: recomputes the plan kernel's geometry (same routine: item m here
 *   is item m of the plan), dL/d w_r = sum_slices D + beta for in-bounds corners, then the chain rule of the bilinear
 *   weights, the softmax over L*P, the camera sigmoid (GD4D_CA_RAW_CAM_WEIGHTS: none), u = cx / (cz W), v = cy / (cz H)
 *   and lidar2img: grad_ref (B, Q, 3), grad_offsets (B, Q, Hh, P, 3), grad_attn_logits (B, Q, Hh, L, P), grad_cam_logits
 *   (B, Q, N) in the raw-view layout of cam_logits - the outputs of gd4d_cross_attn_bwd, which it replaces.  The mask is
 *   piecewise constant (no gradient), as in the reference.  workspace: gd4d_cross_attn_bwd_workspace_bytes (B > 1).
 *   status (or NULL): set to 1 if an item count disagrees with the plan's header.  All sums in a fixed order.
 * gd4d_pyramid_grad_count / _scan / _fill / _sort / _reduce - the gradient of the NCHW pyramid from the plans and grad_agg tables of
 *   ALL decoder layers at once, without atomics on feature data: the (pixel, weight, table row) records are bucketed by
 *   CHUNK (<= 64 pixels: cw x ch of one camera row and level, 32 x 2 on the fine levels, smaller on the coarse ones;
 *   gd4d_pyramid_grad_chunks(level_hw, R, L) chunks in all) and grouped by pixel inside a chunk, then one pass sums each
 *   pixel's records and writes every pixel of the gradient exactly once.  Only that last pass needs the gradients: count
 *   can run in the forward pass, scan / fill / sort beside the backward pass.
 *     count   one call per layer: every pair of the layer's plan with a non-zero weight takes a slot in its chunk's
 *             bucket (count (chunks) int32, zeroed by the caller before the first layer; lanes of a wave that share a
 *             chunk share one atomic); slots (gd4d_pyramid_grad_slots_bytes, the plan's layout) keeps {chunk, pixel in
 *             chunk, slot} for fill
 *     scan    start = exclusive prefix sum of count (n = chunks; workspace: gd4d_pyramid_grad_scan_workspace_bytes)
 *     fill    one call per layer: records[start[chunk] + slot] = {weight, pixel-in-chunk << 26 | id_base + bq * Hh + h}
 *             (8 bytes per counted pair; id_base + B*Q*Hh <= 2^26)
 *     sort    every chunk's records grouped by pixel: sorted (same size and chunk offsets as records), pxoff (chunks, 65)
 *             int32 = where every pixel's run starts inside its chunk ([64] = the chunk's record count)
 *     reduce  grads[l] (R, C, H_l, W_l) fp32 - channels_last != 0: (R, H_l, W_l, C), the layout of levels the gather read in
 *             place - = per pixel, sum over its run of weight * table[id, :]; table
 *             (rows, C) holds the grad_agg of every layer (row id as handed to fill).  The order of the additions follows
 *             the slot hand-out (like the atomicAdd scatter it replaces, sums may differ in the last bits between runs).
 *             chunk_order (or NULL): a permutation of the chunks - the order the workgroups walk them in, XCD x taking
 *             the x-th eighth; results do not depend on it.  gd4d_pyramid_grad_chunk_geometry writes per level
 *             {log2 cw, log2 ch, chunks across, chunks down, first chunk} (5 int32) for callers that build one: chunk id
 *             = first + (r * down + y / ch) * across + x / cw. */
int gd4d_value_proj_heads_bwd(const float* grad_out, const float* weight, const float* bias, float* grad_agg, float* beta,
                              int M, int Hh, int C, void* stream);
size_t gd4d_value_proj_heads_bwd_weight_workspace_bytes(void);
int gd4d_value_proj_heads_bwd_weight(const float* grad_out, const float* agg, const float* wsum, float* grad_weight,
                                     float* grad_bias, void* workspace, size_t workspace_bytes, int M, int Hh, int C,
                                     int accumulate, void* stream);
/* ... for up to 8 layers of a training step in ONE pair of launches (host arrays of `count` device pointers; rows[i] = M of
 * problem i; grad_bias[i] may be NULL; workspace: count x gd4d_value_proj_heads_bwd_weight_workspace_bytes()): nothing reads
 * these gradients before the optimizer, so a step queues them like the Linears' (gd4d_linear_bwd_weight_group). */
int gd4d_value_proj_heads_bwd_weight_group(const void* const* grad_out, const void* const* agg, const void* const* wsum,
                                           void* const* grad_weight, void* const* grad_bias, const int32_t* rows, int count,
                                           void* workspace, size_t workspace_bytes, int Hh, int C, int accumulate, void* stream);
size_t gd4d_cross_attn_dot_bytes(int B, int N, int Q, int Hh, int P);
int gd4d_cross_attn_dot_sliced(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan,
                               const float* grad_agg, void* dpart, size_t dpart_bytes, int B, int N, int Q, int Hh, int C, int L,
                               int P, int feats_dtype, const int32_t* query_order, void* stream);

/* gd4d_cross_attn_dot_sliced_wgrad - gd4d_cross_attn_dot_sliced and gd4d_linear_bwd_weight_group (its arguments: up to 16 weight /
 * bias gradients of the decoder's Linears, dims = {M, K, N, ldx, ldy} per problem) in one launch: the weight-gradient tiles are
 * guest workgroups of the gather-dot (8 waves each, so their row sums are taken in another - fixed - order than the stand-alone
 * kernel's 16 waves: equal within fp32 rounding).  Nothing reads a weight gradient before the optimizer: a training step hands
 * what it has queued to the next backward gather instead of five launches when the backward pass ends.  8 heads, 4 levels, fp32
 * features (GD4D_EUNSUPPORTED otherwise). */
int gd4d_cross_attn_dot_sliced_wgrad(const void* const* level_ptrs, int64_t slice_stride_bytes, const void* plan,
                                     const float* grad_agg, void* dpart, size_t dpart_bytes, int B, int N, int Q, int Hh, int C,
                                     int L, int P, int feats_dtype, const int32_t* query_order, const void* const* x,
                                     const void* const* grad_y, void* const* grad_w, void* const* grad_b, const int32_t* dims,
                                     int count, int accumulate, void* stream);

int gd4d_cross_attn_plan_bwd(const float* ref, const float* offsets, const float* attn_logits, const float* cam_logits,
                             const float* lidar2img, const double* pc_range, float img_h, float img_w, const int32_t* level_hw,
                             const void* plan, const void* dpart, const float* beta, float* grad_ref, float* grad_offsets,
                             float* grad_attn_logits, float* grad_cam_logits, void* workspace, size_t workspace_bytes,
                             int32_t* status, int B, int N, int Q, int Hh, int L, int P, int flags, const int32_t* query_order,
                             void* stream);
int64_t gd4d_pyramid_grad_chunks(const int32_t* level_hw, int R, int L);
size_t gd4d_pyramid_grad_slots_bytes(int B, int N, int Q, int Hh, int P);
int gd4d_pyramid_grad_count(const void* plan, const int32_t* level_hw, const int64_t* cam_stride_bytes, int64_t pix_stride_bytes,
                            int32_t* count, void* slots, size_t slots_bytes, int B, int N, int Q, int Hh, int L, int P,
                            void* stream);
size_t gd4d_pyramid_grad_scan_workspace_bytes(int64_t n);
int gd4d_pyramid_grad_scan(const int32_t* count, int32_t* cursor, void* workspace, size_t workspace_bytes, int64_t n,
                           void* stream);
int gd4d_pyramid_grad_fill(const void* plan, const void* slots, const int32_t* start, void* records, uint32_t id_base,
                           const int32_t* query_order, int B, int N, int Q, int Hh, int P, void* stream);
int gd4d_pyramid_grad_chunk_geometry(const int32_t* level_hw, int R, int L, int32_t* out);
int gd4d_pyramid_grad_sort(const int32_t* count, const int32_t* start, const void* records, void* sorted, int32_t* pxoff,
                           int64_t chunks, void* stream);
int gd4d_pyramid_grad_reduce(const int32_t* start, const int32_t* pxoff, const void* sorted, const float* table,
                             void* const* grads, const int32_t* level_hw, const int32_t* chunk_order, int R, int C, int L,
                             int channels_last, void* stream);

/* gd4d_query_order_fwd - locality order of the queries for gd4d_cross_attn_fwd (no reference counterpart: the
 * reference's MSDA kernel processes queries in index order).  Counting sort by (sample, azimuth of the de-normalised
 * reference point about the lidar origin, deform3d_cross_attn.py:222-224): the cameras sit near that origin, so
 * neighbours in this order project into the same image columns of the same cameras.
 *   ref (B, Q, 3) fp32 in [0,1], pc_range host 6 doubles;
 *   order (B*Q) int32 out: a permutation of [0, B*Q) (order inside an azimuth bin unspecified).
 * Supported: B <= 512, B*Q <= 4096 (callers fall back to no order beyond that).
 */
int gd4d_query_order_fwd(const float* ref, const double* pc_range, int32_t* order, int B, int Q, void* stream);

/* gd4d_refine_reference_order_fwd - gd4d_refine_reference_fwd (detr3d_transformer.py:201-214) and
 * gd4d_query_order_fwd of the refined points in one launch, so that every decoder layer gets a fresh order for free.
 *   tmp (B*Q, ldt), ref (B, Q, 3), out (B, Q, 3), order (B*Q) int32.  Same limits as gd4d_query_order_fwd. */
int gd4d_refine_reference_order_fwd(const float* tmp, const float* ref, float* out, const double* pc_range,
                                    int32_t* order, int B, int Q, int ldt, void* stream);

/* --------------------------------------------------------------------------------------------
 * gd4d_detr3d_fwd - DETR3D-baseline core: feature_sampling (detr3d_transformer.py:397-438) fused
 * with sigmoid(logits) * mask and the sum over levels / points / cameras of
 * Detr3DCrossAtten.forward (detr3d_transformer.py:373-383).  Replaces the per-level ATen
 * F.grid_sample calls (:429-435) and the (B, C, Q, N, 1, L) intermediate.
 *
 *   feats        host array of L device pointers, level l = (B*N, C, H_l, W_l) fp32 NCHW
 *                (exactly the `mlvl_feats` the reference's caller passes, viewed B*N)
 *   level_hw     host, L x 2 int32 (H_l, W_l)
 *   ref          (B, Q, 3) fp32 in [0,1]
 *   attn_logits  (B, Q, N, P=1, L) fp32            (attention_weights(query), :373-374)
 *   lidar2img    (B, N, 4, 4) fp32;  pc_range host 6 doubles;  img_h/img_w as above
 *   out          optional (B, Q, C) fp32 = sum_{n,l} sigmoid(logit) * vis * bilinear  (input of
 *                output_proj, :386)
 *   mask_out     optional (B, N, Q) uint8 visibility (-1 < u',v' < 1 and z > eps), bit-exact
 *   sampled_out  optional (B, C, Q, N, 1, L) fp32: feature_sampling()'s `sampled_feats`, computed
 *                for EVERY camera like the reference (no visibility skipping)
 * At least one of out / mask_out / sampled_out must be non-NULL.  Supported: P == 1, L <= 8.
 */
int gd4d_detr3d_fwd(const void* const* feats, const int32_t* level_hw, const float* ref,
                    const float* attn_logits, const float* lidar2img, const double* pc_range,
                    float img_h, float img_w, float* out, uint8_t* mask_out, float* sampled_out,
                    int B, int N, int Q, int C, int L, int P, void* stream);

/* gd4d_detr3d_bwd - backward of gd4d_detr3d_fwd's `out` (the reference: autograd through feature_sampling's F.grid_sample
 * per level, the sigmoid weights, the mask product and the sums, detr3d_transformer.py:373-383, :397-438): grad_feats
 * (host array of L device pointers to tensors shaped like feats, ZERO-INITIALISED by the caller - the samples' corners are
 * added with fp32 atomics - or NULL: no feature gradient), grad_logits (B, Q, N, 1, L), grad_ref (B, Q, 3) or NULL.  The
 * visibility mask carries no gradient, as in the reference.  Same limits as the forward (P == 1). */
int gd4d_detr3d_bwd(const void* const* feats, const int32_t* level_hw, const float* ref, const float* attn_logits,
                    const float* lidar2img, const double* pc_range, float img_h, float img_w, const float* grad_out,
                    void* const* grad_feats, float* grad_logits, float* grad_ref, int B, int N, int Q, int C, int L, int P,
                    void* stream);

/* gd4d_detr3d_v2_fwd - sampling core of Detr3DCrossAttenV2 (detr3d_transformer.py:441-710, the 2-D-offset deformable
 * variant; registered by the reference, used by no shipped config): projection + [-1,1] visibility test of the reference
 * point (:662-682), per (camera, head) softmax over level x point (:602-611), bilinear samples (grid_sample,
 * align_corners=False, zero padding) of each head's channel slice of the NCHW maps at ref + offset / (W_l, H_l)
 * (:692-705), sum over cameras, levels, points (:617-620).  The reference pairs the sample at (point i, level j) with the
 * weight of (level i, point j) (:611 against :705-707); reproduced, hence P == L is required.
 *   feats / level_hw / ref / lidar2img / pc_range / img_h / img_w as gd4d_detr3d_fwd;
 *   attn_logits (B, Q, N, Hh, L*P), offsets (B, Q, N, Hh, L, P, 2) in pixels of the level;
 *   out (B, Q, C) (channel = head * C/Hh + d: input of output_proj), mask_out optional (B, N, Q) uint8. */
int gd4d_detr3d_v2_fwd(const void* const* feats, const int32_t* level_hw, const float* ref,
                       const float* attn_logits, const float* offsets, const float* lidar2img,
                       const double* pc_range, float img_h, float img_w, float* out, uint8_t* mask_out, int B,
                       int N, int Q, int C, int Hh, int L, int P, void* stream);
/* gd4d_detr3d_v2_bwd - backward of gd4d_detr3d_v2_fwd (the reference: autograd through detr3d_transformer.py:597-710):
 *   grad_feats[l] (B*N, C, H_l, W_l) += (atomic; zero them first; NULL array: not wanted), grad_logits (B, Q, N, Hh, L*P),
 *   grad_offsets (B, Q, N, Hh, L, P, 2), grad_ref (B, Q, 3) or NULL, from grad_out (B, Q, C).  P == L as the forward; C <= 256. */
int gd4d_detr3d_v2_bwd(const void* const* feats, const int32_t* level_hw, const float* ref, const float* attn_logits,
                       const float* offsets, const float* lidar2img, const double* pc_range, float img_h, float img_w,
                       const float* grad_out, void* const* grad_feats, float* grad_logits, float* grad_offsets, float* grad_ref,
                       int B, int N, int Q, int C, int L, int Hh, void* stream);

/* --------------------------------------------------------------------------------------------
 * gd4d_value_proj_fwd - value_proj (Linear C -> C) over the flattened multi-camera pyramid.
 *
 * Replaces, in Deform3DCrossAttn.forward (deform3d_cross_attn.py):
 *   :264-269  per-level .flatten(2).transpose(1,2) (a transposed copy) and torch.cat over levels
 *   :278-280  self.value_proj(value_flatten) and the view to (B*N, S, Hh, Dh)
 * by ONE pass that reads the NCHW maps as the caller holds them and writes the channels-last,
 * head-major value tensor gd4d_cross_attn_fwd consumes.
 *
 *   feats     host array of L device pointers; level l = (R, C, H_l, W_l) fp32 NCHW, R = B*N
 *   level_hw  host, L x 2 int32
 *   weight    (C, C) fp32 row-major [out][in]   (value_proj.weight);  bias (C) fp32 or NULL
 *   out       `out_dtype` (GD4D_F32 | GD4D_BF16); (R, S, C) = (R, S, Hh, Dh) for GD4D_LAYOUT_PIXEL_MAJOR,
 *             (R, Hh, S, C/Hh) for GD4D_LAYOUT_HEAD_MAJOR;  S = sum_l H_l*W_l
 * Arithmetic: split-bf16 x3 MFMA with fp32 accumulation (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi):
 * fp32-class accuracy (<= ~2^-17 relative per product), see DESIGN.md.
 * Supported: C == 256, in_dtype == GD4D_F32, L <= 8.
 *   workspace  device, gd4d_value_proj_workspace_bytes(NL) bytes, 16-byte aligned: the launch first rewrites the weights
 *              as bf16 hi / lo MFMA fragments there (one small kernel, every call: the weights may have changed)
 *   max_cus    0 = the whole device; otherwise at most this many CUs (rounded down to a multiple of 8) are occupied by
 *              the persistent workgroups, so that kernels of another HIP stream find free CUs (a call argument, not
 *              process-wide state)
 */
int gd4d_value_proj_fwd(const void* const* feats, const int32_t* level_hw, const float* weight,
                        const float* bias, void* out, int R, int C, int L, int Hh, int in_dtype,
                        int out_dtype, int out_layout, int precision, void* workspace, size_t workspace_bytes,
                        int max_cus, void* stream);
size_t gd4d_value_proj_workspace_bytes(int NL);

/* gd4d_value_proj_multi_fwd - the same projection for NL decoder layers in ONE launch.
 * Every decoder layer receives the same `value` list (Detr3DTransformerDecoder.forward passes
 * `*args` unchanged to each layer, detr3d_transformer.py:192-198), so the NL value_proj GEMMs share
 * their input: a wave keeps its pixel tile's A fragments in registers for all NL layers and the pyramid is read
 * from HBM once instead of NL times.  weights / biases / outs: host arrays of NL device pointers (biases or its
 * entries may be NULL).  NL <= GD4D_MAX_LAYERS.  Results are bit-identical to NL gd4d_value_proj_fwd calls. */
int gd4d_value_proj_multi_fwd(const void* const* feats, const int32_t* level_hw,
                              const float* const* weights, const float* const* biases,
                              void* const* outs, int R, int C, int L, int NL, int Hh, int in_dtype,
                              int out_dtype, int out_layout, int precision, void* workspace,
                              size_t workspace_bytes, int max_cus, void* stream);

/* gd4d_value_proj_bwd_input / _bwd_weight - backward of gd4d_value_proj_fwd (fp32, pixel-major grad_out): what autograd
 * derives for `self.value_proj(value_flatten)` and the flatten / transpose / cat in front of it
 * (deform3d_cross_attn.py:264-280), without the transposed copies and on the bf16 MFMA with split operands (fp32-class).
 *   grad_out     (R, S, C) fp32: gradient of the projected value tensor, as gd4d_cross_attn_bwd writes it
 *   grad_feats   host array of L device pointers, level l = (R, C, H_l, W_l) fp32 NCHW:
 *                grad_feats_l[r, ci, pix] (+)= sum_co grad_out[r, start_l + pix, co] * weight[co, ci]
 *                accumulate != 0 adds to the tensors' contents (every decoder layer projects the same pyramid, so its
 *                gradient is the sum over the layers); 0 overwrites.
 *   grad_weight  (C, C): sum over rows of grad_out[row, co] * feats[row, ci];  grad_bias (C) or NULL: column sums of
 *                grad_out.  Partial sums per workgroup go to `workspace` (gd4d_value_proj_bwd_weight_workspace_bytes(),
 *                16-byte aligned) and are added in a fixed order: results are run-to-run identical.
 * Supported: C == 256, L <= 8. */
int gd4d_value_proj_bwd_input(const float* grad_out, const float* weight, float* const* grad_feats,
                              const int32_t* level_hw, int R, int C, int L, int accumulate, void* stream);
size_t gd4d_value_proj_bwd_weight_workspace_bytes(void);
int gd4d_value_proj_bwd_weight(const float* grad_out, const void* const* feats, const int32_t* level_hw,
                               float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes, int R,
                               int C, int L, void* stream);

/* --------------------------------------------------------------------------------------------
 * gd4d_linear_fwd - the decoder's small dense layers on the fp32 MFMA, with the elementwise
 * neighbours fused:  y = act((x [+ x2 for output columns < n_split]) W^T + bias) [+ r1] [+ r2]
 *
 * Replaces nn.Linear call sites of the hot path: deform3d_cross_attn.py:211 (cam_attention_weights),
 * :227 (deform_sampling_offsets), :281 (attention_weights) - all three on (query + query_pos), :204 -
 * :326 (output_proj, with the residual adds of :336), :334 (position_encoder Linears);
 * nn.MultiheadAttention in_proj (q,k from query+query_pos, v from query: n_split = 2C) / out_proj;
 * mmcv FFN Linears; Detr3DCrossAtten :373, :386.
 *   x, x2 (M, K) row stride ldx (x2 optional); w (N, K) row-major; bias (N) or NULL;
 *   r1 / r2 optional residuals (M, N) with row strides ldr1 / ldr2; y (M, N) row stride ldy.
 * n_split must be a multiple of 32 when x2 is given (or >= N).  fp32 in, fp32 accumulate.
 * flags: GD4D_LIN_RELU = ReLU on the output (before the residuals); GD4D_LIN_INV_SIGMOID_IN = apply the
 * reference's inverse_sigmoid (deform3d_cross_attn.py:16-31) to x while loading it (position_encoder
 * input, :334).
 * xsum: NULL, or (M, K) contiguous fp32 that receives x + x2 (needs x2, K % 4 == 0, ldx % 4 == 0, no
 * GD4D_LIN_INV_SIGMOID_IN): a training step keeps the sum for the weight gradient without a separate add launch.
 */
#define GD4D_LIN_RELU 1
#define GD4D_LIN_INV_SIGMOID_IN 2
#define GD4D_LIN_RELU_AFTER_LN 4
#define GD4D_LIN_WEIGHT_KN 8        /* gd4d_linear_fwd: `weight` is (K, N) row-major - y = x W (the input gradient of a Linear whose weight is W) */
#define GD4D_GEMM_RELU_IN 16       /* gd4d_gemm_bf16x3_fwd: ReLU on the elements of A as they are read (gd4d_gemm_tn_bf16x3: of B) */
#define GD4D_GEMM_MASK_C 32        /* gd4d_gemm_bf16x3_fwd: c holds a ReLU's output on entry; result written where it was > 0, else 0 */
int gd4d_linear_fwd(const float* x, const float* x2, const float* w, const float* bias,
                    const float* r1, const float* r2, float* y, int M, int K, int N, int n_split,
                    int flags, int ldx, int ldy, int ldr1, int ldr2, float* xsum, void* stream);

/* gd4d_linear_group_fwd - up to 4 Linear layers that share their input, in one launch:
 *   y_g = (x + x2) W_g^T + b_g,  g < G.
 * Deform3DCrossAttn applies cam_attention_weights, deform_sampling_offsets and attention_weights to the same
 * (query + query_pos) (deform3d_cross_attn.py:211, :227, :281): three launches become one.
 *   x, x2 (M, K) row stride ldx (x2 may be NULL); w, bias, y: HOST arrays of G device pointers (bias or its entries may
 *   be NULL), W_g (N_g, K) row-major, y_g (M, N_g) contiguous; n_out: host, G ints.  G <= 4.  xsum: as gd4d_linear_fwd. */
int gd4d_linear_group_fwd(const float* x, const float* x2, const float* const* w, const float* const* bias,
                          float* const* y, const int32_t* n_out, int G, int M, int K, int ldx, float* xsum, void* stream);

/* gd4d_linear_bwd_weight - weight / bias gradient of an nn.Linear over the query rows (training):
 *   grad_w[n][k] = sum_m grad_y[m][n] * x[m][k]   (N, K) contiguous;   grad_b[n] = sum_m grad_y[m][n]  (or NULL)
 * What autograd's `grad_output.t().mm(input)` computes for the decoder's dense layers (config ...ceph.py:71-89,
 * deform3d_cross_attn.py:104-111,204-228,326), where the library GEMM runs these M ~ 900, N, K <= 512 shapes on one
 * compute unit.  fp32 MFMA, one workgroup per 16 x 32 block of grad_w, fixed summation order.
 *   x (M, K) row stride ldx;  grad_y (M, N) row stride ldy.  accumulate != 0: the sums are ADDED to grad_w / grad_b (a
 *   training step that keeps every parameter gradient in one flat buffer lets the kernel accumulate there instead of
 *   leaving autograd an add per parameter). */
int gd4d_linear_bwd_weight(const float* x, const float* grad_y, float* grad_w, float* grad_b, int M, int K, int N,
                           int ldx, int ldy, int accumulate, void* stream);
/* gd4d_linear_bwd_weight_group - up to 16 independent gd4d_linear_bwd_weight problems in ONE launch (host arrays of `count`
 * device pointers; grad_b[i] may be NULL; dims = count x {M, K, N, ldx, ldy}).  A training step queues the weight gradients
 * of a decoder layer (nothing reads them before the optimizer) instead of issuing eleven half-empty launches. */
int gd4d_linear_bwd_weight_group(const void* const* x, const void* const* grad_y, void* const* grad_w, void* const* grad_b,
                                 const int32_t* dims, int count, int accumulate, void* stream);

/* gd4d_layernorm_fwd - y = LayerNorm(x [+ res]) * gamma + beta [, ReLU] over the last dim
 * (biased variance, eps inside the sqrt, like ATen).  Replaces the nn.LayerNorm of
 * position_encoder (deform3d_cross_attn.py:104-111) and the decoder layer's three norms.
 * C % 4 == 0, C <= 1024. */
int gd4d_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                       float* y, int M, int C, float eps, int relu, void* stream);

/* gd4d_linear_ln_fwd - y = [ReLU] LN( act((x [+ x2 for cols < n_split]) W^T + b) + r1 + r2 ), LayerNorm optional
 * (gamma == NULL: plain Linear with the same epilogue as gd4d_linear_fwd).  One workgroup per 16 complete rows: the
 * Linear -> LayerNorm pairs of a decoder layer (out_proj -> norm, output_proj + residual + pos_feat -> norm,
 * FFN layers[1] + residual -> norm, position_encoder[3] -> [4] -> ReLU; deform3d_cross_attn.py:108-110, 326-336 and the
 * mmcv layer of config ...ceph.py:71-89) become one launch each, and few fat workgroups instead of hundreds of small
 * ones when most CUs are busy with value_proj.
 *   flags: GD4D_LIN_RELU (before the residuals), GD4D_LIN_RELU_AFTER_LN; eps: LayerNorm eps.
 *   Supported: K % 64 == 0, ldx % 4 == 0; with LayerNorm N <= 256; n_split % 256 == 0 (or >= N). */
int gd4d_linear_ln_fwd(const float* x, const float* x2, const float* w, const float* bias, const float* r1,
                       const float* r2, const float* gamma, const float* beta, float* y, int M, int K, int N,
                       int n_split, int flags, float eps, int ldx, int ldy, int ldr1, int ldr2, void* stream);

/* gd4d_row_chain_fwd - a chain of ROW-LOCAL operations over blocks of 16 rows in ONE launch: everything a decoder layer
 * does between two points that need all queries at once (attention core, fused sample-aggregate) - out_proj + residual +
 * LayerNorm, the Linears of Deform3DCrossAttn on query + query_pos (deform3d_cross_attn.py:211, :227, :281),
 * output_proj + residuals (:326-336), mmcv FFN, the next layer's in_proj, the head's reg branch and reference-point
 * refinement (detr3d_transformer.py:199-214), position_encoder (:104-111).  A workgroup keeps its 16 rows in LDS buffers
 * 0 .. 3 (16 x 512 fp32 each) and executes `program` (host array of nops <= GD4D_CHAIN_MAX_OPS operations, copied into
 * the kernel argument) in order:
 *   LOAD       buf[dst][:, dst_col .. +N) = f(p0[m, :N]) (+ p1[m, :N]); row strides ld0, ld1; f = inverse_sigmoid with
 *              GD4D_CHAIN_INV_SIGMOID.  LOAD, ADD and SMALL_LINEAR also store their result to gout[m, :N] (row stride ldg) when
 *              it is given
 *   GEMM       v = act(buf[src][:, :K] . W^T + bias), p0 = the IMAGE of W (N, K) made by gd4d_chain_weight_image (bf16 hi /
 *              lo halves in MFMA fragment order; gd4d_chain_weight_image_bytes(N, K) bytes, 16-byte aligned; rebuild it when
 *              W changes), bias = p1 or NULL, act = ReLU with GD4D_CHAIN_RELU; then v += buf[res] (res >= 0) and v += (p2[m, n] + p3[m, n])
 *              (global addends, row strides ld2 / ld1; p3 optional) - residual rows without a LOAD operation in front; written to
 *              buf[dst][:, dst_col + n] (dst >= 0, dst != src) and / or gout[m, n] (row stride ldg).  K % 64 == 0, K <= 512.
 *   LAYERNORM  over N columns (N % 64 == 0) of buf[src]: gamma = p0, beta = p1, eps; ReLU with GD4D_CHAIN_RELU; to buf[dst] and / or gout;
 *              with p2 (row stride ld2) and res >= 0 a second output buf[res] = result + p2[m, :] (the ADD that would follow)
 *   ADD        buf[dst] = buf[src] + buf[res] (res >= 0) + p2[m, :N]
 *   SMALL_LINEAR  buf[dst][:, :N] = act(buf[src][:, :K] . W^T + bias) for K <= 8 (position_encoder's first Linear); with
 *              GD4D_CHAIN_INV_SIGMOID columns 8 .. 8 + K of buf[src] are used as scratch (the transformed inputs)
 *   REFINE     reference-point refinement: src = reg-branch output (>= 5 columns), p0 = reference points (M, 3) in [0, 1],
 *              gout = refined points (M, 3)
 *   HEADGEMM   value_proj of the per-head aggregates of gd4d_cross_attn_agg_fwd, read from GLOBAL memory (what
 *              gd4d_value_proj_heads_fwd computes, as the first operation of the chain that consumes it):
 *              v[m, n] = sum_k p2[m][h][k] W[n][k] + bias[n] p3[m][h], h = n / (N / heads); p0 = the image of W (N, K),
 *              p1 = bias or NULL, p2 = agg (M, heads, K), p3 = wsum (M, heads), heads = ld0; then as GEMM (+ buf[res], to
 *              buf[dst] and / or gout).  (N / heads) % 32 == 0, K % 64 == 0.
 *   LN_BWD     backward of LAYERNORM: buf[src] = gradient of its output, buf[res] - or, when p3 is given, the global rows
 *              p3[m, :N] (row stride ld1) - = the forward's input (statistics are recomputed), p0 = gamma, p1 = beta (needed with GD4D_CHAIN_RELU: the forward's ReLU); dx -> buf[dst] (dst == src
 *              allowed) and / or gout; p2 = (ceil(M / 16), 2, N) partial dgamma / dbeta of the row blocks, the layout of
 *              gd4d_layernorm_bwd's workspace (gd4d_layernorm_bwd_reduce_group adds them), or NULL
 *   DROPMASK   buf[dst][:, :N] = the dropout mask a forward GEMM with GD4D_CHAIN_DROPOUT drew (p0 = its seed, reserved = its
 *              threshold, eps = its scale, N = its N) applied to buf[src]: the gradient at that GEMM's output; also to gout
 * GEMMs: split-bf16 x3 on the bf16 MFMA with fp32 accumulation (fp32-class, the arithmetic of gd4d_value_proj_fwd);
 * everything else fp32.  M = number of rows.
 * A training step runs the same programs with every intermediate also written to global memory (gout), and backward programs
 * built from GEMMs over the TRANSPOSED weights' images (input gradients; GD4D_CHAIN_MASK_P2 at a ReLU), LN_BWD and ADD; the
 * weight gradients are contractions over all rows (gd4d_linear_bwd_weight_group) of what those programs wrote. */
enum { GD4D_CHAIN_LOAD = 1, GD4D_CHAIN_GEMM = 2, GD4D_CHAIN_LAYERNORM = 3, GD4D_CHAIN_ADD = 4, GD4D_CHAIN_REFINE = 5,
       GD4D_CHAIN_SMALL_LINEAR = 6, GD4D_CHAIN_HEADGEMM = 7, GD4D_CHAIN_SIGNAL = 8, GD4D_CHAIN_WAIT = 9, GD4D_CHAIN_LN_BWD = 10, GD4D_CHAIN_DROPMASK = 11 };
#define GD4D_CHAIN_RELU 1
#define GD4D_CHAIN_INV_SIGMOID 2
#define GD4D_CHAIN_SIGMOID 4        /* GEMM: sigmoid on the output (after the bias / ReLU, before the residual) */
#define GD4D_CHAIN_EXACT 8          /* GEMM: fp32-class products - both operands cut into three bf16 pieces, the six products of
                                       combined order <= 2 (~2^-24 relative); p0 = the image of gd4d_chain_weight_image_exact.  For
                                       the GEMMs that produce reference points (initial reference, reg branch): a point's error is
                                       multiplied by the 102-m range and the focal length before it selects pixels */
#define GD4D_CHAIN_SRC2 16          /* GEMM: columns >= ld0 (a multiple of 256) take their A operand from buffer `res` (which is then
                                       no residual): the packed in-projection - q, k from x + pos, v from x - as one operation */
#define GD4D_CHAIN_SPLIT_OUT 32     /* GEMM: the N columns leave for THREE global tensors: [0, ldg) -> gout, the next ld2 -> p2, the
                                       last ld1 -> p3 (each as wide as its row stride; p2 / p3 are outputs, not addends) - the
                                       Linears of one input stacked into one weight (deform3d_cross_attn.py:211, :227, :281) */
#define GD4D_CHAIN_MASK_P2 64       /* GEMM (a backward chain): p2[m, n] is the output a ReLU produced in the forward pass, not an
                                       addend - the result passes where it was > 0 (times eps when eps != 0), else 0 */
#define GD4D_CHAIN_DROPOUT 128      /* GEMM (training, modules in train mode): nn.Dropout on the output - after bias / activation, before
                                       the residuals: element (m, n) is kept iff gd4d_mha_dropout.h's hash of (the 64-bit seed at p3,
                                       m N + n) >= reserved (= round(p 2^32)); kept elements are multiplied by eps = 1 / (1 - p) */
#define GD4D_CHAIN_SPLIT_KV 256      /* GEMM with N = 768 (nn.MultiheadAttention's packed in-projection, C = 256, 8 heads of 32): gout
                                       receives the Q columns [0, 256) only; the K columns [256, 512) and the V columns [512, 768) leave
                                       as the split-bf16 OPERANDS of the attention core (x = hi + lo: the split gd4d_mha_core_fwd makes of every K / V row once per
                                       16-query tile, i.e. 57 times), in its MFMA fragment layout (a wave's load = 1 KB contiguous):
                                       p2 = K, bf16 [head][tile of 16 keys][lane = 16 (c / 8) + key % 16][c % 8] (c = channel within the
                                       head), hi plane then lo plane `ld2` elements further (ld2 >= 8 ceil(M / 16) 512);
                                       p3 = V, bf16 [head][step of 32 keys][d / 16][lane = 16 g + d % 16][j] with key = 32 step +
                                       16 (j >> 2) + 4 g + (j & 3), hi plane then lo plane `ld1` elements further (ld1 >= 8
                                       ceil(M / 32) 1024).  Rows past M of the last block are written too (finite).  p2 / p3 are
                                       outputs, not addends.  Read by gd4d_mha_core_presplit_fwd */
#define GD4D_CHAIN_SPLIT_KV_KEEP 512 /* with GD4D_CHAIN_SPLIT_KV: gout receives the K and V columns too (fp32, for gd4d_mha_core_bwd) */
#define GD4D_CHAIN_ADD_GOUT 1024     /* HEADGEMM: gout (M, N; row stride ldg) is READ, not written: v += gout[m, n] - the part of the sampled value
                                        gd4d_cross_attn_agg_items_coarse_fwd gathered from already projected rows (pagg); needs dst >= 0 */
#define GD4D_CHAIN_MAX_OPS 32
typedef struct gd4d_chain_op {
  int32_t kind, src, dst, res;      /* LDS buffer ids, -1 = none */
  int32_t K, N, flags, dst_col;
  int32_t ld0, ld1, ld2, ldg;       /* row strides of p0 / p1 (LOAD), p2, gout */
  float eps;
  int32_t reserved;
  const float* p0;
  const float* p1;
  const float* p2;
  float* gout;
  const float* p3;
} gd4d_chain_op;
size_t gd4d_chain_op_bytes(void);
size_t gd4d_chain_weight_image_bytes(int N, int K);
int gd4d_chain_weight_image(const float* weight, int N, int K, void* image, void* stream);
size_t gd4d_chain_weight_image_exact_bytes(int N, int K);
int gd4d_chain_weight_image_exact(const float* weight, int N, int K, void* image, void* stream);
/* gd4d_chain_weight_image_group - every image a training step needs in ONE launch (the weights change with each optimizer
 * step; inside a replayed hipGraph the host cannot notice).  jobs_device: `count` jobs in DEVICE memory, sorted by frag0.  A job's
 * logical matrix S = up to three row blocks seg[i] (rows[i] x cols, dense; unused: 0 rows) stacked; image = that of S
 * (N = sum rows, K = cols) or with transposed != 0 of S^T (N = cols, K = sum rows zero-padded to a multiple of 64);
 * planes = 2 (gd4d_chain_weight_image's layout), 3 (.._exact's) or 0 (no image: cols = 1, the segments are concatenated into
 * fp32 `image` - a stacked bias).  A job owns fragments [frag0, frag0 + n): n = ceil(N / 16) * (K_padded / 32) for an image,
 * ceil(sum rows / 64) for a concatenation; total_frags = the sum. */
#define GD4D_IMAGE_JOBS_MAX 1024   /* jobs per gd4d_chain_weight_image_group launch */
typedef struct gd4d_image_job {
  const float* seg[3];
  int32_t rows[3];
  int32_t cols, transposed, planes, frag0, reserved;
  void* image;
} gd4d_image_job;
size_t gd4d_image_job_bytes(void);
int gd4d_chain_weight_image_group(const gd4d_image_job* jobs_device, int count, int total_frags, void* stream);
int gd4d_row_chain_fwd(const gd4d_chain_op* program, int nops, int M, void* stream);
/* Two independent programs over the same M rows in ONE launch (twice the workgroups, each half runs one program on its own
 * compute units): e.g. chain A of a decoder layer next to the previous layer's reg branch + refinement + this layer's
 * position_encoder - what would otherwise need a second stream and two cross-stream dependencies per layer (10 us each in a
 * replayed graph against 2 us for a boundary on one stream).  nops_a + nops_b <= GD4D_CHAIN_MAX_OPS.
 * REFINE with dst >= 0 also parks the refined point in buf[dst][:, 0..2]; SMALL_LINEAR honours GD4D_CHAIN_INV_SIGMOID on its
 * inputs - together: refinement and position_encoder in one program without a round trip through global memory.
 * SIGNAL / WAIT (two-program launches only): a hand-off between the programs, row block by row block.  SIGNAL (gout = an
 * array of >= ceil(M / 16) uint32 flags, zero before the launch) publishes what its program has written to global memory for
 * its 16 rows so far; WAIT (p0 = the same array; gout = a uint32 error counter) holds its program until the OTHER
 * program's workgroup of the same rows has signalled - and takes the flag down again (the array is zero after the launch: one
 * buffer serves launch after launch, a replayed hipGraph needs no fill); the operation after a WAIT must be the LOAD of what was
 * handed over.
 * Row block i of both programs is dispatched to the same XCD (workgroups j and j + 8 k share one - the dispatcher's behaviour
 * today, which no specification promises: gd4d_xcd_placement_probe lets the host check it once per device before it builds such
 * programs), so the hand-off goes through that XCD's L2 without cache maintenance.  E.g. position_encoder next to chain B: its
 * rows are needed by chain B's second operation only.  SIGNAL belongs in program_a, WAIT in program_b (anything else is
 * GD4D_EINVAL): the workgroups of a launch are dispatched in index order, program_a's first, so a waiting workgroup's
 * producer is resident or finished.  A WAIT that is not answered within ~0.2 s gives up LOUDLY: it adds 1 to the error counter
 * at gout, and every LOAD its program executes from then on delivers NaN instead of the rows - the result cannot be mistaken for
 * a right one; the host reads the counter where it synchronises anyway (once per request, after the replayed graph) and raises. */
int gd4d_row_chain2_fwd(const gd4d_chain_op* program_a, int nops_a, const gd4d_chain_op* program_b, int nops_b, int M,
                        void* stream);

/* gd4d_row_chain_guest_fwd - gd4d_row_chain_fwd / gd4d_row_chain2_fwd (nops_b = 0: one program) with GUEST workgroups in the same launch
 * that run value_proj (deform3d_cross_attn.py:264-280: Linear 256 -> 256 over pixel rows) of ONE decoder layer over up to 4 pyramid
 * levels - the job of gd4d_value_proj_fwd with GD4D_LAYOUT_PIXEL_MAJOR fp32 output, same kernel body, same bits.  A chain keeps
 * ceil(M / 16) (two programs: twice that) compute units busy, latency-bound, while the rest of the device idles; the inference step
 * uses that window to project the NEXT layer's two coarse levels (43 800 rows at 24 cameras), which
 * gd4d_cross_attn_agg_items_coarse_fwd then gathers as 128-byte rows.  The guests are dispatched behind the chain's workgroups and
 * share nothing with them (no hand-off; the launch ends when both have ended).
 *   feats[l]: level l as (R, 256, H_l W_l) fp32 NCHW rows, or with chlast != 0 as (R, H_l W_l, 256) channels-last rows;
 *   image: gd4d_value_proj_image of the layer's weight (256, 256) and bias (256; NULL = zeros) - gd4d_value_proj_image_bytes() bytes,
 *   16-byte aligned, remade when either changes; out: (R, sum_l H_l W_l, 256) fp32, level 0's pixels first;
 *   workgroups: guest workgroups to add (0: as many as the chain leaves compute units free, at most one per 8 tiles of 32 pixels).
 * Inference programs only (GD4D_EUNSUPPORTED for the training operations). */
typedef struct gd4d_chain_guest {
  const void* feats[GD4D_MAX_LEVELS];
  int32_t level_hw[2 * GD4D_MAX_LEVELS];
  int32_t L, R, chlast, workgroups;
  const void* image;
  void* out;
} gd4d_chain_guest;
size_t gd4d_chain_guest_bytes(void);
size_t gd4d_value_proj_image_bytes(void);
int gd4d_value_proj_image(const float* weight, const float* bias, void* image, void* stream);
int gd4d_row_chain_guest_fwd(const gd4d_chain_op* program_a, int nops_a, const gd4d_chain_op* program_b, int nops_b, int M,
                             const gd4d_chain_guest* guest, void* stream);
/* gd4d_value_proj_guest_fwd - the same job as a launch of its own (the first decoder layer's, whose gather no chain precedes: it runs on
 * the side stream, beside the first layer's query side).  max_cus as gd4d_value_proj_fwd. */
int gd4d_value_proj_guest_fwd(const gd4d_chain_guest* job, int max_cus, void* stream);

/* gd4d_mha_core_presplit_fwd - gd4d_mha_core_fwd (batch 1, Lq = Lk = L; mask / mask_kind, lse, drop_p / seed as there) with K and V handed over as the
 * split-bf16 operands of its MFMAs instead of fp32 rows: the two plane pairs a GEMM operation with GD4D_CHAIN_SPLIT_KV writes
 * beside the fp32 in-projection (layouts there; k_plane_stride / v_plane_stride = elements from the hi to the lo plane).  The
 * kernel converted every K / V row once per 16-query tile (57 times at 900 queries: 112 of its 313 vector instructions per 32
 * keys) and fetched V with 16 dword loads per step; here a step is eight 16-byte loads, each 1 KB contiguous per wave.
 * Results bit-identical to gd4d_mha_core_fwd on the same q, k, v without dropout; with drop_p > 0 that entry point runs its
 * fp32-MFMA kernel: the same elements dropped, values equal to ~2^-16 relative. */
int gd4d_mha_core_presplit_fwd(const float* q, const void* k_planes, const void* v_planes, float* out, int L, int H, int D,
                               int ldq, int ldo, long long k_plane_stride, long long v_plane_stride, const void* mask,
                               int mask_kind, float scale, float* lse, float drop_p, const void* seed, void* stream);

/* gd4d_mlp2_bf16x3_fwd - out = relu(x W1^T + b1) W2^T + b2 in ONE kernel (gd4d_mlp2.hip): the head's position-embedding MLPs -
 * Conv1x1, ReLU, Conv1x1 over every pixel of every camera (dense_heads/detr3d_head_pe.py:380-390 `position_encoder`, applied at
 * :543-553: 192 -> 1024 -> 256 over 739 800 pixels at 24 cameras).  The hidden activation (3 GB as a tensor) stays in registers:
 * per chunk of 32 hidden units a wave computes the TRANSPOSED first product, whose accumulator layout is the A operand of the
 * second.  Arithmetic as gd4d_gemm_bf16x3_fwd (split bf16 x 3 on the bf16 MFMA, fp32 accumulation).  x (M, K1) fp32 rows of
 * stride ldx; image: gd4d_mlp2_image of W1 (H, K1), b1 (H; NULL = zeros) and W2 (N2, H) (gd4d_mlp2_image_bytes; remake it when one
 * of them changes); b2 (N2) or NULL; out (M, N2) rows of stride ldo.  K1 in {16, 32, 64, 128, 192, 256}, H % 32 == 0, N2 == 256. */
size_t gd4d_mlp2_image_bytes(int K1, int H, int N2);
int gd4d_mlp2_image(const float* w1, const float* b1, const float* w2, int K1, int H, int N2, void* image, void* stream);
int gd4d_mlp2_bf16x3_fwd(const float* x, const void* image, const float* b2, float* out, int M, int K1, int H, int N2, int ldx,
                         int ldo, void* stream);
/* gd4d_mlp2_se_fuse_fwd - the feature-dependent half of the head's position embedding as ONE kernel: SELayer's gate
 * (models/dense_heads/detr3d_head_pe.py:236-243: conv_reduce -> ReLU -> conv_expand -> sigmoid) and the adds of :553-557,
 *     out[r, pix, :] = feat[r, :, pix] + (pe[m, :] * sigmoid(relu(feat[r, :, pix] W1^T + b1) W2^T + b2) + sine[m, :]),
 * over the pixels of L <= 4 NCHW levels feats[l] (R, C, H_l, W_l) laid side by side (row m = r S + start_l + pix, S = sum H_l W_l);
 * pe / sine (R S, C) channels-last rows (gd4d_mlp2_bf16x3_fwd's / gd4d_gemm_bf16x3_fwd's output), image = gd4d_mlp2_image of
 * conv_reduce's (H, C) weight, its bias and conv_expand's (C, H) weight, b2 = conv_expand's bias.  outs[l] (R, H_l W_l, C): the
 * level CHANNELS-LAST - what gd4d_cross_attn_agg_items_fwd gathers in place.  Replaces gd4d_value_proj_fwd + gd4d_gemm_bf16x3_fwd +
 * gd4d_se_fuse_chlast_fwd (two 757-MB intermediates at 24 cameras).  C = 256, H % 32 == 0. */
int gd4d_mlp2_se_fuse_fwd(const void* const* feats, const int32_t* level_hw, int L, int R, const void* image, const float* b2,
                          const float* pe, const float* sine, void* const* outs, int C, int H, void* stream);
/* gd4d_mlp2_frustum_fwd - position_encoder(frustum coordinates) as ONE kernel that reads no X: gd4d_frustum_pe_input_fwd's arithmetic
 * (dense_heads/detr3d_head_pe.py:427-491: pixel centre x depth bin -> lidar frame by the camera's img2lidar -> pc_range units ->
 * inverse_sigmoid; operation for operation) runs in the MLP's prologue - the 3 D = 192 inputs of a row are a function of 12 matrix
 * entries and the pixel index, generated in registers where gd4d_mlp2_bf16x3_fwd loads them; the (R, S, 192) tensor between the two
 * kernels (568 MB written and read at 24 cameras) is gone.  img2lidar (R, 16); level_hw: L <= 4 levels laid side by side (row m =
 * r S + start_l + y W_l + x); pad_h / pad_w, D (64 only), depth_start, pc_range: as gd4d_frustum_pe_input_fwd; image: gd4d_mlp2_image of
 * W1 with its COLUMNS PERMUTED - column 16 st + 8 kg + e of the image's W1 is column 96 kg + 8 st + e of position_encoder[0]'s weight
 * (a lane then holds depth bins 32 kg .. 32 kg + 31: the axis of each register is a compile-time constant) -, b1, W2; out (R S, 256),
 * row stride ldo.  Same values as the two kernels up to the summation order of the first product. */
int gd4d_mlp2_frustum_fwd(const float* img2lidar, const int32_t* level_hw, int L, int R, float pad_h, float pad_w, int D,
                          float depth_start, const double* pc_range, const void* image, const float* b2, float* out, int H, int ldo,
                          void* stream);
/* gd4d_mlp2_pe_se_fwd - gd4d_mlp2_frustum_fwd and gd4d_mlp2_se_fuse_fwd as ONE kernel, for cameras whose embedding nobody keeps (the
 * past-frame cameras of the temporal pattern, whose matrices change with every sample; or every camera): per tile of 128 pixels the
 * position MLP's result waits in 128 registers while the gate's MLP runs, the epilogue is the head's fuse (:553-557) - the (R S, 256)
 * embedding between the two kernels (757 MB written and read at 24 cameras) does not exist.  Arguments: the geometry of
 * gd4d_mlp2_frustum_fwd (pe_image: its permuted image; pe_b2, pe_H), the maps / sine / outs of gd4d_mlp2_se_fuse_fwd (se_image, se_b2,
 * se_H); pe_out: NULL, or (R S, 256) to store the embedding as well.  The same values as the two kernels bit for bit. */
int gd4d_mlp2_pe_se_fwd(const float* img2lidar, const void* const* feats, const int32_t* level_hw, int L, int R, float pad_h,
                        float pad_w, int D, float depth_start, const double* pc_range, const void* pe_image, const float* pe_b2,
                        int pe_H, const void* se_image, const float* se_b2, int se_H, const float* sine, void* const* outs,
                        float* pe_out, void* stream);

/* gd4d_adamw_flat - the optimizer step of the reference's training recipe over ONE flat fp32 parameter / gradient buffer: clipping
 * of the gradient's L2 norm (torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (norm + 1e-6)); max_norm <= 0: none) followed by
 * AdamW (torch.optim.AdamW's update, decoupled weight decay) - optimizer + optimizer_config of
 * projects/configs/detr4d/detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:205-213 (AdamW lr 2e-4, weight_decay 0.01,
 * grad_clip max_norm 35, norm_type 2).  Two launches whatever the number of parameters.  params / grads / exp_avg / exp_avg_sq: n
 * floats each, 16-byte aligned (exp_avg, exp_avg_sq zero before the first step); state: 2 floats on the device - [0] the step
 * counter (zero before the first step; advanced here, so a replayed hipGraph keeps counting), [1] receives the norm before
 * clipping; workspace: gd4d_adamw_flat_workspace_bytes().  The norm's partial sums are added in a fixed order (run-to-run identical). */
size_t gd4d_adamw_flat_workspace_bytes(void);
int gd4d_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* state, void* workspace,
                    size_t workspace_bytes, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float max_norm, void* stream);

/* gd4d_xcd_placement_probe - out[b] = the XCC (XCD) id workgroup b of a `blocks`-workgroup launch ran on: the self-test behind
 * SIGNAL / WAIT above (expected: out[b] == out[b % 8]). */
int gd4d_xcd_placement_probe(int32_t* out, int blocks, void* stream);

/* gd4d_small_linear_layernorm_fwd - y = [ReLU] LN( f(in) W^T + b ) for a Linear with at most 4 inputs: the first stage of
 * position_encoder, Linear(3 or 4 -> 256), LayerNorm, ReLU on inverse_sigmoid(reference points)
 * (deform3d_cross_attn.py:104-111, :331-334), as one launch.  in (M, Kin), w (C, Kin), bias (C) or NULL, y (M, C);
 * flags: GD4D_LIN_RELU on the output, GD4D_LIN_INV_SIGMOID_IN on the input.  Kin <= 4, C % 4 == 0, C <= 1024. */
int gd4d_small_linear_layernorm_fwd(const float* in, const float* w, const float* bias, const float* gamma,
                                    const float* beta, float* y, int M, int Kin, int C, float eps, int flags,
                                    void* stream);

/* --------------------------------------------------------------------------------------------
 * gd4d_mha_core_fwd - softmax(q k^T * scale [+ mask]) v per head, never materialising the score
 * tensor.  The middle of nn.MultiheadAttention as used by the decoder self-attention (mmcv
 * MultiheadAttention; config .../detr4d_res50_deform_pe_testaug_320_fullset_ceph.py:74-78;
 * H-DETR mask: h_detr3d_transformer.py:149-157).
 *   q (Lq*B rows), k, v (Lk*B rows): row index l*B + b, row strides ldq/ldk/ldv, head h at column
 *   offset h*D (so q, k, v may alias the three thirds of one packed in-projection buffer);
 *   mask: NULL, or (Lq, Lk) uint8 with nonzero = masked (mask_kind 1), or (Lq, Lk) fp32 additive
 *   (mask_kind 2);  out (Lq*B, H*D) row stride ldo.  Supported: D == 32.  Softmax in fp32 (base 2); the two products
 *   on the fp32 MFMA when lse or dropout is asked for (training: the backward kernels recompute the probabilities in
 *   fp32), else (inference) as split-bf16 x3 products on the bf16 MFMA with fp32 accumulation (~2^-16 relative per
 *   product, the arithmetic of the GEMMs around it; GD4D_MHA_FP32=1 in the environment keeps the fp32 kernel).
 */
int gd4d_mha_core_fwd(const float* q, const float* k, const float* v, const void* mask, float* out,
                      int Lq, int Lk, int B, int H, int D, int ldq, int ldk, int ldv, int ldo,
                      int mask_kind, float scale, float* lse, float drop_p, const void* seed, void* stream);
/* lse: NULL, or (Lq, B, H) fp32 that receives log sum_k exp(scale q k^T + mask) per (query, batch, head) - what
 * gd4d_mha_core_bwd needs from the forward.
 * drop_p in [0, 1), seed: dropout of the probabilities as nn.MultiheadAttention applies it in training (F.dropout on the
 * softmax output: the reference's attn_drop = 0.1): with drop_p > 0 each probability is kept with chance 1 - drop_p and
 * scaled by 1 / (1 - drop_p) before it multiplies v; lse stays that of the full softmax.  seed = two uint32 words in
 * DEVICE memory, read by the kernel (so a captured launch draws a new mask when the words are advanced on the device);
 * element (b, h, q, key) is kept iff mix(seed, ((b H + h) Lq + q) Lk + key) >= round(drop_p 2^32), mix as in
 * csrc/gd4d_mha_dropout.h (two multiply / xor-shift rounds) - a function of (seed, element) only, so the backward
 * regenerates it and no mask is stored.  B H Lq Lk must be < 2^32 with drop_p > 0.  drop_p == 0: seed is not read.
 *
 * gd4d_mha_core_bwd - backward of the above (what autograd derives for the bmm / softmax / bmm inside
 * nn.MultiheadAttention): given dout (gradient of out), dq = scale dS k, dk = scale dS^T q, dv = P^T dout with
 * P = exp(scale q k^T + mask - lse), dS = P o (dout v^T - Dq), Dq = sum_d dout o.  o = the forward's output, lse = its
 * saved statistic, dsum = (Lq, B, H) fp32 scratch (receives Dq).  dq / dk / dv: row strides lddq / lddk / lddv, head h
 * at column offset h*D.  Two launches (a workgroup per 16 queries, then per 16 keys), fixed summation order.
 * drop_p / seed: the forward's (the seed words must hold what the forward read): with M = keep / (1 - drop_p),
 * dv = (P o M)^T dout and dS = P o (M o dout v^T - Dq). */
int gd4d_mha_core_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout, const void* mask,
                      const float* lse, float* dsum, float* dq, float* dk, float* dv, int Lq, int Lk, int B, int H, int D,
                      int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, int mask_kind,
                      float scale, float drop_p, const void* seed, void* stream);

/* gd4d_mha_core_bwd_fill - gd4d_mha_core_bwd whose second launch (dk / dv) also carries gd4d_pyramid_grad_fill of one or two
 * plans as guest workgroups (jobs: each plan with its slots, first table row and query order; fill_B / fill_N / fill_Hh / fill_P
 * as gd4d_pyramid_grad_fill's B, N, Hh, P; start / records as there).  The fills need the scan over all layers' counts and nothing
 * from the backward pass - a training step hands them to its attention backward launches, whose matrix / vector work hides their
 * memory traffic.  Same results as the separate launches. */
typedef struct gd4d_fill_job {
  const void* plan;             /* pairs form (header + pairs) */
  const void* slots;            /* what gd4d_pyramid_grad_count wrote for it */
  const int32_t* query_order;   /* the plan's (may be NULL) */
  uint32_t id_base;             /* first table row of the plan's layer */
  int32_t Q;                    /* the plan's query count */
} gd4d_fill_job;
int gd4d_mha_core_bwd_fill(const float* q, const float* k, const float* v, const float* o, const float* dout, const void* mask,
                           const float* lse, float* dsum, float* dq, float* dk, float* dv, int Lq, int Lk, int B, int H, int D,
                           int ldq, int ldk, int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, int mask_kind, float scale,
                           float drop_p, const void* seed, const gd4d_fill_job* jobs, int njobs, const int32_t* start,
                           void* records, int fill_B, int fill_N, int fill_Hh, int fill_P, void* stream);

/* gd4d_row_chain_fill_fwd - a TRAINING row chain (program_b / nops_b: a second program, or NULL / 0) that carries up to two layers'
 * record fills of the pyramid gradient as guest workgroups - the jobs of gd4d_mha_core_bwd_fill, on the ~200 compute units a backward
 * chain leaves idle for 50-70 us instead of in the attention backward's dk / dv launch (which they made 20-38 us longer).  workgroups:
 * guest workgroups (0: two per compute unit); they sit behind the chain's workgroups, stay, and walk the (position, head) rows.  Same
 * records as gd4d_pyramid_grad_fill. */
int gd4d_row_chain_fill_fwd(const gd4d_chain_op* program_a, int nops_a, const gd4d_chain_op* program_b, int nops_b, int M,
                            const gd4d_fill_job* jobs, int njobs, const int32_t* start, void* records, int fill_B, int fill_N,
                            int fill_Hh, int fill_P, int workgroups, void* stream);

/* gd4d_layernorm_bwd - backward of gd4d_layernorm_fwd (y = [ReLU] LN(x [+ res]) gamma + beta): dx (also the gradient of
 * res), dgamma, dbeta from dy; mean / rstd are recomputed from x.  beta is read only with relu != 0 (to rebuild the
 * sign of the forward's output).  workspace: gd4d_layernorm_bwd_workspace_bytes(M, C) bytes (per-workgroup partial
 * column sums, added in a fixed order).  C % 4 == 0, C <= 1024.  flags: GD4D_LN_RELU (the forward applied a ReLU),
 * GD4D_LN_ACCUMULATE (dgamma / dbeta are added to, not overwritten).  Replaces ATen's layer_norm_backward kernels behind
 * the decoder layer's norms and position_encoder (deform3d_cross_attn.py:104-111). */
#define GD4D_LN_RELU 1
#define GD4D_LN_ACCUMULATE 2
#define GD4D_LN_DEFER_REDUCE 4   /* dx and the per-workgroup partials only (dgamma / dbeta may be NULL): gd4d_layernorm_bwd_reduce_group later */
size_t gd4d_layernorm_bwd_workspace_bytes(int M, int C);
int gd4d_layernorm_bwd(const float* x, const float* res, const float* gamma, const float* beta, const float* dy, float* dx,
                       float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int M, int C, float eps,
                       int flags, void* stream);
/* gd4d_layernorm_bwd_reduce_group - the column reduces (dgamma, dbeta from the workspaces' partials) of up to 32 earlier
 * gd4d_layernorm_bwd calls made with GD4D_LN_DEFER_REDUCE, in ONE launch: host arrays of `count` device pointers, dims =
 * count x {M, C} of those calls; accumulate != 0: added to dgamma / dbeta. */
int gd4d_layernorm_bwd_reduce_group(const void* const* workspaces, void* const* dgamma, void* const* dbeta, const int32_t* dims,
                                    int count, int accumulate, void* stream);

/* gd4d_inverse_sigmoid_fwd - y = log(clamp(x, eps, 1) / clamp(1 - x, eps, 1)) with x clamped to [0, 1], eps = 1e-5: the
 * reference's inverse_sigmoid (deform3d_cross_attn.py:16-31) as ONE launch for callers that need the tensor itself (the
 * input of position_encoder's first Linear in a training step, where the reference points carry no gradient; the
 * inference kernels apply it while loading). */
int gd4d_inverse_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
/* gd4d_inverse_sigmoid_bwd - grad_x = grad_y * d inverse_sigmoid(x) / dx (+ add, when given): what autograd derives for it
 * (1 / x where x > eps, plus 1 / (1 - x) where 1 - x > eps; 0 outside [0, 1]).  Layer 0's reference points in a training step. */
int gd4d_inverse_sigmoid_bwd(const float* x, const float* grad_y, const float* add, float* grad_x, int64_t n, void* stream);

/* gd4d_refine_reference_fwd - reference-point refinement between decoder layers
 * (Detr3DTransformerDecoder.forward, detr3d_transformer.py:201-214):
 *   out[:, 0:2] = sigmoid(tmp[:, 0:2] + inverse_sigmoid(ref[:, 0:2]));
 *   out[:, 2]   = sigmoid(tmp[:, 4]   + inverse_sigmoid(ref[:, 2]))
 * tmp (M, ldt) = reg_branches[lid](output), ldt >= 5; ref, out (M, 3) fp32. */
int gd4d_refine_reference_fwd(const float* tmp, const float* ref, float* out, int M, int ldt, void* stream);

/* --------------------------------------------------------------------------------------------
 * The step right after the decoder: head epilogue + NMS-free box decoding.
 *
 * gd4d_box_head_fwd - Detr3DHeadPE.forward's per-layer box epilogue
 * (projects/mmdet3d_plugin/models/dense_heads/detr3d_head_pe.py:571-600):
 *   out = tmp;  out[:,0:2] = sigmoid(tmp[:,0:2] + inverse_sigmoid(ref[:,0:2])) * (hi-lo) + lo  [* scale]
 *               out[:,4]   = sigmoid(tmp[:,4]   + inverse_sigmoid(ref[:,2]))   * (hi-lo) + lo  [* scale]
 * tmp, out (M, code) fp32 (may alias), ref (M, 3) in [0,1]; pc_range: HOST, 6 doubles; the span hi-lo is
 * formed in double and rounded to fp32 as the reference's Python-scalar arithmetic does; scale = 1 unless the
 * head runs with scale_pred (img_metas[0]['depth_factors'][0], :591-593).  code >= 5.
 */
int gd4d_box_head_fwd(const float* tmp, const float* ref, const double* pc_range, float scale, float* out,
                      int M, int code, void* stream);

/* gd4d_nms_free_decode_fwd - NMSFreeCoder.decode_single for every batch element in one launch
 * (projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py:47-96; denormalize_bbox,
 * projects/mmdet3d_plugin/core/bbox/util.py:58-87):
 *   scores = sigmoid(cls_scores).view(-1);  (score, index) = top-K, descending (ties: lower index first);
 *   labels = index % C;  box = bbox_preds[index / C] de-normalised to (cx, cy, cz, exp w, exp l, exp h, atan2(sin, cos)
 *   [, vx, vy]);  keep = centre inside post_center_range (closed) [and score > score_threshold].
 * cls_scores (B, Q, C) fp32 logits; bbox_preds (B, Q, code), code 10 -> 9 box columns, code 8 -> 7;
 * post_center_range: HOST, 6 floats (lo xyz, hi xyz); score_threshold < 0 disables the score test;
 * boxes (B, K, 9|7), scores (B, K) fp32, labels (B, K) int32, keep (B, K) uint8.  The reference's final
 * boolean compaction (a data-dependent size) stays with the caller.  K <= 1024, Q*C <= 32768 and K <= Q*C
 * (the reference's topk raises otherwise; so does the host side).
 */
int gd4d_nms_free_decode_fwd(const float* cls_scores, const float* bbox_preds, const float* post_center_range,
                             float score_threshold, float* boxes, float* scores, int32_t* labels, uint8_t* keep,
                             int B, int Q, int C, int code_size, int K, void* stream);

/* --------------------------------------------------------------------------------------------
 * The step feeding the path: Detr3DHeadPE's feature position embedding
 * (projects/mmdet3d_plugin/models/dense_heads/detr3d_head_pe.py:427-491, :525-557).  Its 1x1 convolutions are plain
 * GEMMs and stay with the library; these are the bandwidth-bound pieces.
 *
 * gd4d_frustum_pe_input_fwd - position_embeding :438-481 up to the input of position_encoder: for every pixel centre
 * (x * pad_w / W, y * pad_h / H) and depth bin d_i = depth_start + bin * i * (i + 1), bin = (pc_range[3] -
 * depth_start) / (D (D + 1)): p = img2lidar @ [u * max(d, eps), v * max(d, eps), d, 1], normalised by pc_range,
 * inverse_sigmoid.  img2lidar (R = B*N, 4, 4) fp32 (the reference inverts lidar2img with numpy on the host);
 * out (R, 3*D, H, W) fp32, channel = 3 * i + axis; outside (R, H, W) uint8 = more than D/2 of the 3*D normalised
 * coordinates fall outside [0, 1] (:477-478; the caller ORs it with the padding mask).
 */
int gd4d_frustum_pe_input_fwd(const float* img2lidar, float* out, uint8_t* outside, int R, int H, int W, int D,
                              float pad_h, float pad_w, float depth_start, const double* pc_range, int row_pixels,
                              int row_start, void* stream);
/*   row_pixels == 0: the NCHW layout above.  row_pixels > 0: channels-last, out is (R, row_pixels, 3*D) and this level
 *   fills pixels [row_start, row_start + H*W) of every row (all levels of the pyramid side by side, the A operand of
 *   gd4d_gemm_bf16x3_fwd). */

/* gd4d_sine_pe3d_fwd - SinePositionalEncoding3D (models/utils/positional_encoding.py:82-99) after the cumulative
 * sums: out[r, part*F + f, pix] = f < F/2 ? sin(e / dim_t[2f]) : cos(e / dim_t[2(f - F/2) + 1]), e = embed_part[r, pix],
 * part = camera, row, column (the reference stacks sin / cos BEFORE the feature axis, :90-98).
 * n_embed / y_embed / x_embed (R, HW) fp32, dim_t (F) fp32, out (R, 3*F, HW). */
int gd4d_sine_pe3d_fwd(const float* n_embed, const float* y_embed, const float* x_embed, const float* dim_t,
                       float* out, int R, int HW, int F, int row_pixels, int row_start, void* stream);
/*   row_pixels == 0: (R, 3*F, HW) as above; row_pixels > 0: channels-last (R, row_pixels, 3*F), this level at pixels
 *   [row_start, row_start + HW) (the A operand of gd4d_gemm_bf16x3_fwd for adapt_pos3d). */

/* gd4d_se_fuse_fwd - out = feat + (pe * sigmoid(gate) + sine): SELayer's gate (:243) and the adds of :553-557 in one
 * pass over n elements (n % 4 == 0, 16-byte aligned pointers; out may alias feat). */
int gd4d_se_fuse_fwd(const float* feat, const float* gate, const float* pe, const float* sine, float* out, size_t n,
                     void* stream);
/* gd4d_se_fuse_chlast_fwd - the same with channels-last gate / pe (R, row_pixels, C) (this level at pixels
 * [row_start, row_start + HW)) against NCHW feat / sine / out (R, C, HW): a tiled transpose through LDS.  C % 32 == 0. */
int gd4d_se_fuse_chlast_fwd(const float* feat, const float* gate, const float* pe, const float* sine, float* out, int R,
                            int C, int HW, int row_pixels, int row_start, int sine_chlast, int out_chlast, void* stream);
/*   sine_chlast != 0: `sine` is channels-last (R, row_pixels, C) like gate / pe instead of NCHW.
 *   out_chlast != 0 (needs sine_chlast): `out` is stored (R, HW, C) - the level in the layout the decoder's gathers read in place
 *   (gd4d_cross_attn_agg_items_fwd with pixel stride C * 4: no per-sample slice-planar copy); the same bits as the NCHW result. */
/* gd4d_se_fuse_chlast_bwd - its backward for one level (what autograd derives from detr3d_head_pe.py:241-243, :556):
 * grad_out NCHW (R, C, HW) -> channels-last rows [row_start, row_start + HW) of grad_sine = g, grad_pe = g sigmoid(gate),
 * grad_gate = g pe sigmoid'(gate).  grad_pe / grad_gate may alias pe / gate (in place).  The gradient of `feat` through
 * the sum is grad_out itself. */
int gd4d_se_fuse_chlast_bwd(const float* grad_out, const float* gate, const float* pe, float* grad_gate, float* grad_pe,
                            float* grad_sine, int R, int C, int HW, int row_pixels, int row_start, void* stream);

/* --------------------------------------------------------------------------------------------
 * gd4d_gemm_bf16x3_fwd - C = act(A W^T + b): row-major fp32 A (M, K) and C (M, N), W (N, K) given as its bf16 split
 * (w = w_hi + w_lo, made once per weight by gd4d_split_bf16_fwd); three bf16 MFMA products per output accumulate in fp32
 * (fp32-class: about 2^-16 relative per product).  The 1x1 convolutions of the head's feature position embedding
 * (detr3d_head_pe.py:380-390 applied at :481-482, :551-556) in channels-last form.  flags: GD4D_LIN_RELU (output),
 * GD4D_GEMM_RELU_IN (input: the previous layer's activation).
 * Supported: N % 256 == 0, K % 32 == 0, lda % 4 == 0.
 */
int gd4d_split_bf16_fwd(const float* w, uint16_t* hi, uint16_t* lo, size_t n, void* stream);
int gd4d_gemm_bf16x3_fwd(const float* a, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias, float* c, int M,
                         int N, int K, int lda, int ldc, int flags, void* stream);
/* GD4D_GEMM_MASK_C: the input gradient of Linear -> ReLU -> Linear's first half: A = gradient of the second Linear's
 * output, W = its weight transposed, c = the ReLU's output of the forward, replaced in place by the gradient at the
 * ReLU's input.
 *
 * gd4d_gemm_tn_bf16x3 - C (M, N) = sum_r A[r, :M]^T B[r, :N] over R rows, colsum (M) = sum_r A[r, :] (NULL: skipped): the
 * weight and bias gradients of a Linear / 1x1 convolution over R pixels (A = output gradient (R, M), B = the layer's
 * input (R, N)), i.e. what autograd computes for detr3d_head_pe.py:380-390's convolutions.  Same split-bf16 x 3
 * arithmetic; the rows are cut into ranges whose partial products go to `workspace`
 * (gd4d_gemm_tn_bf16x3_workspace_bytes(R, M, N)) and are added in range order - the same bits every run.
 * flags: GD4D_GEMM_RELU_IN = ReLU on B as it is read.  Supported: M % 128 == 0, N % 64 == 0, lda % 4 == 0, ldb % 2 == 0.
 */
size_t gd4d_gemm_tn_bf16x3_workspace_bytes(long long R, int M, int N);
int gd4d_gemm_tn_bf16x3(const float* a, const float* b, float* c, float* colsum, void* workspace, long long R, int M, int N,
                        int lda, int ldb, int flags, void* stream);

/* --------------------------------------------------------------------------------------------
 * DGCNNAttn (projects/mmdet3d_plugin/models/utils/dgcnn_attn.py:10-96), registered by the reference, used by no
 * shipped config.
 *
 * gd4d_knn_farthest_fwd - edge_feats :84-86: for every row of x (B, N, C) the indices of the K rows of the same sample at
 * the LARGEST Euclidean distance (the reference takes topk of cdist), descending, ties to the lower index.
 * idx (B, N, K) int32.  N <= 2048, C % 4 == 0, K <= N.
 *
 * gd4d_edge_conv_max_fwd - conv + BatchNorm(eval) + ReLU + max over K of the edge features (:72-74, :77-78) with the
 * 1x1 convolution split as W[:, :C] x_j + W[:, C:] x_i:  out[b, n, c] = max_k relu((a[b, idx[b,n,k], c] +
 * b_self[b, n, c]) * scale[c] + shift[c]);  a / b_self (B*N, C) with row stride lda (two column blocks of one
 * (B*N, 2C) Linear output), scale / shift (C) = the BatchNorm affine of eval mode, out (B*N, C) contiguous.
 */
int gd4d_knn_farthest_fwd(const float* x, int32_t* idx, int B, int N, int C, int K, void* stream);
int gd4d_edge_conv_max_fwd(const float* a, const float* b_self, const int32_t* idx, const float* scale,
                           const float* shift, float* out, int B, int N, int C, int K, int lda, void* stream);

/* --------------------------------------------------------------------------------------------
 * gd4d_cross_attn_bwd - backward of gd4d_cross_attn_fwd.
 *
 * Replaces what autograd runs in the reference for deform3d_cross_attn.py:220-324: the third-party mmcv
 * `ms_deformable_col2im` kernel (grad of value by atomic adds, grad of sampling locations and
 * attention weights) plus the elementwise backward chain through softmax * mask, the sigmoid camera
 * weights (with the raw-view scramble), the divisions and the lidar2img matmul.  The visibility
 * mask carries no gradient (piecewise constant), as in the reference.
 *
 *   inputs as gd4d_cross_attn_fwd, plus grad_out (B, Q, Hh*Dh) = dL/d out
 *   grad_value        same shape/layout as value, fp32, MUST be zero-filled by the caller (atomic adds)
 *   grad_ref          (B, Q, 3)        dL/d reference_points (normalised units)
 *   grad_offsets      (B, Q, Hh, P, 3) dL/d offsets (metres)
 *   grad_attn_logits  (B, Q, Hh, L, P)
 *   grad_cam_logits   (B, Q, N)        in the un-scrambled layout of cam_logits
 *   workspace         B > 1 only: gd4d_cross_attn_bwd_workspace_bytes(B, Q, Hh, L, P) bytes.  The forward weights value row
 *                     i = b*N + n with the logits of batch (i % B) (:277), so grad_attn_logits[bb] collects contributions of
 *                     every sample b: each (b, q) workgroup writes its partial per logit class, a second launch adds them
 *                     over b in a fixed order and applies the softmax backward.  NULL / 0 for B == 1.
 *   flags             GD4D_CA_RAW_CAM_WEIGHTS as in the forward (Deform3DCrossAttnMP's neighbour pass).
 * Supported: B <= 8, fp32 pixel-major value, L <= 4.
 * Accumulation order of grad_value is not deterministic (fp32 atomics), like the mmcv kernel.
 * query_order: optional, as in gd4d_cross_attn_fwd (scheduling only: the atomic adds into grad_value of queries that
 * look at the same camera region meet in one XCD's L2; 847 -> 826 us at the headline size).
 */
int gd4d_cross_attn_bwd(const void* value, const int32_t* level_hw, const float* ref, const float* offsets,
                        const float* attn_logits, const float* cam_logits, const float* lidar2img,
                        const double* pc_range, float img_h, float img_w, const float* grad_out,
                        void* grad_value, float* grad_ref, float* grad_offsets, float* grad_attn_logits,
                        float* grad_cam_logits, int B, int N, int Q, int Hh, int Dh, int L, int P,
                        int value_dtype, int value_layout, int flags, const int32_t* query_order, void* workspace,
                        size_t workspace_bytes, void* stream);
size_t gd4d_cross_attn_bwd_workspace_bytes(int B, int Q, int Hh, int L, int P);

/* --------------------------------------------------------------------------------------------
 * Training-side step right after the path: HungarianAssigner3D's cost matrix and the per-layer
 * losses of Detr3DHeadPE.loss_single for ALL decoder layers at once, so that a step has one device -> host copy (the
 * cost matrices; the assignment itself runs on the host like the reference's scipy call), one host -> device copy and
 * no .item().
 *
 * gd4d_match_cost_fwd - cost[l][b] (Q, G_b) = FocalLossCost(cls)[q, label_g] * cls_weight
 *                       + sum_{k<8} |box[q,k] - normalize_bbox(gt)[g,k]| * reg_weight, then nan_to_num(100, 100, -100)
 *   (core/bbox/assigners/hungarian_assigner_3d.py:117-130, core/bbox/match_costs/match_cost.py:17-30,
 *    core/bbox/util.py:38-58; FocalLossCost is mmdet's: alpha, gamma = 2, eps = 1e-12.)
 *   cls (NL, B, Q, C) logits; box (NL, B, Q, code >= 8); gt_boxes (sum_gt, gt_dim in 7..9) gravity-centre boxes
 *   (cx, cy, cz, w, l, h, rot[, vx, vy]); gt_labels (sum_gt) int32; gt_start DEVICE (B + 1) int32 prefix offsets;
 *   cost: block (l, b) at element offset Q * (l * sum_gt + gt_start[b]), row-major (Q, G_b).  max_gt = max_b G_b <= 1024.
 *
 * gd4d_head_loss_fwd_bwd - loss (NL, 2) = (loss_cls, loss_bbox) per decoder layer and their gradients
 *   (dense_heads/detr3d_head_pe.py:782-845 with :700-742; mmdet FocalLoss(use_sigmoid, gamma = 2, alpha) and L1Loss):
 *   assigned (NL, B, Q) int32 = index into gt_boxes / gt_labels or -1 for background; code_weights (10);
 *   avg_factors DEVICE 2 floats (cls_avg_factor, num_total_pos: the two all-reduced normalisers, clamped to >= 1 here);
 *   grad_cls / grad_box = d(loss_cls[l]) / d cls[l], d(loss_bbox[l]) / d box[l].
 *   Labels: a gt_labels entry outside [0, C) makes gd4d_match_cost_fwd write NaN into that column's costs (valid costs
 *   are never NaN: nan_to_num has been applied) - the host raises, as the reference's indexing does; the loss kernel
 *   treats assigned >= sum_gt or such a label as background (memory safety only).  The loss terms pass through
 *   torch.nan_to_num semantics (nan -> 0, +/-inf -> +/-FLT_MAX). */
int gd4d_match_cost_fwd(const float* cls, const float* box, const float* gt_boxes, const int32_t* gt_labels,
                        const int32_t* gt_start, float* cost, int NL, int B, int Q, int C, int code, int gt_dim,
                        int sum_gt, int max_gt, float cls_weight, float reg_weight, float alpha, void* stream);
int gd4d_head_loss_fwd_bwd(const float* cls, const float* box, const int32_t* assigned, const float* gt_boxes,
                           const int32_t* gt_labels, const float* code_weights, const float* avg_factors, float* loss,
                           float* grad_cls, float* grad_box, int NL, int B, int Q, int C, int code, int gt_dim,
                           int sum_gt, float alpha, float loss_cls_weight, float loss_bbox_weight, void* stream);

/* gd4d_linear_sum_assignment_batch - HOST function (no GPU work): the linear sum assignment the reference delegates to
 * scipy (hungarian_assigner_3d.py:125-131), for a batch of independent problems solved on `num_threads` host threads.
 * Problem p: cost + cost_offset[p], row-major (rows[p], cols[p]) fp32, finite (run nan_to_num first; the cost kernel
 * does).  Output: col_of_row + out_offset[p], rows[p] int32: the column assigned to each row or -1.  Every column is
 * assigned when cols <= rows (all ground-truth boxes matched), every row when rows <= cols.  The method is the
 * shortest-augmenting-path algorithm scipy uses; the optimum is the same, and so is the matching whenever it is unique. */
int gd4d_linear_sum_assignment_batch(const float* cost, const int64_t* cost_offset, const int32_t* rows,
                                     const int32_t* cols, int num_problems, int32_t* col_of_row,
                                     const int64_t* out_offset, int num_threads);

/* gd4d_hungarian_assign_fwd - the same assignment ON THE DEVICE: HungarianAssigner3D.assign's matching
 * (core/bbox/assigners/hungarian_assigner_3d.py:120-131: nan_to_num - done by gd4d_match_cost_fwd -, linear_sum_assignment, the scatter
 * of :140-144) for every (decoder layer, sample) of a step in one launch, so that the step's loss needs no device -> host copy
 * (dense_heads/detr3d_head_pe.py:822-836 calls the assigner once per layer and sample).  cost: gd4d_match_cost_fwd's buffer (block
 * (l, b) at Q (l sum_gt + gt_start[b]), (Q, G_b) row-major); gt_start (B + 1) on the device; assigned (NL, B, Q) receives the index
 * into the packed ground truth (gt_start[b] + column) of a matched prediction, -1 otherwise; status (NL * B): 0 = solved, 1 = the
 * block holds a NaN (a label outside [0, classes): gd4d_match_cost_fwd's marker - everything stays -1; the host raises when it
 * looks), 2 = infeasible / more boxes than max_gt.  One workgroup per problem runs the shortest-augmenting-path solver of
 * gd4d_linear_sum_assignment_batch operation for operation in fp64, its scan spread over the threads with an arg-min that reproduces
 * the sequential scan's choice (ties included): the matching is IDENTICAL to the host solver's.  workspace:
 * gd4d_hungarian_assign_workspace_bytes (a double copy of every problem, the shorter side as rows); max(Q, max_gt) <= ~4800 (LDS). */
size_t gd4d_hungarian_assign_workspace_bytes(int NL, int B, int Q, int max_gt);
int gd4d_hungarian_assign_fwd(const float* cost, const int32_t* gt_start, int32_t* assigned, int32_t* status, void* workspace,
                              size_t workspace_bytes, int NL, int B, int Q, int sum_gt, int max_gt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GD4D_H_ */
