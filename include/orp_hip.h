/* include/orp_hip.h -- C ABI of liborp_hip.so: the MI355X (gfx950) drop-in for the native operators on the
 * Oriented RepPoints dense-head hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); device-pointer entry points only
 *     enqueue work on it and never synchronise, allocate or free: scratch comes in through `workspace`
 *     (size it with the matching *_workspace_bytes), so every call is hipGraph-capturable;
 *   - return value: 0 on success, a negative ORP_E* code for argument errors, or a positive hipError_t;
 *   - row layouts are those of the reference: oriented boxes / gts [.,8] = x1,y1,...,x4,y4; dets [.,9] = box + score;
 *     point sets [.,18] = nine (x,y) pairs; rotated boxes [.,5] = cx,cy,w,h,theta(radians).
 *
 * Each entry point cites the reference interface (file:line under LiWentomng/OrientedRepPoints) it replaces.
 */
#ifndef ORP_HIP_H_
#define ORP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORP_OK 0
#define ORP_EINVAL (-1)      /* bad argument (NULL pointer, negative size, unsupported option) */
#define ORP_EWORKSPACE (-2)  /* workspace too small */
#define ORP_ETOOBIG (-3)     /* problem exceeds a documented limit */

/* library / build identification: "orp_hip gfx950 <abi>" */
const char* orp_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * Rotated NMS.
 * Replaces rnms_cuda.rnms(Tensor[M,9] f32 cuda, float) -> LongTensor   (mmdet/ops/nms/src/rnms_cuda.cpp:8-13,
 * rnms_kernel.cu:204-265) and the device half of _poly_nms (DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:277-329).
 *   dets      [n,9] f32 (8 corner coords + score), any order
 *   flavor    0 = rnms (devrIoU, rnms_kernel.cu:131-147); 1 = poly_nms (devPolyIoU with the union==0 guard,
 *             poly_nms_kernel.cu:192-212)
 *   presorted 0: sort here by (score desc, index asc) -- stable; 1: rows are already in visiting order
 *   keep_out  [n] int64: ORIGINAL row indices of the kept boxes; ascending when order_out==0 (rnms_cuda's
 *             contract, rnms_kernel.cu:261-264), in visiting (score) order when order_out==1 (_poly_nms' contract)
 *   num_keep  [1] int32 (device): number of valid entries in keep_out
 * Limits: n <= ORP_NMS_MAX_BOXES.  The suppression mask (n * ceil(n/64) u64) lives in the workspace.
 * ------------------------------------------------------------------------------------------------------- */
#define ORP_NMS_MAX_BOXES 131072
size_t orp_rnms_workspace_bytes(int n);
int orp_rnms(const float* dets, int n, float iou_thr, int flavor, int presorted, int order_out,
             int64_t* keep_out, int32_t* num_keep, void* workspace, size_t workspace_bytes, void* stream);

/* Batched form: `nseg` independent segments (image x class) laid back to back in dets; seg_offsets [nseg+1] int32
 * (device).  keep_out holds, segment after segment at offset seg_offsets[s], the kept ORIGINAL (global) row indices
 * ascending; num_keep [nseg].  One launch sequence for all segments (the (image x class) batching of BASELINE.md
 * section 3).  max_seg = host-known upper bound on a segment's size: the actual sizes are read from seg_offsets ON THE
 * DEVICE, so a caller that only knows a capacity (sync-free post-processing, hipGraph replay) passes the capacity here
 * and a device-computed [0, count] pair as seg_offsets; rows past the count must carry a score of -inf. */
size_t orp_rnms_batched_workspace_bytes(int n_total, int nseg, int max_seg);
int orp_rnms_batched(const float* dets, int n_total, const int32_t* seg_offsets, int nseg, int max_seg,
                     float iou_thr, int flavor, int64_t* keep_out, int32_t* num_keep,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Host-pointer API of DOTA_devkit/poly_nms_gpu (exact signatures of poly_nms.hpp:9-10 and poly_overlaps.hpp:1):
 * host buffers in, host buffers out, own device allocation and a blocking copy, errors printed to stderr. */
void _poly_nms(int* keep_out_host, int* num_out_host, const float* polys_host, int polys_num, int polys_dim,
               float nms_overlap_thresh, int device_id);
void _overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k,
               int device_id);

/* ---------------------------------------------------------------------------------------------------------
 * Pairwise IoU matrices.
 * orp_quad_iou_matrix: fp32 quad-quad IoU of every (a_i, b_j) with the arithmetic of devrIoU (guard=0) or
 *   devPolyIoU (guard=1); rows of a/b are `stride` floats apart (8 or 9).  out [n,k] f32.
 * orp_poly_overlaps: device half of _overlaps (poly_overlaps_kernel.cu:330-353): boxes [n,5], query [k,5] -> [n,k].
 * ------------------------------------------------------------------------------------------------------- */
int orp_quad_iou_matrix(const float* a, int n, const float* b, int k, int stride, int guard, float* out, void* stream);
int orp_poly_overlaps(const float* boxes, int n, const float* query, int k, float* out, void* stream);
/* orp_box_iou_rotated: box_iou_rotated (mmdet/ops/box_iou_rotated/src/box_iou_rotated_cuda.cu:14-94,
 *   box_iou_rotated_utils.h:314-341): boxes1 [n,5], boxes2 [k,5] (cx,cy,w,h,theta radians) -> [n,k]. */
int orp_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int k, float* out, void* stream);
/* CPU branch of the reference's dispatcher (box_iou_rotated.h:20-33 -> box_iou_rotated_cpu.cpp): HOST pointers,
 * the same per-pair arithmetic compiled for the host (the reference API accepts CPU tensors here). */
int orp_box_iou_rotated_host(const float* boxes1, int n, const float* boxes2, int k, float* out);

/* ---------------------------------------------------------------------------------------------------------
 * minaerarect: convex hull of 9 points -> minimum-area enclosing rectangle -> 4 corners.
 * Replaces minareabbox(Tensor[M,18]) -> Tensor[M,8]  (mmdet/ops/minarearect/src/minarearect_cuda.cpp:5-9,
 * minarearect_kernel.cu:455-505).  Optional fused decode of get_bboxes_single
 * (orientedreppoints_head.py:746-749): out = rect * scale + (cx,cy) per row when `centers` != NULL
 * (centers [m,2], scales [m]).
 * ------------------------------------------------------------------------------------------------------- */
int orp_minarearect(const float* pts, int m, float* out, void* stream);
int orp_minarearect_decode(const float* pts, int m, const float* centers, const float* scales, float* out, void* stream);
/* Self-check of the single-precision elementary functions the kernels evaluate with the HOST C library's algorithms (csrc/orp_libm.hpp),
 * so that minareabbox's `cos(unique_angles[i])` (minarearect_kernel.cu:117-120), RotBox2Poly's `cos(dbox[4])` (poly_overlaps_kernel.cu:
 * 281-282) and the focal loss's expf / logf / powf (sigmoid_focal_loss_cuda.cu:36-57) give the bits the reference compiled for the host
 * gives: out[i] = cosf (which 0) | sinf (1) | expf (2) | logf (3) of x[i], or powf(x[i], y[i]) (4; y may be NULL otherwise).  Tests only. */
int orp_libm_eval(const float* x, const float* y, long n, int which, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * convex_iou: IoU(hull(9 points), gt quad) for all pairs, fp64 internals.
 * Replaces convex_iou_cuda(ex[N,18], gt[K,8]) -> flat [N*K]   (mmdet/ops/iou/src/convex_iou_kernel.cu:298-360).
 * out [n,k] f32 row-major (point set major), exactly the reference's flat layout.
 * ------------------------------------------------------------------------------------------------------- */
int orp_convex_iou(const float* pts, int n, const float* gts, int k, float* out, void* stream);
/* convex_giou: aligned pairs, GIoU(hull(9 points), gt) and its gradient w.r.t. the 18 coordinates;
 * out19 [n,19] = 18 grads + giou, the row layout of convex_giou_kernel (convex_giou_kernel.cu:806-868). */
int orp_convex_giou(const float* pts, const float* gts, int n, float* out19, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * pointsJf: ray-casting point-in-quad flags for all M x K pairs (mmdet/ops/point_justify/src/
 * points_justify_kernel.cu:25-119) and the aligned form the SpatialBorderLoss actually needs (row i of 9 points
 * against quad i; replaces 9 x (M x M) + torch.diag, spatial_border_loss.py:24-67).
 * ------------------------------------------------------------------------------------------------------- */
int orp_points_justify(const float* points, int m, const float* polygons, int k, float* out, void* stream);
int orp_points_in_quad_aligned(const float* pts18, const float* quads, int m, float* out9, void* stream);

/* ChamferDistance2D forward (mmdet/ops/chamfer_2d/src/chamfer_2d.cu:12-141): per batch nearest-neighbour
 * squared distance + index, both directions.  xyz1 [b,n,2], xyz2 [b,m,2]; dist1/idx1 [b,n], dist2/idx2 [b,m]. */
int orp_chamfer2d_forward(const float* xyz1, const float* xyz2, int b, int n, int m,
                          float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, void* stream);
int orp_chamfer2d_backward(const float* xyz1, const float* xyz2, int b, int n, int m,
                           const float* grad_dist1, const float* grad_dist2, const int32_t* idx1, const int32_t* idx2,
                           float* grad_xyz1, float* grad_xyz2, void* stream);

/* sigmoid focal loss forward/backward (mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:23-167).
 * logits [num,classes] f32, targets [num] int64 (0 = background, 1..classes), losses / d_logits [num,classes]. */
int orp_sigmoid_focal_loss_forward(const float* logits, const int64_t* targets, int num, int classes,
                                   float gamma, float alpha, float* losses, void* stream);
int orp_sigmoid_focal_loss_backward(const float* logits, const int64_t* targets, const float* d_losses, int num,
                                    int classes, float gamma, float alpha, float* d_logits, void* stream);
/* scalar_t = double of the reference's dispatch (sigmoid_focal_loss_cuda.cu:121,160): double tensors, the expressions' single-precision
 * expf / logf / powf calls kept as the template instantiates them. */
int orp_sigmoid_focal_loss_forward_f64(const double* logits, const int64_t* targets, int num, int classes,
                                       float gamma, float alpha, double* losses, void* stream);
int orp_sigmoid_focal_loss_backward_f64(const double* logits, const int64_t* targets, const double* d_losses, int num,
                                        int classes, float gamma, float alpha, double* d_logits, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Deformable convolution forward (DCNv1 / DCNv2).
 * Replaces deform_conv_forward_cuda / modulated_deform_conv_cuda_forward (mmdet/ops/dcn/src/deform_conv_cuda.cpp:
 * 152-260, 490-590) and their im2col kernels (deform_conv_cuda_kernel.cu:190-277, 570-700).
 *
 * orp_dcn_forward_multi: the MFMA implicit-GEMM path.  ALL FPN levels of one DeformConv layer in one launch (the
 *   head applies the same layer to 5 levels, orientedreppoints_head.py:148-174): levels_host[i] = {input, offset,
 *   output, height, width}.  input [B,Cin,H,W] (in_layout 0 = NCHW, converted through the workspace; 1 = NHWC),
 *   offset [B,2*kh*kw,Ho,Wo] NCHW, output [B,Cout,Ho,Wo] (out_layout 0 = NCHW, 1 = NHWC).  weight_packed comes from
 *   orp_dcn_pack_weight ([Cout,Cin,kh,kw] -> [kh*kw,Cin,Cout]).  Requires orp_dcn_fast_path_ok(...) == 1
 *   (groups = deformable_groups = 1, kh*kw <= 9, Cin % 32 == 0, Cout % 64 == 0); fp32, fp32-exact MFMA.
 * orp_dcn_forward_direct: every other configuration (groups, deformable groups; mask != NULL = DCNv2 modulation,
 *   optional bias), NCHW in/out, original [Cout,Cin/groups,kh,kw] weight.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct { const float* input; const float* offset; float* output; int height; int width; } orp_dcn_level;
int orp_dcn_fast_path_ok(int c_in, int c_out, int kh, int kw, int groups, int deformable_groups);
/* `packed` holds orp_dcn_packed_weight_floats() floats: [kh*kw][Cin][Cout] followed by [kh*kw][Cin/4][Cout][4], followed
 * (Cin % 64 == 0) by the weights split exactly into three bf16 planes [3][kh*kw][Cin/16][2][Cout][8]. */
size_t orp_dcn_packed_weight_floats(int c_out, int c_in, int kh, int kw);
int orp_dcn_pack_weight(const float* weight, int c_out, int c_in, int kh, int kw, float* packed, void* stream);
/* The contraction of the fp32 forward entry points below (same tensors, same fp32 accumulation) is issued on the 16-bit matrix
 * pipe (csrc/orp_dcn_split.hip): mode 3 (the default since round 5) = every fp32 operand as TWO fp16 pieces after an exact
 * power-of-two range scaling, three partial products; 6 / 9 = every operand split EXACTLY into three bf16 pieces, six partial
 * products (without the three below 2^-24 of the product) / all nine (no representation error); 0 = off (exact fp32 MFMA);
 * -1 = back to the environment's choice (ORP_DCN_SPLIT = 0 | 3 | 6 | 9, 1 = 6; unset = 3).  Process-wide; takes effect for
 * Cin % 64 == 0, Cout % 64 == 0; a launch without a known input range (DCNv2 modulation) runs mode 3 as mode 6.  Every mode's
 * error against the fp64-accumulated oracle is at or below the exact-fp32 kernel's own (tests/test_gpu_dcn_split.py); the
 * reference has one arithmetic: deform_conv_cuda.cpp:222-237. */
int orp_dcn_set_split_mode(int mode);
int orp_dcn_get_split_mode(void);
/* (includes 25 MB of scratch for launches of more tiles than CUs: those split every layer's (tile, tap) steps evenly over
 * the workgroups, and the accumulators of a tile cut between two workgroups pass through it -- fixed order, reproducible) */
size_t orp_dcn_forward_workspace_bytes(const orp_dcn_level* levels_host, int nlevels, int batch, int c_in, int in_layout);
int orp_dcn_forward_multi(const orp_dcn_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                          const float* weight_packed, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                          int dil_h, int dil_w, int in_layout, int out_layout, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Extended form: DCNv2 on the same MFMA path (masks_host[i] = modulation [B,kh*kw,Ho,Wo] of level i, or masks_host NULL
 * for DCNv1), optional bias [Cout] (ModulatedDeformConv's bias, deform_conv.py:411-418) and an optional fused ReLU in
 * the epilogue (the head applies self.relu right after both DeformConvs, orientedreppoints_head.py:166-170). */
int orp_dcn_forward_multi_ex(const orp_dcn_level* levels_host, const float* const* masks_host, int nlevels, int batch,
                             int c_in, int c_out, const float* weight_packed, const float* bias, int relu, int kh, int kw,
                             int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout,
                             int out_layout, void* workspace, size_t workspace_bytes, void* stream);
/* Two DeformConv layers that share their offsets (and masks) in ONE launch: the head's reppoints_cls_conv and
 * reppoints_pts_refine_conv both take `dcn_offset` (orientedreppoints_head.py:164-170).  levels_a[i] / levels_b[i] carry
 * each layer's input and output of level i (levels_b[i].offset is ignored); the bilinear coefficient table of a tile
 * is built once for both.  Requires c_in % 256 == 0 (else ORP_EINVAL: call orp_dcn_forward_multi_ex twice). */
int orp_dcn_forward_pair(const orp_dcn_level* levels_a, const orp_dcn_level* levels_b, const float* const* masks_host,
                         int nlevels, int batch, int c_in, int c_out, const float* weight_a_packed,
                         const float* weight_b_packed, const float* bias_a, const float* bias_b, int relu, int kh, int kw,
                         int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout,
                         int out_layout, void* workspace, size_t workspace_bytes, void* stream);
/* The head's complete refinement stage in one launch: both DeformConvs (256 -> 256, same offsets), ReLU, and the 1x1
 * output convolution behind each (reppoints_cls_out: k_a = 15 channels; reppoints_pts_refine_out: k_b = 18, with
 * `+ pts_out_init` as residual_b), orientedreppoints_head.py:164-170.  The 256-channel DeformConv outputs are never
 * written (levels_a[i].output / levels_b[i].output are ignored).  heads->weight_*_packed: orp_dcn_head_packed_floats()
 * floats from orp_dcn_pack_head_weight ([k,256] -> [256][20]); heads->levels[i]: NCHW outputs [B,k_a,Ho,Wo] / [B,k_b,Ho,Wo]
 * and the optional residual of the second head.  k <= 20.  The partial sums of the eight waves are added in a fixed order. */
/* the same with an upper bound of max |x| of the channels-last inputs from their producer (see orp_conv_split_multi):
 * layers a / b read amax_in[0] / amax_in[amax_stride]; used in the two-fp16-pieces mode with in_layout 1 only */
int orp_dcn_forward_pair_amax(const orp_dcn_level* levels_a, const orp_dcn_level* levels_b, const float* const* masks_host,
                              int nlevels, int batch, int c_in, int c_out, const float* weight_a_packed,
                              const float* weight_b_packed, const float* bias_a, const float* bias_b, int relu, int kh, int kw,
                              int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout,
                              int out_layout, void* workspace, size_t workspace_bytes, const uint32_t* amax_in, int amax_stride,
                              void* stream);
typedef struct { float* output_a; float* output_b; const float* residual_b; } orp_dcn_head_level;
typedef struct { const float* weight_a_packed; const float* bias_a; int k_a; const float* weight_b_packed; const float* bias_b;
                 int k_b; const orp_dcn_head_level* levels; } orp_dcn_heads;
size_t orp_dcn_head_packed_floats(void);
int orp_dcn_pack_head_weight(const float* weight, int k, float* packed, void* stream);
int orp_dcn_forward_pair_heads(const orp_dcn_level* levels_a, const orp_dcn_level* levels_b, int nlevels, int batch,
                               const float* weight_a_packed, const float* weight_b_packed, const orp_dcn_heads* heads,
                               int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                               int in_layout, void* workspace, size_t workspace_bytes, void* stream);
/* fp16 / bf16 DeformConv forward on v_mfma_f32_32x32x16_{f16,bf16} (the reference dispatches its DCN kernels over float
 * AND half: deform_conv_cuda_kernel.cu:259,353,451,781,813 AT_DISPATCH_FLOATING_TYPES_AND_HALF; BASELINE configs[4]).
 * dtype: 1 = fp16, 2 = bf16 -- inputs, offsets, masks, bias, packed weights and outputs are all of that type; the bilinear
 * combine and the accumulation are fp32.  Requires c_in % 256 == 0, c_out % 64 == 0, groups = deformable_groups = 1
 * (orp_dcn_half_path_ok); other configurations: convert to fp32 and use the entries above.
 * packed weights: c_out * c_in * kh * kw elements, layout [tap][c_in/16][2][c_out][8]. */
typedef struct { const void* input; const void* offset; void* output; int height; int width; } orp_dcn_level_h;
int orp_dcn_half_path_ok(int c_in, int c_out, int kh, int kw, int groups, int deformable_groups);
int orp_dcn_pack_weight_h(const void* weight, int c_out, int c_in, int kh, int kw, void* packed, int dtype, void* stream);
size_t orp_dcn_forward_h_workspace_bytes(const orp_dcn_level_h* levels_host, int nlevels, int batch, int c_in, int in_layout);
int orp_dcn_forward_multi_h(const orp_dcn_level_h* levels_host, const void* const* masks_host, int nlevels, int batch, int c_in,
                            int c_out, const void* weight_packed, const void* bias, int relu, int kh, int kw, int stride_h,
                            int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout, int out_layout, int dtype,
                            void* workspace, size_t workspace_bytes, void* stream);
int orp_dcn_forward_direct(const float* input, const float* offset, const float* mask, const float* weight,
                           const float* bias, float* output, int batch, int c_in, int height, int width, int c_out,
                           int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                           int groups, int deformable_groups, void* stream);

/* Deformable convolution (DCNv1) backward as two MFMA implicit GEMMs, no column buffer in HBM
 * (deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda, deform_conv_cuda.cpp:262-488), all levels of
 * one layer in one call.  Requires orp_dcn_backward_mfma_ok (c_in = c_out = 256, groups = deformable_groups = 1,
 * kh*kw <= 9); every other configuration: the column entries below.  All tensors NCHW fp32:
 *   input [B,256,H,W], offset [B,2*kh*kw,Ho,Wo], grad_output [B,256,Ho,Wo] ->
 *   grad_input [B,256,H,W], grad_offset [B,2*kh*kw,Ho,Wo] (both OVERWRITTEN; written when need_input_grads != 0),
 *   grad_weight [256,256,kh,kw] (OVERWRITTEN; NULL = not wanted) = sum over levels and images, added in a fixed order.
 * need_input_grads: 0 = grad_weight only; ORP_DCN_BWD_INPUT (1) = grad_input / grad_offset without atomics: every
 *   8 x 8-pixel region of grad_input is accumulated by ONE workgroup in LDS in a fixed order (bitwise reproducible, each
 *   byte of grad_input written once); ORP_DCN_BWD_INPUT | ORP_DCN_BWD_SPARSE (3) = the caller expects grad_output to be
 *   zero almost everywhere (a detection head's regression branch: gradient at the positive points only): the few live
 *   rows are scattered with fp32 atomics instead, which skips the fixed cost of the region pass (same values to 1e-4;
 *   summation order not fixed, as in the reference's deformable_col2im).
 * weight is the layer's [256,256,kh,kw] tensor.  workspace: orp_dcn_backward_workspace_bytes() bytes. */
#define ORP_DCN_BWD_INPUT 1
#define ORP_DCN_BWD_SPARSE 2
typedef struct { const void* input; const float* offset; const void* grad_output; void* grad_input; float* grad_offset;
                 int height; int width; } orp_dcn_bwd_level;   /* input / grad_output / grad_input: fp32, or io_dtype of the _ex entry */
int orp_dcn_backward_mfma_ok(int c_in, int c_out, int kh, int kw, int groups, int deformable_groups);
size_t orp_dcn_backward_workspace_bytes(const orp_dcn_bwd_level* levels_host, int nlevels, int batch, int kh, int kw,
                                        int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w);
int orp_dcn_backward_multi(const orp_dcn_bwd_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                           const float* weight, float* grad_weight, int need_input_grads, int kh, int kw, int stride_h,
                           int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The same with DCNv2 modulation and half-precision tensors (modulated_deform_conv_cuda_backward,
 * deform_conv_cuda.cpp:592-685; AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:353,451,781,813):
 *   masks_host[i] [B,kh*kw,Ho,Wo] fp32 (NULL array = DCNv1): the sample of (position, tap) is scaled by its mask value in
 *     grad_weight's columns, grad_input's scatter and grad_offset; grad_masks_host[i] receives d loss / d mask
 *     (= G . sampled value, summed over the channels in a fixed order);
 *   io_dtype 0 / 1 / 2 = input, grad_output and grad_input are fp32 / fp16 / bf16 (converted inside the layout passes the
 *     call runs anyway); offsets, masks, weight and the remaining gradients stay fp32; fp32 accumulation -- grad_weight on fp32
 *     MFMAs, the grad_input / grad_offset contraction on two fp16 pieces per operand like the forward's mode 3 (environment
 *     ORP_DCN_BWD_SPLIT=0: fp32 MFMAs). */
int orp_dcn_backward_multi_ex(const orp_dcn_bwd_level* levels_host, const float* const* masks_host,
                              float* const* grad_masks_host, int io_dtype, int nlevels, int batch, int c_in, int c_out,
                              const float* weight, float* grad_weight, int need_input_grads, int kh, int kw, int stride_h,
                              int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Deformable convolution backward, column formulation (deform_conv_cuda.cpp:262-488, kernels
 * deform_conv_cuda_kernel.cu:190-465 and the modulated twins :570-867).  NCHW fp32.
 * orp_dcn_im2col: columns [Cin*kh*kw, B*Ho*Wo] (x mask when mask != NULL) -- feeds grad_W = grad_out . columns^T.
 * orp_dcn_col2im: from grad_columns = W^T . grad_out: grad_input (ACCUMULATED: caller zeroes), grad_offset and, for
 *   DCNv2 (mask/grad_mask both non-NULL), grad_mask -- one kernel instead of col2im + col2im_coord. */
int orp_dcn_im2col(const float* input, const float* offset, const float* mask, int batch, int c_in, int height,
                   int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                   int deformable_groups, float* columns, void* stream);
int orp_dcn_col2im(const float* grad_columns, const float* input, const float* offset, const float* mask, int batch,
                   int c_in, int height, int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                   int dil_h, int dil_w, int deformable_groups, float* grad_input, float* grad_offset, float* grad_mask,
                   void* stream);
/* The same two kernels for DOUBLE tensors -- the `double` branch of the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF
 * (deform_conv_cuda_kernel.cu:259,353,451,720,777,838): DeformConvFunction / ModulatedDeformConvFunction with float64 tensors run the
 * reference's own column formulation (deform_conv_cuda.cpp:152-488) in double -- these sampling kernels around the library's double
 * GEMMs -- instead of being narrowed to fp32 (rounds 1-5). */
int orp_dcn_im2col_f64(const double* input, const double* offset, const double* mask, int batch, int c_in, int height,
                       int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                       int deformable_groups, double* columns, void* stream);
int orp_dcn_col2im_f64(const double* grad_columns, const double* input, const double* offset, const double* mask, int batch,
                       int c_in, int height, int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                       int dil_h, int dil_w, int deformable_groups, double* grad_input, double* grad_offset, double* grad_mask,
                       void* stream);
/* channel-parallel variant for deformable_groups = 1: grad_columns_t [B*Ho*Wo, kh*kw, Cin] (position-major, from
 * grad_out(NHWC) . W[Cout, kh*kw*Cin]), input / grad_input NHWC (grad_input zeroed by the caller), offsets NCHW. */
int orp_dcn_col2im_nhwc(const float* grad_columns_t, const float* input_nhwc, const float* offset, int batch, int c_in,
                        int height, int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                        int dil_h, int dil_w, float* grad_input_nhwc, float* grad_offset, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Assignment side of the APAA training path (what the reference does with Python loops over ground truths).
 * orp_point_assign: PointAssigner.assign (mmdet/core/bbox/assigners/point_assigner.py:22-145).  points [n,3] =
 *   (x, y, stride), gts [k,8]; gt_inds [n] int64 (0 = background, i+1 = gt i).  Each gt takes its `pos_num` nearest
 *   points (normalised centre distance) on its pyramid level; a point wanted by several gts goes to the closest,
 *   ties to the earlier gt.
 * orp_max_iou_assign: MaxIoUAssigner.assign_wrt_overlaps (max_iou_assigner.py:88-152) on the POINT-MAJOR overlap
 *   matrix [n,k] produced by orp_convex_iou.  neg range [neg_iou_lo, neg_iou_hi) -> 0, >= pos_iou_thr -> argmax+1,
 *   then every point whose overlap equals a gt's maximum (>= min_pos_iou) -> that gt (later gt overwrites);
 *   gt_inds [n] int64 in {-1, 0, 1..k}, max_overlaps [n] (may be NULL).
 * orp_apaa_feature_dissimilarity: get_adaptive_points_feature + feature_cosine_similarity
 *   (orientedreppoints_head.py:495-520, 576-600) for the POSITIVES only: bilinear samples (grid_sample, zeros padding,
 *   align_corners=False) of the level's [B,C,H,W] feature map at the 9 refined points of each positive, then
 *   max_k (1 - cos(f_k, mean_k f_k)) with norms clamped at 1e-2.  feats_host etc. are HOST arrays of per-level
 *   device pointers / sizes; pts18 [p,18] image-space (x,y); img_index / level_index [p] int32; out [p].
 * orp_apaa_select: point_samples_selection (orientedreppoints_head.py:602-671): per gt, per level the
 *   per_level_topk smallest-quality positives, merged, sorted ascending, first ceil(top_ratio * n) kept
 *   (all kept when n < 2).  quality [p], pos_gt_inds [p] int64 (1-based), pos_level [p] int32; keep [p] uint8.
 * ------------------------------------------------------------------------------------------------------- */
size_t orp_point_assign_workspace_bytes(int n);
int orp_point_assign(const float* points, int n, const float* gts, int k, float scale, int pos_num, int64_t* gt_inds,
                     void* workspace, size_t workspace_bytes, void* stream);
size_t orp_max_iou_assign_workspace_bytes(int k);
int orp_max_iou_assign(const float* overlaps_nk, int n, int k, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
                       float min_pos_iou, int gt_max_assign_all, int64_t* gt_inds, float* max_overlaps,
                       void* workspace, size_t workspace_bytes, void* stream);
int orp_apaa_feature_dissimilarity(const float* const* feats_host, const int* heights_host, const int* widths_host,
                                   const float* strides_host, int num_levels, int channels, const float* pts18,
                                   const int32_t* img_index, const int32_t* level_index, int p, float* out,
                                   void* stream);
int orp_apaa_select(const float* quality, const int64_t* pos_gt_inds, const int32_t* pos_level, int p, int num_gt,
                    int num_level, int per_level_topk, double top_ratio, uint8_t* keep, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Batched glue of the training path (csrc/orp_train.hip): what the reference does with per-image / per-level Python
 * loops of small tensor operations between its compiled operators.  Level tensors are described by HOST arrays of
 * orp_level_desc { data [B,C,H,W] fp32 on the device, grad (same shape; only for the backward entry), H, W, point stride };
 * a location is addressed as  b * N + i  with i running over the levels in order (fine -> coarse), row-major inside a
 * level -- the order of PointGenerator.grid_points / levels_to_images.
 * orp_pointset_target: init_ / refine_pointset_target_single + unmap + images_to_levels
 *   (mmdet/core/bbox/pointset_target.py:61-121,171-230) for all images in ONE launch.  gt_inds [batch*n] int64 from the
 *   assigner (-1 ignore, 0 negative, i+1 = gt i of the image), valid [batch*n] uint8 or NULL (invalid locations get the
 *   reference's unmap fill 0), gt_boxes [K_total,8] / gt_labels [K_total] int64 (NULL: label 1) of all images
 *   concatenated, gt_offset [batch+1] int32, proposals [batch*n, dim] or NULL.  Outputs at full-N positions:
 *   labels int64, label_weights (pos_weight <= 0 -> 1), rbbox_gt [.,8], pos_proposals [., dim] (NULL = not wanted),
 *   proposal_weights, gt_inds_out int64, counts [batch,2] int32 = positives / negatives per image (NULL = not wanted).
 * orp_points_from_offsets: every location's 9-point set from the [B,18,H,W] offset maps -> out [batch, N, 18].
 *   mode 0 = offset_to_pts (orientedreppoints_head.py:204-222): (x, y) pairs, x = pred[2k+1] * stride + cx;
 *   mode 1 = the refine-stage proposals of loss() (:378-381): element j = centre[j & 1] + pred[j] * stride (no swap).
 * orp_gather_levels: out [p, C] = the C channels at location index[p] (mode 0), or the image-space point set of
 *   offset_to_pts at that location (mode 1, C = 18).  orp_gather_levels_backward: the level gradients (zero-filled
 *   here) receive grad_out at the selected locations (distinct locations: plain stores, deterministic).
 * orp_outline_samples: sampling_points (:250-292): corners [p,8] -> out [p, 4*n, 2], n points per edge at the
 *   caller's ratios [n] (device; the reference's torch.linspace(0, 1, n)), edges 1->2->3->4->1.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct { const float* data; float* grad; int height; int width; float stride; } orp_level_desc;
int orp_pointset_target(const int64_t* gt_inds, const uint8_t* valid, int batch, int n, const float* gt_boxes,
                        const int64_t* gt_labels, const int32_t* gt_offset, const float* proposals, int dim,
                        float pos_weight, int64_t* labels, float* label_weights, float* rbbox_gt, float* pos_proposals,
                        float* proposal_weights, int64_t* gt_inds_out, int32_t* counts, void* stream);
int orp_points_from_offsets(const orp_level_desc* levels_host, int nlevels, int batch, int channels, int mode, float* out,
                            void* stream);
int orp_gather_levels(const orp_level_desc* levels_host, int nlevels, int batch, int channels, const int64_t* index, int p,
                      int mode, float* out, void* stream);
int orp_gather_levels_backward(const orp_level_desc* levels_host, int nlevels, int batch, int channels, const int64_t* index,
                               int p, int mode, const float* grad_out, void* stream);
int orp_outline_samples(const float* corners, int p, int n, const float* ratios, float* out, void* stream);

/* Segment losses of the head's loss(): rows = the positive point sets of a stage, seg[row] = its segment (FPN level of the
 * init stage, or 0), nseg <= 16.  Replaces the tensor-op composition around GIoULoss (iou_loss.py:69-129, reduction mean per
 * level, orientedreppoints_head.py:294-318) and SpatialBorderLoss (spatial_border_loss.py:8-92) -- ~25 framework launches each.
 * orp_border_rows: over the points of a row that lie OUTSIDE its gt quad (pointsJf == 0) and only for weight > 0:
 *   row_sum = sum 0.2 |p - centre|, row_cnt = their number, gdir [p,18] = d(0.2 |p - centre|)/dp (0 elsewhere).
 * orp_giou_rows: contrib = (1 - giou) * weight; gsave [p,18] = -(grad, or 1e-6 in rows with any component > 1)
 *   * weight / max(denom[seg], 1) * loss_weight -- the gradient the reference returns from backward().
 * orp_segment_finish (one workgroup, fixed summation order): mode 0: loss[s] = sum_s(row_val) / max(denom[s], 1) *
 *   loss_weight; mode 1: loss[s] = loss_weight * (sum_s(row_val) / max(sum_s(row_cnt), 1)) / (denom[s] + 1e-6) and
 *   scale[s] = loss_weight / max(sum_s(row_cnt), 1) / (denom[s] + 1e-6). */
int orp_border_rows(const float* pts18, const float* gt8, const float* weight, int p, float* row_sum, float* row_cnt,
                    float* gdir, void* stream);
int orp_giou_rows(const float* gious, const float* grad18, const float* weight, const int64_t* seg, const float* denom, int p,
                  float loss_weight, float* contrib, float* gsave, void* stream);
int orp_segment_finish(const float* row_val, const float* row_cnt, const int64_t* seg, int p, int nseg, const float* denom,
                       float loss_weight, int mode, float* loss, float* scale, void* stream);

/* fp64 greedy polygon NMS -- the merge step of the DOTA evaluation workflow (DOTA_devkit/ResultMerge.py:18-41
 * py_cpu_nms_poly over polyiou.cpp:108-128 iou_poly), SURVEY 8f rank 2.  dets_sorted [n,9] DOUBLE on device, already
 * in visiting order (the caller applies numpy's `scores.argsort()[::-1]` exactly as the reference does); a box
 * survives a kept predecessor only if iou <= thr (NaN suppresses, ResultMerge.py:38).  keep_out [n] int64 receives the
 * kept POSITIONS in visiting order, num_keep[0] their count. */
size_t orp_poly_nms_f64_workspace_bytes(int n);
int orp_poly_nms_f64(const double* dets_sorted, int n, double iou_thr, int64_t* keep_out, int32_t* num_keep,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Detection -> ground-truth matching of the DOTA Task1 evaluation (DOTA_devkit/dota_evaluation_task1.py:160-206, the
 * per-detection python loop of voc_eval): for detection d of image det_image[d], over the ground truths
 * gts[gt_offsets[img] .. gt_offsets[img + 1]) that pass the fp64 horizontal-box pre-filter (`overlaps > 0`, "+ 1." pixel
 * convention), ovmax[d] = np.max and jmax[d] = np.argmax (image-local index) of polyiou.iou_poly(GT, detection) in fp64;
 * (-inf, -1) when none passes; a NaN IoU wins as in numpy.  dets [nd,8], gts [ng,8] fp64 device; int32 device arrays. */
int orp_voc_best_match_f64(const double* dets, const int32_t* det_image, int num_dets, const double* gts,
                           const int32_t* gt_offsets, int num_images, double* ovmax, int32_t* jmax, void* stream);

/* Soft rotated NMS on the HOST -- replaces rnms_cpu.soft_rnms (mmdet/ops/nms/src/rnms_cpu.cpp:165-333), CPU-only in
 * the reference too.  dets_host [m,9] fp32 (8 corners + score); method 0 = hard, 1 = linear, 2 = gaussian;
 * out_host [m,10] receives the surviving rows (8 corners, rescored score, original index as float) in selection
 * order, *num_out their count. */
int orp_soft_rnms_host(const float* dets_host, int m, float iou_thr, int method, float sigma, float min_score,
                       float* out_host, int* num_out);

/* ---------------------------------------------------------------------------------------------------------
 * Fused normalisation + activation passes (inference), SURVEY 8f rank 3 "head towers fused for MI355X".
 * orp_groupnorm_act_multi: GroupNorm (+ ReLU) of the dense-head ConvModules (mmdet/ops/conv_module.py:130-140 as used
 *   by orientedreppoints_head.py:91-113), ALL FPN levels of one layer in one launch pair (statistics, then one
 *   read-modify-write pass).  levels_host[i] = {input, output, height, width}, tensors NCHW [B,C,H,W] fp32; output may
 *   alias input.  workspace: orp_groupnorm_workspace_bytes().
 * orp_affine_act: y = relu?(x*scale[c] + shift[c] (+ residual)) -- eval-mode BatchNorm folded to a per-channel affine
 *   and fused with the bottleneck's residual add + ReLU (mmdet/models/backbones/resnet.py:133-170); residual may be
 *   NULL, y may alias x.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct { const float* input; float* output; int height; int width; } orp_norm_level;
size_t orp_groupnorm_workspace_bytes(const orp_norm_level* levels_host, int nlevels, int batch, int channels, int groups);
int orp_groupnorm_act_multi(const orp_norm_level* levels_host, int nlevels, int batch, int channels, int groups,
                            const float* gamma, const float* beta, float eps, int relu, void* workspace,
                            size_t workspace_bytes, void* stream);
/* per-tensor affine parameters (up to 16 tensors: both towers' five levels in one launch pair) */
int orp_groupnorm_act_multi_ex(const orp_norm_level* levels, const float* const* gammas_host,
                               const float* const* betas_host, int nlevels, int batch, int channels, int groups,
                               float eps, int relu, void* workspace, size_t workspace_bytes, void* stream);
/* the same normalisation written TRANSPOSED: nhwc_out_host[i] = [B, H*W, C] (channels-last), through an LDS tile; where
 * levels[i].output != NULL the NCHW result is written as well (may alias the input), NULL = channels-last only.  The head's
 * DeformConv (orp_dcn_forward_pair, in_layout 1) reads the towers' last layer channels-last: this replaces a separate
 * transposition launch.  channels % 32 == 0, 32 % (channels / groups) == 0.  Same values as orp_groupnorm_act_multi_ex. */
int orp_groupnorm_act_multi_nhwc(const orp_norm_level* levels, const float* const* gammas_host,
                                 const float* const* betas_host, float* const* nhwc_out_host, int nlevels, int batch,
                                 int channels, int groups, float eps, int relu, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* GroupNorm(+ReLU) of channels-last tensors, [B, H, W, C] in and out (in place allowed): what sits between two
 * orp_conv_split_multi launches.  Three launches for all tensors (up to 16): per-chunk (mean, M2) partials of every group,
 * their fixed-order merge into (mean, rstd) per (tensor, image, group), one elementwise pass.  1024 % channels == 0,
 * (channels / groups) % 4 == 0.  workspace: orp_groupnorm_cl_workspace_bytes. */
size_t orp_groupnorm_cl_workspace_bytes(const orp_norm_level* levels_host, int nlevels, int batch, int channels, int groups);
int orp_groupnorm_act_multi_cl(const orp_norm_level* levels, const float* const* gammas_host, const float* const* betas_host,
                               int nlevels, int batch, int channels, int groups, float eps, int relu, void* workspace,
                               size_t workspace_bytes, void* stream);
/* ... leaving an upper bound of max |y| over the tensors of every slot in amax_out[slot] (float bits; zeroed here): from the
 * statistics pass, (max |x| + |mean|) rstd max |gamma| + max |beta| per group -- the range the fp16-pieces convolution that
 * reads y scales by (orp_conv_split_multi amax_in, orp_dcn_forward_pair_amax) without a pass of its own */
int orp_groupnorm_act_multi_cl_amax(const orp_norm_level* levels, const float* const* gammas_host, const float* const* betas_host,
                                    int nlevels, int batch, int channels, int groups, float eps, int relu, const int* slots_host,
                                    uint32_t* amax_out, int nslots, void* workspace, size_t workspace_bytes, void* stream);
/* training: the same launch pair, additionally storing (mean, rstd) of every (image, group) in
 * stats [sum over tensors of batch * groups][2] (tensor i's rows follow tensor i-1's), and the backward:
 *   grad_inputs[i] = d loss / d x_i given grad_outputs[i] = d loss / d y_i (dy masked where y <= 0 when relu),
 *   dgammas / dbetas [C] of every DISTINCT parameter set (tensors sharing a gamma pointer are summed; pass the same
 *   dgamma / dbeta pointers for them), all sums in a fixed order.  levels_host[i].input = x_i (the forward's input),
 *   .output = y_i (the forward's output: its sign is the ReLU mask).  workspace: orp_groupnorm_backward_workspace_bytes(). */
int orp_groupnorm_act_multi_train(const orp_norm_level* levels, const float* const* gammas_host,
                                  const float* const* betas_host, int nlevels, int batch, int channels, int groups,
                                  float eps, int relu, float* stats, void* workspace, size_t workspace_bytes, void* stream);
size_t orp_groupnorm_backward_workspace_bytes(const orp_norm_level* levels_host, int nlevels, int batch, int channels, int groups);
int orp_groupnorm_act_multi_backward(const orp_norm_level* levels, const float* const* grad_outputs_host,
                                     float* const* grad_inputs_host, const float* const* gammas_host,
                                     const float* const* betas_host, float* const* dgammas_host, float* const* dbetas_host,
                                     int nlevels, int batch, int channels, int groups, int relu, const float* stats,
                                     void* workspace, size_t workspace_bytes, void* stream);
int orp_affine_act(const float* x, const float* residual, const float* scale, const float* shift, float* y, int batch,
                   int channels, int hw, int relu, void* stream);

/* orp_bias_act_multi: y = relu?(x + bias[c] (+ residual)), optionally y2 = y - sub[c], for ALL FPN levels in one launch:
 *   the passes around the head's bias-carrying output convolutions (orientedreppoints_head.py:156-170): conv bias,
 *   ReLU, `pts_out_refine + pts_out_init`, `pts_out_init - dcn_base_offset`.  levels_host[i] = {input, residual|NULL,
 *   output, output2|NULL, height, width}, tensors NCHW [B,C,H,W] fp32 (output may alias input); bias / sub: [C] or NULL. */
typedef struct { const float* input; const float* residual; float* output; float* output2; int height; int width; } orp_bias_level;
int orp_bias_act_multi(const orp_bias_level* levels_host, int nlevels, int batch, int channels, const float* bias,
                       const float* sub, int relu, void* stream);

/* orp_conv1x1_multi: the head's 1x1 output convolutions (reppoints_pts_init_out / reppoints_cls_out /
 *   reppoints_pts_refine_out, orientedreppoints_head.py:105-113 applied at :156-170) for ALL FPN levels in one launch,
 *   with what follows them fused in the order the head applies it: y = relu?(W.x + bias (+ residual)), optionally
 *   y2 = y - sub[k].  levels_host[i] = {input [B,Cin,H,W], residual [B,Cout,H,W] | NULL, output [B,Cout,H,W], output2 | NULL,
 *   height, width}, NCHW fp32.  weight_packed: orp_conv1x1_packed_floats(Cin) floats from orp_conv1x1_pack_weight
 *   ([Cout,Cin] -> [Cin][32]).  Requires orp_conv1x1_ok: Cin % 8 == 0, Cout <= 32.  fp32 FMA chain in channel order per
 *   channel slice, the eight slices added in a fixed order. */
size_t orp_conv1x1_packed_floats(int c_in);
int orp_conv1x1_ok(int c_in, int c_out);
int orp_conv1x1_pack_weight(const float* weight, int c_out, int c_in, float* packed, void* stream);
int orp_conv1x1_multi(const orp_bias_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                      const float* weight_packed, const float* bias, const float* sub, int relu, void* stream);

/* orp_conv3x3_small_multi: 3x3 / stride 1 / pad 1 convolution (no bias) of the SMALL FPN levels -- the 32^2 / 16^2 / 8^2
 *   maps the head's seven 256->256 convolutions (orientedreppoints_head.py:91-132) also visit -- all levels in ONE
 *   launch: exact-fp32 MFMA implicit GEMM reading and writing NCHW [B,C,H,W] fp32 (input != output).  weight_packed: the
 *   buffer produced by orp_dcn_pack_weight for the [Cout,Cin,3,3] weight.  orp_conv3x3_small_ok: Cin % 128 == 0 and
 *   Cout % 64 == 0.  The big levels stay on the library's Winograd kernels. */
int orp_conv3x3_small_ok(int c_in, int c_out);
int orp_conv3x3_small_multi(const orp_norm_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                            const float* weight_packed, void* stream);
/* per-tensor weights (both towers' small levels in one launch): weights_packed_host[i] belongs to levels_host[i] */
int orp_conv3x3_small_multi_ex(const orp_norm_level* levels_host, const float* const* weights_packed_host, int nlevels,
                               int batch, int c_in, int c_out, void* stream);
/* ... with a stride per level (1 or 2; pad 1): levels_host[i].height / width are the INPUT sizes, the output of level i is
 * [B, c_out, (H - 1) / s + 1, (W - 1) / s + 1].  Stride 2 = the FPN's extra levels P6 / P7 (mmdet/models/necks/fpn.py:160-174;
 * ConvModule(stride=2) -> F.conv2d).  With a workspace (orp_conv3x3_small_workspace_bytes; NULL = none) launches with
 * few positions also split K over the grid: partial images summed in slice order by a second launch -- fixed summation
 * order, bitwise reproducible (the library's split-K kernel for these shapes accumulates with atomics and is not). */
size_t orp_conv3x3_small_workspace_bytes(const orp_norm_level* levels_host, const int* strides_host, int nlevels, int batch,
                                         int c_out);
int orp_conv3x3_small_multi_strided(const orp_norm_level* levels_host, const float* const* weights_packed_host,
                                    const int* strides_host, int nlevels, int batch, int c_in, int c_out, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* orp_conv_split_multi: the head's tower convolutions (mmdet/models/anchor_heads/orientedreppoints_head.py:91-113 cls_convs /
 *   reg_convs, :107 reppoints_pts_init_conv; mmdet/ops/conv_module.py:130-140 `self.conv(x)`) -- kh x kw convolution of ALL FPN
 *   levels, ONE layer (weight_b_packed NULL) or TWO layers of equal shape (the two towers' layer k: grid halves of one
 *   launch), fp32 in / fp32 out / fp32 accumulation on the bf16 matrix pipe with every operand split exactly into three bf16
 *   pieces (nprod = 6 or 9 partial products; the DeformConv forward's kernel without offsets, orp_dcn_set_split_mode), or
 *   (nprod = 3) into two fp16 pieces after an exact power-of-two range scaling (products hi*hi, hi*lo, lo*hi; max |x| of the
 *   inputs is taken by a pre-pass into `workspace`, >= 256 bytes of device memory, which nprod = 6 / 9 do not need -- or,
 *   amax_in != NULL, the producer of the inputs left an UPPER BOUND of it there as float bits (device memory;
 *   orp_groupnorm_act_multi_cl_amax, orp_nchw_to_nhwc_multi_amax): layer a reads amax_in[0], layer b amax_in[amax_stride],
 *   amax_stride 0 or 1; no pre-pass then).
 *   input_* : channels-last [B, H, W, Cin]; output_* : [B, Cout, Ho, Wo] (out_layout 0) or [B, Ho, Wo, Cout] (1);
 *   weight_*_packed: orp_dcn_pack_weight of the [Cout, Cin, kh, kw] weight; bias_* [Cout] or NULL; relu fused.
 *   orp_conv_split_ok: Cin % 64 == 0, Cout % 64 == 0, kh * kw <= 9.
 * orp_nchw_to_nhwc_multi: [B, C, H, W] -> [B, H, W, C] of up to 16 tensors in one launch (the FPN outputs entering the
 *   towers); levels_host[i] = {input, output, height, width}. */
typedef struct { const float* input_a; const float* input_b; float* output_a; float* output_b; int height; int width; } orp_conv_level;
int orp_conv_split_ok(int c_in, int c_out, int kh, int kw);
int orp_conv_split_multi(const orp_conv_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                         const float* weight_a_packed, const float* weight_b_packed, const float* bias_a, const float* bias_b,
                         int relu, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                         int out_layout, int nprod, void* workspace, size_t workspace_bytes, const uint32_t* amax_in,
                         int amax_stride, void* stream);
/* one layer PER LEVEL (the FPN's output convolutions, mmdet/models/necks/fpn.py:150-153 `self.fpn_convs[i](laterals[i])`):
 * weights_packed_host[i] / biases_host[i] (biases_host or its entries may be NULL) belong to levels_host[i]; input_b / output_b
 * are ignored */
int orp_conv_split_multi_ex(const orp_conv_level* levels_host, const float* const* weights_packed_host,
                            const float* const* biases_host, int nlevels, int batch, int c_in, int c_out, int relu, int kh, int kw,
                            int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int out_layout, int nprod,
                            void* workspace, size_t workspace_bytes, const uint32_t* amax_in, void* stream);
/* The tower ConvModules are conv -> GroupNorm -> ReLU (mmdet/models/anchor_heads/orientedreppoints_head.py:91-113, applied per level
 * by forward_single :148-158; mmdet/ops/conv_module.py:130-140 `conv`, `norm`, `activate`).  Fused around orp_conv_split_multi
 * (inference): the normalisation's statistics come out of the convolution's own epilogue and its affine + ReLU are applied by the NEXT
 * layer while it reads the tensor -- the normalised tensor of an inner layer never exists in HBM.
 *   orp_conv_split_multi_gn   one or two layers ('same' stride-1 convolutions, no bias, channels-last in / out) over all levels.
 *       coef_in (or NULL): [layer][level][image][c_in] pairs (a, b) from orp_conv_split_gn_finish of the previous layer; the inputs
 *       are read as relu_in ? max(x a[c] + b[c], 0) : x a[c] + b[c] (c_in <= 512).  partials (out): per (layer, tile, group)
 *       (mean, M2 about it, max |y|, count) of the RAW outputs, partial_floats >= orp_conv_split_gn_partial_floats(...); no tile
 *       spans two images.  c_out / groups must divide 32.  amax_in / amax_stride / amax_count (nprod = 3): layer k scales its
 *       samples by the maximum of the amax_count words at amax_in[k * amax_stride] (bound_out of the previous finish); with coef_in
 *       and no amax_in the launch runs as nprod = 6.
 *   orp_conv_split_gn_finish  merges the tiles' statistics per (tensor, image, group) in tile order (Chan et al.; one wave each) ->
 *       coef_out [nlayers * nlevels][batch][channels] pairs (a = rstd gamma_c, b = beta_c - mean a); tensor i = layer * nlevels + level
 *       takes gammas_host[i] / betas_host[i]; bound_out (or NULL): [nlayers][nlevels * batch * groups] float bits of upper bounds
 *       of max |y| after the affine.  levels_host: the same levels as the launch that wrote `partials`.
 *   orp_affine_act_multi_cl   y = relu?(x a[c] + b[c]) for nlevels channels-last tensors with coef [tensor][batch][channels] (the last
 *       layer's normalisation, materialised; in place allowed); bound_in (or NULL): nsets x per_set words folded into slot_out[nsets].
 * Statistics partition differently from orp_groupnorm_act_multi_cl (tiles instead of 16-position chunks): results agree to a few
 * 1e-7 of scale, not bit for bit; every step is fixed-order (bitwise reproducible). */
size_t orp_conv_split_gn_partial_floats(const orp_conv_level* levels_host, int nlevels, int batch, int groups, int nlayers);
int orp_conv_split_multi_gn(const orp_conv_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                            const float* weight_a_packed, const float* weight_b_packed, int kh, int kw, int pad_h, int pad_w,
                            int dil_h, int dil_w, int nprod, const float* coef_in, int relu_in, float* partials,
                            size_t partial_floats, int groups, void* workspace, size_t workspace_bytes, const uint32_t* amax_in,
                            int amax_stride, int amax_count, void* stream);
int orp_conv_split_gn_finish(const orp_conv_level* levels_host, int nlevels, int batch, int channels, int groups, int nlayers,
                             float eps, const float* const* gammas_host, const float* const* betas_host, const float* partials,
                             float* coef_out, uint32_t* bound_out, void* stream);
int orp_affine_act_multi_cl(const orp_norm_level* levels, int nlevels, int batch, int channels, const float* coef, int relu,
                            const uint32_t* bound_in, int nsets, int per_set, uint32_t* slot_out, void* stream);
int orp_nchw_to_nhwc_multi(const orp_norm_level* levels_host, int nlevels, int batch, int channels, void* stream);
/* ... leaving max |x| of the tensors of every slot (slots_host[i] in [0, nslots)) in amax_out[slot] as float bits, by
 * atomicMax; reset != 0 zeroes amax_out first (0: accumulate into what another producer left there) */
int orp_nchw_to_nhwc_multi_amax(const orp_norm_level* levels_host, int nlevels, int batch, int channels, const int* slots_host,
                                uint32_t* amax_out, int nslots, int reset, void* stream);

/* orp_conv_wgrad_split: the weight gradient of those convolutions in training (what autograd runs behind ConvModule.conv,
 *   mmdet/ops/conv_module.py:130-140): grad_weight [Cout, Cin, kh, kw] = sum over all levels, images and positions of
 *   grad_output[b, o, p] * input[b, c, p + shift(tap)], stride 1, 'same' padding, Cin = Cout = 256 (orp_conv_wgrad_split_ok).
 *   levels_host[i] = {input, grad_output, height, width}, both NCHW fp32 -- positions are the contraction axis and contiguous
 *   per channel row, so no transposition.  fp16-pieces arithmetic (two pieces per operand after a power-of-two range scaling,
 *   fp32 accumulation); amax_x / amax_g: device scalars (float bits of an upper bound of max |input| / max |grad_output| over
 *   all levels) from the producers, or NULL (both are then taken by a pre-pass).  Fixed summation order: deterministic. */
typedef struct { const float* input; const float* grad_output; int height; int width; } orp_wgrad_level;
int orp_conv_wgrad_split_ok(int c_in, int c_out, int kh, int kw);
size_t orp_conv_wgrad_split_workspace_bytes(const orp_wgrad_level* levels_host, int nlevels, int batch, int kh, int kw);
int orp_conv_wgrad_split(const orp_wgrad_level* levels_host, int nlevels, int batch, int c_in, int c_out, int kh, int kw,
                         int pad_h, int pad_w, int dil_h, int dil_w, const uint32_t* amax_x, const uint32_t* amax_g,
                         float* grad_weight, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused test-time post-processing around the rotated NMS: replaces the tensor-op chains of
 * get_bboxes_single (orientedreppoints_head.py:707-779), multiclass_rnms (bbox_nms.py:93-182) and rbbox2result
 * (transforms.py:356-375) with fixed-shape, stream-ordered kernels -- no host synchronisation, hipGraph-capturable.
 * All pointers are device pointers unless named *_host.
 *   orp_pp_select : the candidate list itself (head :730-737: `scores.max(dim=1)`, `max_scores.topk(nms_pre)` per level):
 *     sig_all [num_classes, n] sigmoid scores of all levels; per level with more than nms_pre points the nms_pre points
 *     with the highest class-maximum score in DESCENDING score order (exact ties: ascending index), other levels in grid
 *     order; cand [sum_l min(n_l, nms_pre)] int64 global point indices.  nms_pre <= 4096; scratch:
 *     orp_pp_select_scratch_bytes(n, nms_pre, nlevels) bytes.
 *   orp_pp_gather : cand [m0] int64 = global point indices (levels concatenated, row-major inside a level) of the
 *     candidates in the reference's order; pts_all [18, n] = the refine offsets of all levels, (y,x)-interleaved channels;
 *     level tables (host): first point, feature-map width, stride of every level.  Writes pts_xy [m0,18] (grid units,
 *     (x,y)), centers [m0,2], strides [m0] (inputs of orp_minarearect_decode) and reppoints [m0,18] (image space).
 *   orp_pp_compact: sig_all [num_classes, n] sigmoid scores, boxes [m0,8] decoded corners.  Emits every (candidate,
 *     class) pair with score > score_thr, row-major, as dets [capacity,9] = corners + label*(max_coordinate+1), score
 *     (rows past the count: score -inf), sel_cand / sel_label [capacity], seg2 = {0, count} (the seg_offsets of
 *     orp_rnms_batched), total[0] = number of pairs found (> capacity = overflow, caller must fall back).  num_classes
 *     <= 32; scratch: orp_pp_compact_scratch_bytes(m0) bytes of device memory.
 *   orp_pp_pack   : keep / num_keep from orp_rnms_batched -> packed [max_out + 1, 28] fp32: rows = [reppoints(18) |
 *     corners(8) | score | label] in the reference's output order (ascending index, or the max_out highest scores in
 *     descending order when more survive); last row = (count, overflow, 0...).
 * ------------------------------------------------------------------------------------------------------- */
size_t orp_pp_select_scratch_bytes(int n, int nms_pre, int nlevels);
int orp_pp_select(const float* sig_all, int num_classes, int n, const int* level_offsets_host, int nlevels, int nms_pre,
                  int64_t* cand, void* scratch, size_t scratch_bytes, void* stream);
int orp_pp_gather(const float* pts_all, const int64_t* cand, int m0, int n, const int* level_offsets_host,
                  const int* level_widths_host, const float* level_strides_host, int nlevels, float* pts_xy,
                  float* centers, float* strides, float* reppoints, void* stream);
size_t orp_pp_compact_scratch_bytes(int m0);
int orp_pp_compact(const float* sig_all, const int64_t* cand, int m0, int n, int num_classes, const float* boxes,
                   float score_thr, int capacity, float* dets, int32_t* sel_cand, int32_t* sel_label, int32_t* seg2,
                   int32_t* total, void* scratch, size_t scratch_bytes, void* stream);
int orp_pp_pack(const int64_t* keep, const int32_t* num_keep, const float* dets, const int32_t* sel_cand,
                const int32_t* sel_label, const float* boxes, const float* reppoints, const int32_t* total, int capacity,
                int max_out, float* packed, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Built-in kernel timing (measurement aid for bench.py): when enabled every instrumented launch is bracketed by
 * a HIP event pair recorded on the launch stream.  Slots: 0 nms mask, 1 nms sweep, 2 nms sort, 3 dcn forward,
 * 4 minaerarect, 5 convex_iou, 6 convex_giou, 7 iou matrix, 8 dcn backward (9 / 10 / 11: its input-gradient GEMM, scatter,
 * weight-gradient parts), 12 orp_conv_split_multi, 13 orp_conv_wgrad_split.
 * orp_profile_read synchronises on the recorded events and returns their summed duration and count.
 * ------------------------------------------------------------------------------------------------------- */
int orp_profile_enable(int on);
/* Development aid (tests/checks/graph_bitwise.py): while `log` is non-NULL every fp16-pieces launch (nprod = 3) of
 * orp_conv_split_multi(_ex) / orp_dcn_forward_pair(_amax) leaves, in launch order, four words in device memory at
 * log + 4 * k (k < capacity_launches): the range word layer a / layer b actually READ and the scale of their weight planes.
 * Returns the number of launches logged since the previous call; NULL switches the log off.  No reference counterpart. */
int orp_debug_amax_log(uint32_t* log, int capacity_launches);
int orp_profile_read(int slot, double* total_ms, int* count, int reset);

#ifdef __cplusplus
}
#endif
#endif /* ORP_HIP_H_ */
