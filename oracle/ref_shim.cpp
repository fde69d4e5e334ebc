// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Host-compiles the reference's own __device__ function bodies (sliced at build time from the files under
// /root/reference by oracle/build_ref.py into a scratch include dir; nothing is copied into this repo) and
// exports them through a small extern "C" surface so tests can (1) pin oracle/orp_oracle.c against the REAL
// reference arithmetic and (2) generate tests/golden/*.npz.
//
// The only code in this file is glue: stubs for the CUDA keywords/builtins, a one-thread-per-block launch
// emulation for the element-wise kernels, and the reference's host-side greedy sweep re-expressed over the
// reference's own IoU function (rnms_kernel.cu:239-264).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <math.h>
#include <stdio.h>
#include <vector>

// ---- CUDA keyword / builtin stubs -------------------------------------------------------------------------
#define __device__
#define __global__
#define __host__
#define __shared__ static
#define __syncthreads() ((void)0)
struct orp_dim3 { int x, y, z; };
static orp_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
template <typename T> static inline T atomicAdd(T* p, T v) { T old = *p; *p = old + v; return old; }
#define CUDA_1D_KERNEL_LOOP(i, n) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x)
#define CUDA_KERNEL_LOOP(i, n) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)

// ---- the reference slices, one namespace each (they all define sig/Point/cross/...) -------------------------
namespace nsref_rnms_kernel { using std::min; using std::max;
#include "rnms_kernel.inc"
}
namespace nsref_rnms_cpu { using std::min; using std::max;
#include "rnms_cpu.inc"
}
namespace nsref_poly_nms { using namespace std;
#include "poly_nms.inc"
}
namespace nsref_poly_overlaps { using namespace std;
#include "poly_overlaps.inc"
}
namespace nsref_minarearect { using std::min; using std::max;
#include "minarearect.inc"
}
namespace nsref_convex_iou { using std::min; using std::max;
#include "convex_iou.inc"
}
namespace nsref_convex_giou { using std::min; using std::max;
#include "convex_giou.inc"
}
namespace nsref_points_justify { using std::min; using std::max;
#include "points_justify.inc"
}
namespace nsref_focal { using std::min; using std::max;
#include "focal.inc"
}
namespace nsref_chamfer { using std::min; using std::max;
#include "chamfer.inc"
}
namespace nsref_dcn { using std::min; using std::max;
#include "dcn_im2col.inc"
#include "dcn_col2im.inc"
#include "dcn_col2im_coord.inc"
#include "dcn_modulated.inc"
}
// polyiou.cpp is plain C++: include it whole (its std headers are already guarded above).
namespace nsref_polyiou {
#include "DOTA_devkit/polyiou.cpp"
}

// box_iou_rotated_utils.h is a plain header (HOST_DEVICE macros): include it where it lies (CPU branch, std::sort)
#include "mmdet/ops/box_iou_rotated/src/box_iou_rotated_utils.h"

extern "C" {

void ref_box_iou_rotated(const float* b1, int n, const float* b2, int k, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = single_box_iou_rotated<float>(b1 + i * 5, b2 + j * 5);
}

// ---- pairwise IoU scalars -------------------------------------------------------------------------------
float ref_rnms_iou(const float* p, const float* q) { return nsref_rnms_kernel::devrIoU(p, q); }
float ref_rnms_cpu_iou(const float* p, const float* q) {
  return nsref_rnms_cpu::rotate_iou(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7],
                                  q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
}
float ref_poly_nms_iou(const float* p, const float* q) { return nsref_poly_nms::devPolyIoU(p, q); }
float ref_poly_overlaps_iou(const float* b1, const float* b2) { return nsref_poly_overlaps::devPolyIoU(b1, b2); }
double ref_polyiou_iou(const double* p, const double* q) {
  std::vector<double> P(p, p + 8), Q(q, q + 8);
  return nsref_polyiou::iou_poly(P, Q);
}
float ref_convex_iou(const float* pts18, const float* gt8) { return nsref_convex_iou::devrIoU(pts18, gt8); }
// out19 = 18 grads + giou, exactly the row layout of convex_giou_kernel (convex_giou_kernel.cu:817-823)
void ref_convex_giou(const float* pts18, const float* gt8, float* out19) {
  float g = nsref_convex_giou::devrIoU(pts18, gt8, out19, 0);
  out19[18] = g;
}
void ref_minarearect(const float* pts18, float* out8) { nsref_minarearect::Findminbox(pts18, out8); }

// ---- batched helpers ------------------------------------------------------------------------------------
void ref_rnms_iou_matrix(const float* a, int n, const float* b, int k, int stride, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = nsref_rnms_kernel::devrIoU(a + i * stride, b + j * stride);
}
void ref_poly_overlaps(const float* boxes, int n, const float* query, int k, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = nsref_poly_overlaps::devPolyIoU(boxes + i * 5, query + j * 5);
}
void ref_convex_iou_matrix(const float* pts, int n, const float* gts, int k, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = nsref_convex_iou::devrIoU(pts + i * 18, gts + j * 8);
}
void ref_convex_giou_batch(const float* pts, const float* gts, int n, float* out19) {
  for (int i = 0; i < n; i++) ref_convex_giou(pts + i * 18, gts + i * 8, out19 + (size_t)i * 19);
}
void ref_minarearect_batch(const float* pts, int n, float* out) {
  for (int i = 0; i < n; i++) nsref_minarearect::Findminbox(pts + i * 18, out + i * 8);
}

// Greedy sweep of rnms_cuda / _poly_nms over ALREADY SORTED dets[n,9] (rnms_kernel.cu:149-201 tile rule:
// row box i suppresses column box j>i iff IoU(row=i, col=j) > thr; host sweep :239-257).
// which: 0 = rnms_kernel devrIoU, 1 = poly_nms devPolyIoU.  keep_out gets sorted-order positions.
int ref_nms_sorted(const float* dets, int n, float thr, int which, int* keep_out) {
  std::vector<unsigned char> removed(n, 0);
  int nk = 0;
  for (int i = 0; i < n; i++) {
    if (removed[i]) continue;
    keep_out[nk++] = i;
    for (int j = i + 1; j < n; j++) {
      if (removed[j]) continue;  // a bit already set stays set: skipping is equivalent to OR-ing
      float v = which == 0 ? nsref_rnms_kernel::devrIoU(dets + i * 9, dets + j * 9)
                           : nsref_poly_nms::devPolyIoU(dets + i * 9, dets + j * 9);
      if (v > thr) removed[j] = 1;
    }
  }
  return nk;
}

// ---- CPU baselines of bench.py: the reference's OWN polyiou.cpp / rnms_cpu.cpp arithmetic ----------------
// The greedy loops are the reference's Python loops (ResultMerge.py:18-41, ResultMerge_multi_process.py:60-121) restated
// in C++ around the reference's own iou_poly -- i.e. WITHOUT the SWIG / numpy call overhead of the real script, which
// makes this a conservative (fast) baseline.  dets [n,9] fp64, order = numpy's scores.argsort()[::-1].
int ref_py_cpu_nms_poly(const double* dets, int n, const int64_t* order, double thr, int64_t* keep_out) {
  std::vector<std::vector<double> > polys(n);
  for (int i = 0; i < n; i++) polys[i].assign(dets + (size_t)i * 9, dets + (size_t)i * 9 + 8);
  std::vector<unsigned char> removed(n, 0);
  int nk = 0;
  for (int a = 0; a < n; a++) {
    if (removed[a]) continue;
    const int64_t i = order[a];
    keep_out[nk++] = i;
    for (int b = a + 1; b < n; b++) {
      if (removed[b]) continue;
      const double v = nsref_polyiou::iou_poly(polys[i], polys[order[b]]);
      if (!(v <= thr)) removed[b] = 1;                      // inds = np.where(ovr <= thresh)
    }
  }
  return nk;
}
// py_cpu_nms_poly_fast: polyiou only where the horizontal bounding boxes overlap (hbb_ovr > 0), else the HBB IoU (0)
int ref_py_cpu_nms_poly_fast(const double* dets, int n, const int64_t* order, double thr, int64_t* keep_out) {
  std::vector<std::vector<double> > polys(n);
  std::vector<double> x1(n), y1(n), x2(n), y2(n), area(n);
  for (int i = 0; i < n; i++) {
    const double* d = dets + (size_t)i * 9;
    polys[i].assign(d, d + 8);
    x1[i] = std::min(std::min(d[0], d[2]), std::min(d[4], d[6])); x2[i] = std::max(std::max(d[0], d[2]), std::max(d[4], d[6]));
    y1[i] = std::min(std::min(d[1], d[3]), std::min(d[5], d[7])); y2[i] = std::max(std::max(d[1], d[3]), std::max(d[5], d[7]));
    area[i] = (x2[i] - x1[i] + 1) * (y2[i] - y1[i] + 1);
  }
  std::vector<unsigned char> removed(n, 0);
  int nk = 0;
  for (int a = 0; a < n; a++) {
    if (removed[a]) continue;
    const int64_t i = order[a];
    keep_out[nk++] = i;
    for (int b = a + 1; b < n; b++) {
      if (removed[b]) continue;
      const int64_t j = order[b];
      const double w = std::max(0.0, std::min(x2[i], x2[j]) - std::max(x1[i], x1[j]));
      const double h = std::max(0.0, std::min(y2[i], y2[j]) - std::max(y1[i], y1[j]));
      const double hi = w * h;
      double ovr = hi / (area[i] + area[j] - hi);
      if (ovr > 0) ovr = nsref_polyiou::iou_poly(polys[i], polys[j]);
      if (!(ovr <= thr)) removed[b] = 1;
    }
  }
  return nk;
}
// hard NMS (soft_rnms method 0) with rnms_cpu.cpp's own fp32 rotate_iou over SORTED dets [n,9]
int ref_rnms_cpu_hard(const float* dets, int n, float thr, int* keep_out) {
  std::vector<unsigned char> removed(n, 0);
  int nk = 0;
  for (int i = 0; i < n; i++) {
    if (removed[i]) continue;
    keep_out[nk++] = i;
    const float* p = dets + (size_t)i * 9;
    for (int j = i + 1; j < n; j++) {
      if (removed[j]) continue;
      const float* q = dets + (size_t)j * 9;
      const float v = nsref_rnms_cpu::rotate_iou(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]);
      if (v > thr) removed[j] = 1;
    }
  }
  return nk;
}
// n independent pairs through polyiou.cpp's iou_poly (us per IoU)
void ref_polyiou_many(const double* a, const double* b, int n, double* out) {
  for (int i = 0; i < n; i++) {
    std::vector<double> P(a + (size_t)i * 8, a + (size_t)i * 8 + 8), Q(b + (size_t)i * 8, b + (size_t)i * 8 + 8);
    out[i] = nsref_polyiou::iou_poly(P, Q);
  }
}

// ---- element-wise kernels: emulate <<<n blocks, 1 thread>>> ------------------------------------------------
void ref_points_justify(const float* points, int rows, const float* polys, int cols, float* out) {
  int n = rows * cols;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) { blockIdx.x = b; nsref_points_justify::PointsJF<float>(n, points, polys, rows, cols, out); }
  blockIdx.x = 0; gridDim.x = 1;
}
void ref_focal_forward(const float* logits, const int64_t* targets, int num, int classes, float gamma, float alpha,
                       float* losses) {
  int n = num * classes;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) { blockIdx.x = b; nsref_focal::SigmoidFocalLossForward<float>(n, logits, targets, classes, gamma, alpha, num, losses); }
  blockIdx.x = 0; gridDim.x = 1;
}
void ref_focal_backward(const float* logits, const int64_t* targets, const float* d_losses, int num, int classes,
                        float gamma, float alpha, float* d_logits) {
  int n = num * classes;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) { blockIdx.x = b; nsref_focal::SigmoidFocalLossBackward<float>(n, logits, targets, d_losses, classes, gamma, alpha, num, d_logits); }
  blockIdx.x = 0; gridDim.x = 1;
}
// chamfer: one block of one thread walks everything (the kernel is correct for any launch shape).
void ref_chamfer_nn(int b, int n, const float* xyz, int m, const float* xyz2, float* result, int* result_i) {
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = 1; blockIdx.x = 0; gridDim.y = 1; blockIdx.y = 0;
  nsref_chamfer::NmDistanceKernel(b, n, xyz, m, xyz2, result, result_i);
}

// deformable_im2col_gpu_kernel (deform_conv_cuda_kernel.cu:190-243) under the 1-thread-per-block emulation.
// data_col [C*kh*kw][B][Ho][Wo]
void ref_dcn_im2col(const float* im, const float* offset, int B, int C, int H, int W, int kh, int kw, int pad_h,
                    int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* col) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int n = C * Ho * Wo * B;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) {
    blockIdx.x = b;
    nsref_dcn::deformable_im2col_gpu_kernel<float>(n, im, offset, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h,
                                                   dil_w, C / dg, B, C, dg, Ho, Wo, col);
  }
  blockIdx.x = 0; gridDim.x = 1;
}

// deformable_col2im_gpu_kernel (:279-335) and deformable_col2im_coord_gpu_kernel (:373-436), 1 thread per block.
// col [C*kh*kw][B][Ho][Wo]; grad_im [B,C,H,W] (accumulated, caller zeroes); grad_offset [B, dg*2*kh*kw, Ho, Wo]
void ref_dcn_col2im(const float* col, const float* offset, int B, int C, int H, int W, int kh, int kw, int pad_h,
                    int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* grad_im) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int n = C * kh * kw * Ho * Wo * B;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) {
    blockIdx.x = b;
    nsref_dcn::deformable_col2im_gpu_kernel<float>(n, col, offset, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w,
                                                   dil_h, dil_w, C / dg, B, dg, Ho, Wo, grad_im);
  }
  blockIdx.x = 0; gridDim.x = 1;
}
void ref_dcn_col2im_coord(const float* col, const float* im, const float* offset, int B, int C, int H, int W, int kh,
                          int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                          float* grad_offset) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int n = Ho * Wo * 2 * kh * kw * dg * B;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) {
    blockIdx.x = b;
    nsref_dcn::deformable_col2im_coord_gpu_kernel<float>(n, col, im, offset, C, H, W, kh, kw, pad_h, pad_w, stride_h,
                                                         stride_w, dil_h, dil_w, C * kh * kw / dg, B,
                                                         2 * kh * kw * dg, dg, Ho, Wo, grad_offset);
  }
  blockIdx.x = 0; gridDim.x = 1;
}

// DCNv2: modulated_deformable_{im2col,col2im,col2im_coord}_gpu_kernel (deform_conv_cuda_kernel.cu:570-767) under the
// 1-thread-per-block emulation, driven PER IMAGE with batch_size = 1 exactly as the reference's host side does
// (deform_conv_cuda.cpp:540-545, 636-655).  Single-image layouts: im [C,H,W], offset [dg*2*taps,Ho,Wo],
// mask [dg*taps,Ho,Wo], col [C*taps][Ho*Wo].
void ref_dcn_v2_im2col(const float* im, const float* offset, const float* mask, int C, int H, int W, int kh, int kw,
                       int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* col) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int n = C * 1 * Ho * Wo;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) {
    blockIdx.x = b;
    nsref_dcn::modulated_deformable_im2col_gpu_kernel<float>(n, im, offset, mask, H, W, kh, kw, pad_h, pad_w, stride_h,
                                                             stride_w, dil_h, dil_w, C / dg, 1, C, dg, Ho, Wo, col);
  }
  blockIdx.x = 0; gridDim.x = 1;
}
// grad_im [C,H,W] accumulated (caller zeroes)
void ref_dcn_v2_col2im(const float* col, const float* offset, const float* mask, int C, int H, int W, int kh, int kw,
                       int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* grad_im) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int n = C * kh * kw * 1 * Ho * Wo;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) {
    blockIdx.x = b;
    nsref_dcn::modulated_deformable_col2im_gpu_kernel<float>(n, col, offset, mask, C, H, W, kh, kw, pad_h, pad_w,
                                                             stride_h, stride_w, dil_h, dil_w, C / dg, 1, dg, Ho, Wo,
                                                             grad_im);
  }
  blockIdx.x = 0; gridDim.x = 1;
}
// grad_offset [dg*2*taps,Ho,Wo], grad_mask [dg*taps,Ho,Wo]
void ref_dcn_v2_col2im_coord(const float* col, const float* im, const float* offset, const float* mask, int C, int H,
                             int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                             int dil_w, int dg, float* grad_offset, float* grad_mask) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  int n = 1 * Ho * Wo * 2 * kh * kw * dg;
  blockDim.x = 1; threadIdx.x = 0; gridDim.x = n;
  for (int b = 0; b < n; b++) {
    blockIdx.x = b;
    nsref_dcn::modulated_deformable_col2im_coord_gpu_kernel<float>(n, col, im, offset, mask, C, H, W, kh, kw, pad_h,
                                                                   pad_w, stride_h, stride_w, dil_h, dil_w,
                                                                   C * kh * kw / dg, 1, 2 * kh * kw * dg, dg, Ho, Wo,
                                                                   grad_offset, grad_mask);
  }
  blockIdx.x = 0; gridDim.x = 1;
}

}  // extern "C"
