// orp_eval.hip -- detection-to-ground-truth matching of the DOTA Task1 evaluation on the GPU (gfx950).
//
// Replaces the per-detection python loop of DOTA_devkit/dota_evaluation_task1.py:160-214 (voc_eval): for every
// detection, the horizontal-box pre-filter against the ground truths of its image (numpy fp64, "+ 1" pixel
// convention, `overlaps > 0`), then polyiou.iou_poly(GT, detection) (DOTA_devkit/polyiou.cpp:108-128, fp64) on the
// survivors, `np.max` / `np.argmax` over them.  Neither depends on the matching state, so all detections are
// evaluated at once -- one wavefront per detection, lanes over the image's ground truths -- and only the trivial
// tp / fp bookkeeping (dota_evaluation_task1.py:216-224) stays sequential on the host.
// The IoU is the fp64 instantiation of the triangle-fan core (orp_quadfast.hpp), bit-identical to polyiou.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"
#include "orp_quadfast.hpp"

namespace {

constexpr int kThreads = 256;          // 4 detections per workgroup

__device__ __forceinline__ double dmin4(double a, double b, double c, double d) { return fmin(fmin(a, b), fmin(c, d)); }
__device__ __forceinline__ double dmax4(double a, double b, double c, double d) { return fmax(fmax(a, b), fmax(c, d)); }

// ovmax[d] / jmax[d]: np.max / np.argmax over the ground truths of detection d's image that pass the HBB pre-filter
// (image-local ground-truth index); (-inf, -1) when none passes.  A NaN IoU wins as in numpy (first NaN index).
__global__ void __launch_bounds__(kThreads)
voc_best_match_kernel(const double* __restrict__ dets, const int32_t* __restrict__ det_img, int nd,
                      const double* __restrict__ gts, const int32_t* __restrict__ gt_off,
                      double* __restrict__ ovmax, int32_t* __restrict__ jmax) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (d >= nd) return;
  double bb[8];
#pragma unroll
  for (int k = 0; k < 8; k++) bb[k] = dets[(size_t)d * 8 + k];
  const double bb_xmin = dmin4(bb[0], bb[2], bb[4], bb[6]), bb_ymin = dmin4(bb[1], bb[3], bb[5], bb[7]);
  const double bb_xmax = dmax4(bb[0], bb[2], bb[4], bb[6]), bb_ymax = dmax4(bb[1], bb[3], bb[5], bb[7]);
  orp::QuadPrepT<double> pd;
  orp::quad_prepare<double>(bb, pd);
  const int img = det_img[d];
  const int g0 = gt_off[img], g1 = gt_off[img + 1];

  double best = -HUGE_VAL; int best_j = 0x7fffffff;     // lane-local max and its first index
  int nan_j = 0x7fffffff;                               // first NaN index seen by this lane
  for (int g = g0 + lane; g < g1; g += 64) {
    double q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = gts[(size_t)g * 8 + k];
    const double gxmin = dmin4(q[0], q[2], q[4], q[6]), gymin = dmin4(q[1], q[3], q[5], q[7]);
    const double gxmax = dmax4(q[0], q[2], q[4], q[6]), gymax = dmax4(q[1], q[3], q[5], q[7]);
    const double ixmin = fmax(gxmin, bb_xmin), iymin = fmax(gymin, bb_ymin);
    const double ixmax = fmin(gxmax, bb_xmax), iymax = fmin(gymax, bb_ymax);
    const double iw = fmax(ixmax - ixmin + 1., 0.), ih = fmax(iymax - iymin + 1., 0.);
    const double inters = iw * ih;
    const double uni = ((bb_xmax - bb_xmin + 1.) * (bb_ymax - bb_ymin + 1.) +
                        (gxmax - gxmin + 1.) * (gymax - gymin + 1.) - inters);
    const double hov = inters / uni;
    if (!(hov > 0)) continue;
    orp::QuadPrepT<double> pg;
    orp::quad_prepare<double>(q, pg);
    const double iou = orp::quad_iou_two_phase_t<double, false>(&pg, &pd);     // iou_poly(GT, detection)
    const int j = g - g0;
    if (iou != iou) { nan_j = j < nan_j ? j : nan_j; }
    else if (iou > best) { best = iou; best_j = j; }      // strict: ascending j per lane keeps the first maximum
  }
  // wave reduction: smallest NaN index; otherwise maximum value, ties to the smallest index
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int on = __shfl_xor(nan_j, o, 64);
    nan_j = on < nan_j ? on : nan_j;
    const double ob = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(best_j, o, 64);
    if (ob > best || (ob == best && oj < best_j)) { best = ob; best_j = oj; }
  }
  if (lane == 0) {
    if (nan_j != 0x7fffffff) { ovmax[d] = __longlong_as_double(0x7ff8000000000000LL); jmax[d] = nan_j; }
    else if (best_j == 0x7fffffff) { ovmax[d] = -HUGE_VAL; jmax[d] = -1; }
    else { ovmax[d] = best; jmax[d] = best_j; }
  }
}

}  // namespace

extern "C" int orp_voc_best_match_f64(const double* dets, const int32_t* det_image, int num_dets, const double* gts,
                                      const int32_t* gt_offsets, int num_images, double* ovmax, int32_t* jmax,
                                      void* stream) {
  if (num_dets < 0 || num_images < 0 || (num_dets > 0 && (!dets || !det_image || !gt_offsets || !ovmax || !jmax)))
    return ORP_EINVAL;
  if (num_dets == 0) return ORP_OK;
  const int per = kThreads / 64;
  hipLaunchKernelGGL(voc_best_match_kernel, dim3((num_dets + per - 1) / per), dim3(kThreads), 0, (hipStream_t)stream,
                     dets, det_image, num_dets, gts, gt_offsets, ovmax, jmax);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
