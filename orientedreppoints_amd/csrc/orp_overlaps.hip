// orp_overlaps.hip -- pairwise rotated / polygon IoU matrices for gfx950.
//
// Replaces
//   DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-427  (RotBox2Poly, devPolyIoU, overlaps_kernel, _overlaps)
// and provides the all-pairs fp32 quad IoU (devrIoU / devPolyIoU arithmetic) used by tests and by the merge /
// evaluation tools.  Layout: lane = column (query / b row) so the [N,K] result is written coalesced, the row box
// is wave-uniform (scalar loads); one wave per row x 64 columns, 4 waves per workgroup; clipping scratch in
// per-lane LDS columns (orp_geom.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"

namespace {
using orp::Pt;
constexpr int kThreads = 256;

// (cx,cy,w,h,theta) -> 4 corners, mixed precision exactly as poly_overlaps_kernel.cu:280-297
// (cos/sin of the float angle; products with (w / 2.0) in double; one rounding to float per coordinate).
__device__ __forceinline__ void rotbox2poly(const float* dbox, float* p8) {
  const float cs = (float)cos((double)dbox[4]);
  const float ss = (float)sin((double)dbox[4]);
  const float w = dbox[2], h = dbox[3], x_ctr = dbox[0], y_ctr = dbox[1];
  p8[0] = (float)(x_ctr + cs * (w / 2.0) - ss * (-h / 2.0));
  p8[2] = (float)(x_ctr + cs * (w / 2.0) - ss * (h / 2.0));
  p8[4] = (float)(x_ctr + cs * (-w / 2.0) - ss * (h / 2.0));
  p8[6] = (float)(x_ctr + cs * (-w / 2.0) - ss * (-h / 2.0));
  p8[1] = (float)(y_ctr + ss * (w / 2.0) + cs * (-h / 2.0));
  p8[3] = (float)(y_ctr + ss * (w / 2.0) + cs * (h / 2.0));
  p8[5] = (float)(y_ctr + ss * (-w / 2.0) + cs * (h / 2.0));
  p8[7] = (float)(y_ctr + ss * (-w / 2.0) + cs * (-h / 2.0));
}

// MODE 0: rows are quads (stride floats apart); MODE 1: rows are 5-param rotated boxes
template <int MODE, bool GUARD>
__global__ void __launch_bounds__(kThreads)
iou_matrix_kernel(const float* __restrict__ a, int n, const float* __restrict__ b, int k, int stride,
                  float* __restrict__ out) {
  __shared__ Pt<float> scratch[2 * orp::ORP_CLIP_CAP][kThreads];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.y * (kThreads / 64) + wave);
  if (row >= n) return;
  orp::PolyLds<float> P{&scratch[0][threadIdx.x], kThreads};
  orp::PolyLds<float> Q{&scratch[orp::ORP_CLIP_CAP][threadIdx.x], kThreads};
  float p8[8], q8[8];
  if (MODE == 0) {
    const float* rp = a + (size_t)row * stride;
#pragma unroll
    for (int i = 0; i < 8; i++) p8[i] = rp[i];
  } else {
    rotbox2poly(a + (size_t)row * 5, p8);
  }
  if (col >= k) return;
  if (MODE == 0) {
    const float* cp = b + (size_t)col * stride;
#pragma unroll
    for (int i = 0; i < 8; i++) q8[i] = cp[i];
  } else {
    rotbox2poly(b + (size_t)col * 5, q8);
  }
  out[(size_t)row * k + col] = orp::quad_iou<float, GUARD>(P, Q, p8, q8);
}

int launch(int mode, int guard, const float* a, int n, const float* b, int k, int stride, float* out, hipStream_t st) {
  if (n < 0 || k < 0 || ((n > 0 && k > 0) && (!a || !b || !out))) return ORP_EINVAL;
  if (n == 0 || k == 0) return ORP_OK;
  dim3 grid((k + 63) / 64, (n + kThreads / 64 - 1) / (kThreads / 64)), block(kThreads);
  if (mode == 1) hipLaunchKernelGGL((iou_matrix_kernel<1, true>), grid, block, 0, st, a, n, b, k, 5, out);
  else if (guard) hipLaunchKernelGGL((iou_matrix_kernel<0, true>), grid, block, 0, st, a, n, b, k, stride, out);
  else hipLaunchKernelGGL((iou_matrix_kernel<0, false>), grid, block, 0, st, a, n, b, k, stride, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
}  // namespace

extern "C" {
int orp_quad_iou_matrix(const float* a, int n, const float* b, int k, int stride, int guard, float* out, void* stream) {
  if (stride < 8) return ORP_EINVAL;
  return launch(0, guard, a, n, b, k, stride, out, (hipStream_t)stream);
}
int orp_poly_overlaps(const float* boxes, int n, const float* query, int k, float* out, void* stream) {
  return launch(1, 1, boxes, n, query, k, 5, out, (hipStream_t)stream);
}

// Host-pointer API, exact signature of DOTA_devkit/poly_nms_gpu/poly_overlaps.hpp:1 (errors are printed, as the
// reference's CUDA_CHECK does, poly_overlaps_kernel.cu:20-27).
void _overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k,
               int device_id) {
  if (n <= 0 || k <= 0) return;
  float *d_o = nullptr, *d_b = nullptr, *d_q = nullptr;
  int rc;
#define ORP_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "_overlaps: %s\n", hipGetErrorString(e_)); goto done; } } while (0)
  ORP_CHK(hipSetDevice(device_id));
  ORP_CHK(hipMalloc(&d_b, sizeof(float) * 5 * (size_t)n));
  ORP_CHK(hipMalloc(&d_q, sizeof(float) * 5 * (size_t)k));
  ORP_CHK(hipMalloc(&d_o, sizeof(float) * (size_t)n * (size_t)k));
  ORP_CHK(hipMemcpy(d_b, boxes_host, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice));
  ORP_CHK(hipMemcpy(d_q, query_boxes_host, sizeof(float) * 5 * (size_t)k, hipMemcpyHostToDevice));
  rc = orp_poly_overlaps(d_b, n, d_q, k, d_o, nullptr);
  if (rc != ORP_OK) { fprintf(stderr, "_overlaps: launch failed (%d)\n", rc); goto done; }
  ORP_CHK(hipMemcpy(overlaps_host, d_o, sizeof(float) * (size_t)n * (size_t)k, hipMemcpyDeviceToHost));
done:
  if (d_o) (void)hipFree(d_o);
  if (d_b) (void)hipFree(d_b);
  if (d_q) (void)hipFree(d_q);
#undef ORP_CHK
}

const char* orp_version(void) { return "orp_hip gfx950 abi1"; }
}
