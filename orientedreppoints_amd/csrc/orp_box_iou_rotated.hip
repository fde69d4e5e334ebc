// orp_box_iou_rotated.hip -- pairwise IoU of (cx, cy, w, h, theta[rad]) boxes for gfx950.
//
// Replaces box_iou_rotated_cuda_kernel / box_iou_rotated_cuda (mmdet/ops/box_iou_rotated/src/box_iou_rotated_cuda.cu:
// 14-94) with the algorithm of box_iou_rotated_utils.h:50-341 (edge-edge intersections + contained vertices ->
// Graham scan -> triangle-fan area on centre-shifted boxes).  Layout: lane = column (boxes2) so the [N,K] row is
// written coalesced, the row box is wave-uniform; the <=24 candidate points live in a per-lane LDS column
// (stride = workgroup size: bank-conflict free for any per-lane index) instead of two 24-point private arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"

namespace {
struct P2 { float x, y; };
constexpr int kThreads = 128;
#define ORP_BHD __host__ __device__ __forceinline__

ORP_BHD float dot2(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }
ORP_BHD float crs2(P2 a, P2 b) { return a.x * b.y - b.x * a.y; }
ORP_BHD P2 sub2(P2 a, P2 b) { P2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }

ORP_BHD void vertices(const float* box, P2* pts) {
  const double theta = box[4];
  const float c2 = (float)cos(theta) * 0.5f, s2 = (float)sin(theta) * 0.5f;
  pts[0].x = box[0] - s2 * box[3] - c2 * box[2];
  pts[0].y = box[1] + c2 * box[3] - s2 * box[2];
  pts[1].x = box[0] + s2 * box[3] - c2 * box[2];
  pts[1].y = box[1] - c2 * box[3] - s2 * box[2];
  pts[2].x = 2 * box[0] - pts[0].x; pts[2].y = 2 * box[1] - pts[0].y;
  pts[3].x = 2 * box[0] - pts[1].x; pts[3].y = 2 * box[1] - pts[1].y;
}

// IoU of one pair (single_box_iou_rotated, box_iou_rotated_utils.h:314-341).  q / dist: scratch for the <= 24 candidate
// points, element i at q[i * STRIDE] -- a per-lane LDS column on the device (STRIDE = workgroup size), a plain array
// on the host (STRIDE = 1): the same source serves the CUDA and the CPU branch of the reference's dispatcher.
template <int STRIDE>
ORP_BHD float box_iou_rotated_pair(const float* r1, const float* r2, P2* q, float* dist) {
  const double sx = (r1[0] + r2[0]) / 2.0, sy = (r1[1] + r2[1]) / 2.0;
  const float A[5] = {(float)(r1[0] - sx), (float)(r1[1] - sy), r1[2], r1[3], r1[4]};
  const float B[5] = {(float)(r2[0] - sx), (float)(r2[1] - sy), r2[2], r2[3], r2[4]};
  const float area1 = A[2] * A[3], area2 = B[2] * B[3];
  if (area1 < 1e-14 || area2 < 1e-14) return 0.f;
  P2 p1[4], p2[4], v1[4], v2[4];
  vertices(A, p1); vertices(B, p2);
#pragma unroll
  for (int i = 0; i < 4; i++) { v1[i] = sub2(p1[(i + 1) & 3], p1[i]); v2[i] = sub2(p2[(i + 1) & 3], p2[i]); }
  int num = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float det = crs2(v2[j], v1[i]);
      if (fabs(det) <= 1e-14) continue;
      const P2 v12 = sub2(p2[j], p1[i]);
      const float t1 = crs2(v2[j], v12) / det, t2 = crs2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        P2 x; x.x = p1[i].x + v1[i].x * t1; x.y = p1[i].y + v1[i].y * t1;
        q[num * STRIDE] = x; num++;
      }
    }
  }
  {
    const P2 AB = v2[0], DA = v2[3];
    const float abab = dot2(AB, AB), adad = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const P2 AP = sub2(p1[i], p2[0]);
      const float apab = dot2(AP, AB), apad = -dot2(AP, DA);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) { q[num * STRIDE] = p1[i]; num++; }
    }
  }
  {
    const P2 AB = v1[0], DA = v1[3];
    const float abab = dot2(AB, AB), adad = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const P2 AP = sub2(p2[i], p1[0]);
      const float apab = dot2(AP, AB), apad = -dot2(AP, DA);
      if (apab >= 0 && apad >= 0 && apab <= abab && apad <= adad) { q[num * STRIDE] = p2[i]; num++; }
    }
  }
  float inter = 0.f;
  if (num > 2) {
    int t = 0;
    P2 best = q[0];
    for (int i = 1; i < num; i++) { const P2 v = q[i * STRIDE]; if (v.y < best.y || (v.y == best.y && v.x < best.x)) { t = i; best = v; } }
    for (int i = 0; i < num; i++) q[i * STRIDE] = sub2(q[i * STRIDE], best);
    { const P2 tmp = q[0]; q[0] = q[t * STRIDE]; q[t * STRIDE] = tmp; }
    for (int i = 0; i < num; i++) { const P2 v = q[i * STRIDE]; dist[i * STRIDE] = dot2(v, v); }
    for (int i = 1; i < num - 1; i++) {
      P2 qi = q[i * STRIDE]; float di = dist[i * STRIDE];
      for (int j = i + 1; j < num; j++) {
        const P2 qj = q[j * STRIDE]; const float dj = dist[j * STRIDE];
        const float cp = crs2(qi, qj);
        if ((cp < -1e-6) || (fabs(cp) < 1e-6 && di > dj)) {
          q[j * STRIDE] = qi; dist[j * STRIDE] = di; qi = qj; di = dj;
        }
      }
      q[i * STRIDE] = qi; dist[i * STRIDE] = di;
    }
    int kk;
    for (kk = 1; kk < num; kk++) if (dist[kk * STRIDE] > 1e-8) break;
    if (kk < num) {
      q[1 * STRIDE] = q[kk * STRIDE];
      int m = 2;
      for (int i = kk + 1; i < num; i++) {
        const P2 qi = q[i * STRIDE];
        while (m > 1 && crs2(sub2(qi, q[(m - 2) * STRIDE]), sub2(q[(m - 1) * STRIDE], q[(m - 2) * STRIDE])) >= 0) m--;
        q[m * STRIDE] = qi; m++;
      }
      if (m > 2) {
        float area = 0.f;
        const P2 q0 = q[0];
        for (int i = 1; i < m - 1; i++) area += fabs(crs2(sub2(q[i * STRIDE], q0), sub2(q[(i + 1) * STRIDE], q0)));
        inter = (float)(area / 2.0);
      }
    }
  }
  return inter / (area1 + area2 - inter);
}

__global__ void __launch_bounds__(kThreads)
box_iou_rotated_kernel(const float* __restrict__ b1, int n, const float* __restrict__ b2, int k, float* __restrict__ out) {
  __shared__ P2 s_q[24][kThreads];
  __shared__ float s_d[24][kThreads];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int row = blockIdx.y * (kThreads / 64) + wave;
  if (row >= n || col >= k) return;
  out[(size_t)row * k + col] = box_iou_rotated_pair<kThreads>(b1 + (size_t)row * 5, b2 + (size_t)col * 5,
                                                             &s_q[0][threadIdx.x], &s_d[0][threadIdx.x]);
}
}  // namespace

// CPU branch of the reference's dispatcher (box_iou_rotated.h:20-33 -> box_iou_rotated_cpu.cpp): host pointers in / out,
// the same per-pair function compiled for the host.
extern "C" int orp_box_iou_rotated_host(const float* boxes1, int n, const float* boxes2, int k, float* out) {
  if (n < 0 || k < 0 || ((n > 0 && k > 0) && (!boxes1 || !boxes2 || !out))) return ORP_EINVAL;
  P2 q[24];
  float d[24];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = box_iou_rotated_pair<1>(boxes1 + (size_t)i * 5, boxes2 + (size_t)j * 5, q, d);
  return ORP_OK;
}

extern "C" int orp_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int k, float* out, void* stream) {
  if (n < 0 || k < 0 || ((n > 0 && k > 0) && (!boxes1 || !boxes2 || !out))) return ORP_EINVAL;
  if (n == 0 || k == 0) return ORP_OK;
  dim3 grid((k + 63) / 64, (n + kThreads / 64 - 1) / (kThreads / 64));
  hipLaunchKernelGGL(box_iou_rotated_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, boxes1, n, boxes2, k, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
