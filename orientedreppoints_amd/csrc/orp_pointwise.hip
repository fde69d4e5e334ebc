// orp_pointwise.hip -- the small per-element ops of the APAA / loss path on gfx950:
//   pointsJf          mmdet/ops/point_justify/src/points_justify_kernel.cu:25-119
//   ChamferDistance2D mmdet/ops/chamfer_2d/src/chamfer_2d.cu:12-182
//   sigmoid focal     mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:23-167
//   segment losses    the per-level / per-stage GIoU and SpatialBorder losses of the head's loss() (orientedreppoints_head.py
//                     :294-318,474-520 over iou_loss.py:69-129 and spatial_border_loss.py:8-92) as a rows kernel + ONE
//                     fixed-order segment reduction each (the framework composition is ~25 tiny launches per loss)
// All are HBM-bound element-wise / tiny-reduction kernels: grid-stride, coalesced, no host synchronisation.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_libm.hpp"

namespace {

constexpr int kThreads = 256;
inline int grid_for(long n) { long b = (n + kThreads - 1) / kThreads; if (b > 256L * 32) b = 256L * 32; if (b < 1) b = 1; return (int)b; }

// ---- point in quad (ray casting with the reference's early `break` semantics) -------------------------------------
// Returns 1.0f when the crossing count accumulated so far is odd.  A vertex hit or a point exactly on an edge stops
// the edge scan (points_justify_kernel.cu:70-84), so the parity of the crossings counted BEFORE it decides.
__device__ __forceinline__ float point_in_quad(float px, float py, const float* q) {
  int ncross = 0;
  // edge order of the reference loop: (i, j) = (0,3), (1,0), (2,1), (3,2)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = (i + 3) & 3;
    const float sx = q[2 * i], sy = q[2 * i + 1], tx = q[2 * j], ty = q[2 * j + 1];
    if (py < fminf(sy, ty)) continue;
    if (py > fmaxf(sy, ty)) continue;
    if ((sx == px && sy == py) || (tx == px && ty == py)) break;
    if ((sy < py && ty >= py) || (sy >= py && ty < py)) {
      const float x = sx + (py - sy) * (tx - sx) / (ty - sy);
      if (x == px) break;
      if (x > px) ncross++;
    }
  }
  return (ncross & 1) ? 1.0f : 0.0f;
}

__global__ void points_justify_kernel(const float* __restrict__ points, int m, const float* __restrict__ polys, int k,
                                      float* __restrict__ out) {
  const long total = (long)m * k;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int row = (int)(idx / k), col = (int)(idx % k);
    float q[8];
    const float4* qp = reinterpret_cast<const float4*>(polys + (size_t)col * 8);
    float4 a = qp[0], b = qp[1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    out[idx] = point_in_quad(points[2 * (size_t)row], points[2 * (size_t)row + 1], q);
  }
}

// aligned: the 9 points of row i against quad i -> out[i, 0..9)
__global__ void points_in_quad_aligned_kernel(const float* __restrict__ pts18, const float* __restrict__ quads, int m,
                                              float* __restrict__ out9) {
  const long total = (long)m * 9;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int row = (int)(idx / 9), t = (int)(idx % 9);
    float q[8];
    const float4* qp = reinterpret_cast<const float4*>(quads + (size_t)row * 8);
    float4 a = qp[0], b = qp[1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    out9[idx] = point_in_quad(pts18[(size_t)row * 18 + 2 * t], pts18[(size_t)row * 18 + 2 * t + 1], q);
  }
}

// ---- chamfer 2d -----------------------------------------------------------------------------------------------
// one thread per (batch, query point): exact nearest neighbour, FIRST minimum wins (strict <), squared distance.
__global__ void chamfer_nn_kernel(const float* __restrict__ xyz, const float* __restrict__ xyz2, int b, int n, int m,
                                  float* __restrict__ result, int32_t* __restrict__ result_i) {
  const long total = (long)b * n;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int bi = (int)(idx / n);
    const float x1 = xyz[2 * idx], y1 = xyz[2 * idx + 1];
    const float2* o = reinterpret_cast<const float2*>(xyz2 + (size_t)bi * m * 2);
    float best = 0.f; int best_i = 0;
    for (int kk = 0; kk < m; kk++) {
      const float2 p = o[kk];
      const float x2 = p.x - x1, y2 = p.y - y1;
      const float d = x2 * x2 + y2 * y2;
      if (kk == 0 || d < best) { best = d; best_i = kk; }
    }
    result[idx] = best; result_i[idx] = best_i;
  }
}

__global__ void chamfer_grad_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, int b, int n, int m,
                                    const float* __restrict__ grad_dist1, const int32_t* __restrict__ idx1,
                                    float* __restrict__ grad_xyz1, float* __restrict__ grad_xyz2) {
  const long total = (long)b * n;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int bi = (int)(idx / n);
    const float x1 = xyz1[2 * idx], y1 = xyz1[2 * idx + 1];
    const int j2 = idx1[idx];
    const size_t o2 = ((size_t)bi * m + j2) * 2;
    const float x2 = xyz2[o2], y2 = xyz2[o2 + 1];
    const float g = grad_dist1[idx] * 2;
    atomicAdd(&grad_xyz1[2 * idx], g * (x1 - x2));
    atomicAdd(&grad_xyz1[2 * idx + 1], g * (y1 - y2));
    atomicAdd(&grad_xyz2[o2], -(g * (x1 - x2)));
    atomicAdd(&grad_xyz2[o2 + 1], -(g * (y1 - y2)));
  }
}

// ---- sigmoid focal loss -----------------------------------------------------------------------------------------
// mixed float/double expression structure kept as written in the reference (sigmoid_focal_loss_cuda.cu:36-57,73-96); expf / logf /
// powf are the HOST C library's (csrc/orp_libm.hpp), so that the losses and gradients are the bits the reference compiled for the host
// -- the parity oracle -- produces (round 6; rounds 1-5: the device library's, held to 1e-4)
__global__ void focal_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets, long total,
                                 int classes, float gamma, float alpha, float* __restrict__ losses) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / classes), d = (int)(i % classes);
    const int t = (int)targets[n];
    const float c1 = (t == (d + 1));
    const float c2 = (t >= 0 & t != (d + 1));
    const float zn = (float)(1.0 - alpha);
    const float zp = alpha;
    const float x = logits[i];
    const float p = (float)(1. / (1. + orp::libm::expf_host(-x)));
    const float term1 = orp::libm::powf_host((float)(1. - p), gamma) * orp::libm::logf_host(fmaxf(p, FLT_MIN));
    const float term2 = (float)(orp::libm::powf_host(p, gamma) *
                                (-1. * x * (x >= 0) - orp::libm::logf_host((float)(1. + orp::libm::expf_host((float)(x - 2. * x * (x >= 0)))))));
    float l = 0.0f;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}

__global__ void focal_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                 const float* __restrict__ d_losses, long total, int classes, float gamma, float alpha,
                                 float* __restrict__ d_logits) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / classes), d = (int)(i % classes);
    const int t = (int)targets[n];
    const float c1 = (t == (d + 1));
    const float c2 = (t >= 0 & t != (d + 1));
    const float zn = (float)(1.0 - alpha);
    const float zp = alpha;
    const float x = logits[i];
    const float p = (float)(1. / (1. + orp::libm::expf_host(-x)));
    const float term1 = (float)(orp::libm::powf_host((float)(1. - p), gamma) * (1. - p - (p * gamma * orp::libm::logf_host(fmaxf(p, FLT_MIN)))));
    const float term2 = (float)(orp::libm::powf_host(p, gamma) *
                                ((-1. * x * (x >= 0) - orp::libm::logf_host((float)(1. + orp::libm::expf_host((float)(x - 2. * x * (x >= 0)))))) *
                                     (1. - p) * gamma -
                                 p));
    float g = 0.0f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}


// scalar_t = double (AT_DISPATCH_FLOATING_TYPES, sigmoid_focal_loss_cuda.cu:121,160): the reference's templated expressions keep
// their SINGLE-precision transcendental calls (expf / logf / powf take a float argument and return a float) inside double
// arithmetic; restated with every implicit conversion written out
__global__ void focal_fwd_kernel_f64(const double* __restrict__ logits, const int64_t* __restrict__ targets, long total,
                                     int classes, float gamma, float alpha, double* __restrict__ losses) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / classes), d = (int)(i % classes);
    const int t = (int)targets[n];
    const double c1 = (t == (d + 1));
    const double c2 = (t >= 0 & t != (d + 1));
    const double zn = (1.0 - alpha);
    const double zp = alpha;
    const double x = logits[i];
    const double p = 1. / (1. + (double)orp::libm::expf_host((float)-x));
    const double term1 = (double)orp::libm::powf_host((float)(1. - p), gamma) * (double)orp::libm::logf_host((float)fmax(p, (double)FLT_MIN));
    const double term2 = (double)orp::libm::powf_host((float)p, gamma) *
                         (-1. * x * (x >= 0) - (double)orp::libm::logf_host((float)(1. + (double)orp::libm::expf_host((float)(x - 2. * x * (x >= 0))))));
    double l = 0.0;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}

__global__ void focal_bwd_kernel_f64(const double* __restrict__ logits, const int64_t* __restrict__ targets,
                                     const double* __restrict__ d_losses, long total, int classes, float gamma, float alpha,
                                     double* __restrict__ d_logits) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / classes), d = (int)(i % classes);
    const int t = (int)targets[n];
    const double c1 = (t == (d + 1));
    const double c2 = (t >= 0 & t != (d + 1));
    const double zn = (1.0 - alpha);
    const double zp = alpha;
    const double x = logits[i];
    const double p = 1. / (1. + (double)orp::libm::expf_host((float)-x));
    const double term1 = (double)orp::libm::powf_host((float)(1. - p), gamma) * (1. - p - (p * gamma * (double)orp::libm::logf_host((float)fmax(p, (double)FLT_MIN))));
    const double term2 = (double)orp::libm::powf_host((float)p, gamma) *
                         ((-1. * x * (x >= 0) - (double)orp::libm::logf_host((float)(1. + (double)orp::libm::expf_host((float)(x - 2. * x * (x >= 0)))))) *
                              (1. - p) * gamma -
                          p);
    double g = 0.0;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}


// ---- segment losses ---------------------------------------------------------------------------------------------------
// border: per row (a positive point set and its gt quad), over the points OUTSIDE the quad (pointsJf == 0; rows with
// weight <= 0 do not count): sum of 0.2 * |p - centre|, their number, and d(0.2 |p - c|)/dp for the backward pass.
__global__ void border_rows_kernel(const float* __restrict__ pts18, const float* __restrict__ gt8,
                                   const float* __restrict__ weight, int P, float* __restrict__ row_sum,
                                   float* __restrict__ row_cnt, float* __restrict__ gdir) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < P; row += gridDim.x * blockDim.x) {
    float q[8];
    const float4* qp = reinterpret_cast<const float4*>(gt8 + (size_t)row * 8);
    const float4 a = qp[0], b = qp[1];
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    const float cx = (q[0] + q[4]) / 2.0f, cy = (q[1] + q[5]) / 2.0f;
    const bool counts = weight[row] > 0.f;
    float sum = 0.f, cnt = 0.f;
    for (int t = 0; t < 9; t++) {
      const float px = pts18[(size_t)row * 18 + 2 * t], py = pts18[(size_t)row * 18 + 2 * t + 1];
      float gx = 0.f, gy = 0.f;
      if (counts && point_in_quad(px, py, q) == 0.f) {
        const float dx = px - cx, dy = py - cy;
        const float r = sqrtf(dx * dx + dy * dy);
        sum += 0.2f * r; cnt += 1.f;
        gx = 0.2f * dx / r; gy = 0.2f * dy / r;
      }
      gdir[(size_t)row * 18 + 2 * t] = gx; gdir[(size_t)row * 18 + 2 * t + 1] = gy;
    }
    row_sum[row] = sum; row_cnt[row] = cnt;
  }
}

// GIoU: contrib = (1 - giou) w; the gradient the reference takes out of the forward kernel, rows with any component > 1
// replaced by 1e-6 (iou_loss.py:87-89), scaled by -w / max(denom[seg], 1) * loss_weight
__global__ void giou_rows_kernel(const float* __restrict__ gious, const float* __restrict__ grad18,
                                 const float* __restrict__ weight, const int64_t* __restrict__ seg,
                                 const float* __restrict__ denom, int P, float loss_weight, float* __restrict__ contrib,
                                 float* __restrict__ gsave) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < P; row += gridDim.x * blockDim.x) {
    const float w = weight[row];
    const float d = fmaxf(denom[seg[row]], 1.0f);
    contrib[row] = (1.0f - gious[row]) * w;
    float g[18];
    bool unvalid = false;
    for (int j = 0; j < 18; j++) { g[j] = grad18[(size_t)row * 18 + j]; unvalid = unvalid || g[j] > 1.0f; }
    const float t = w / d;
    for (int j = 0; j < 18; j++) gsave[(size_t)row * 18 + j] = ((-(unvalid ? 1e-6f : g[j])) * t) * loss_weight;
  }
}

// one workgroup: per-segment sums of row values (and counts) in a FIXED order (thread i takes rows i, i + 1024, ...;
// then a tree over the threads), and the loss formulas:
//   mode 0  loss[s] = sum[s] / max(denom[s], 1) * loss_weight
//   mode 1  loss[s] = loss_weight * (sum[s] / max(cnt[s], 1)) / (denom[s] + 1e-6),  scale[s] = d loss[s] / d (a row value)
constexpr int kMaxSeg = 16;
constexpr int kSegThreads = 1024;
__global__ void __launch_bounds__(kSegThreads)
segment_finish_kernel(const float* __restrict__ row_val, const float* __restrict__ row_cnt, const int64_t* __restrict__ seg,
                      int P, int nseg, const float* __restrict__ denom, float loss_weight, int mode,
                      float* __restrict__ loss, float* __restrict__ scale) {
  __shared__ float red[kSegThreads];
  __shared__ float tot[2][kMaxSeg];
  const int tid = threadIdx.x;
  for (int s = 0; s < nseg; s++) {
    for (int which = 0; which < (mode == 1 ? 2 : 1); which++) {
      const float* src = which ? row_cnt : row_val;
      float v = 0.f;
      for (int i = tid; i < P; i += kSegThreads) if ((int)seg[i] == s) v += src[i];
      red[tid] = v;
      __syncthreads();
      for (int off = kSegThreads / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
      }
      if (tid == 0) tot[which][s] = red[0];
      __syncthreads();
    }
  }
  if (tid < nseg) {
    if (mode == 0) {
      loss[tid] = tot[0][tid] / fmaxf(denom[tid], 1.0f) * loss_weight;
    } else {
      const float n = fmaxf(tot[1][tid], 1.0f), dd = denom[tid] + 1e-6f;
      loss[tid] = loss_weight * (tot[0][tid] / n) / dd;
      if (scale) scale[tid] = loss_weight / n / dd;
    }
  }
}

inline int done() { hipError_t e = hipGetLastError(); return e == hipSuccess ? ORP_OK : (int)e; }
}  // namespace

extern "C" {
int orp_border_rows(const float* pts18, const float* gt8, const float* weight, int p, float* row_sum, float* row_cnt,
                    float* gdir, void* stream) {
  if (p < 0 || (p > 0 && (!pts18 || !gt8 || !weight || !row_sum || !row_cnt || !gdir))) return ORP_EINVAL;
  if (p == 0) return ORP_OK;
  hipLaunchKernelGGL(border_rows_kernel, dim3(grid_for(p)), dim3(kThreads), 0, (hipStream_t)stream, pts18, gt8, weight, p,
                     row_sum, row_cnt, gdir);
  return done();
}
int orp_giou_rows(const float* gious, const float* grad18, const float* weight, const int64_t* seg, const float* denom, int p,
                  float loss_weight, float* contrib, float* gsave, void* stream) {
  if (p < 0 || (p > 0 && (!gious || !grad18 || !weight || !seg || !denom || !contrib || !gsave))) return ORP_EINVAL;
  if (p == 0) return ORP_OK;
  hipLaunchKernelGGL(giou_rows_kernel, dim3(grid_for(p)), dim3(kThreads), 0, (hipStream_t)stream, gious, grad18, weight, seg,
                     denom, p, loss_weight, contrib, gsave);
  return done();
}
int orp_segment_finish(const float* row_val, const float* row_cnt, const int64_t* seg, int p, int nseg, const float* denom,
                       float loss_weight, int mode, float* loss, float* scale, void* stream) {
  if (p < 0 || nseg <= 0 || nseg > kMaxSeg || !denom || !loss || (mode != 0 && mode != 1) ||
      (p > 0 && (!row_val || !seg || (mode == 1 && !row_cnt))))
    return ORP_EINVAL;
  hipLaunchKernelGGL(segment_finish_kernel, dim3(1), dim3(kSegThreads), 0, (hipStream_t)stream, row_val, row_cnt, seg, p, nseg,
                     denom, loss_weight, mode, loss, scale);
  return done();
}
int orp_points_justify(const float* points, int m, const float* polygons, int k, float* out, void* stream) {
  if (m < 0 || k < 0 || ((m > 0 && k > 0) && (!points || !polygons || !out))) return ORP_EINVAL;
  if (m == 0 || k == 0) return ORP_OK;
  hipLaunchKernelGGL(points_justify_kernel, dim3(grid_for((long)m * k)), dim3(kThreads), 0, (hipStream_t)stream, points,
                     m, polygons, k, out);
  return done();
}
int orp_points_in_quad_aligned(const float* pts18, const float* quads, int m, float* out9, void* stream) {
  if (m < 0 || (m > 0 && (!pts18 || !quads || !out9))) return ORP_EINVAL;
  if (m == 0) return ORP_OK;
  hipLaunchKernelGGL(points_in_quad_aligned_kernel, dim3(grid_for((long)m * 9)), dim3(kThreads), 0, (hipStream_t)stream,
                     pts18, quads, m, out9);
  return done();
}
int orp_chamfer2d_forward(const float* xyz1, const float* xyz2, int b, int n, int m, float* dist1, float* dist2,
                          int32_t* idx1, int32_t* idx2, void* stream) {
  if (b < 0 || n < 0 || m < 0) return ORP_EINVAL;
  if (b == 0 || n == 0 || m == 0) return ORP_OK;
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2) return ORP_EINVAL;
  hipLaunchKernelGGL(chamfer_nn_kernel, dim3(grid_for((long)b * n)), dim3(kThreads), 0, (hipStream_t)stream, xyz1, xyz2,
                     b, n, m, dist1, idx1);
  hipLaunchKernelGGL(chamfer_nn_kernel, dim3(grid_for((long)b * m)), dim3(kThreads), 0, (hipStream_t)stream, xyz2, xyz1,
                     b, m, n, dist2, idx2);
  return done();
}
int orp_chamfer2d_backward(const float* xyz1, const float* xyz2, int b, int n, int m, const float* grad_dist1,
                           const float* grad_dist2, const int32_t* idx1, const int32_t* idx2, float* grad_xyz1,
                           float* grad_xyz2, void* stream) {
  if (b < 0 || n < 0 || m < 0) return ORP_EINVAL;
  if (b == 0 || n == 0 || m == 0) return ORP_OK;
  if (!xyz1 || !xyz2 || !grad_dist1 || !grad_dist2 || !idx1 || !idx2 || !grad_xyz1 || !grad_xyz2) return ORP_EINVAL;
  hipLaunchKernelGGL(chamfer_grad_kernel, dim3(grid_for((long)b * n)), dim3(kThreads), 0, (hipStream_t)stream, xyz1,
                     xyz2, b, n, m, grad_dist1, idx1, grad_xyz1, grad_xyz2);
  hipLaunchKernelGGL(chamfer_grad_kernel, dim3(grid_for((long)b * m)), dim3(kThreads), 0, (hipStream_t)stream, xyz2,
                     xyz1, b, m, n, grad_dist2, idx2, grad_xyz2, grad_xyz1);
  return done();
}
int orp_sigmoid_focal_loss_forward(const float* logits, const int64_t* targets, int num, int classes, float gamma,
                                   float alpha, float* losses, void* stream) {
  if (num < 0 || classes < 0) return ORP_EINVAL;
  if (num == 0 || classes == 0) return ORP_OK;
  if (!logits || !targets || !losses) return ORP_EINVAL;
  const long total = (long)num * classes;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     total, classes, gamma, alpha, losses);
  return done();
}
int orp_sigmoid_focal_loss_backward(const float* logits, const int64_t* targets, const float* d_losses, int num,
                                    int classes, float gamma, float alpha, float* d_logits, void* stream) {
  if (num < 0 || classes < 0) return ORP_EINVAL;
  if (num == 0 || classes == 0) return ORP_OK;
  if (!logits || !targets || !d_losses || !d_logits) return ORP_EINVAL;
  const long total = (long)num * classes;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     d_losses, total, classes, gamma, alpha, d_logits);
  return done();
}
int orp_sigmoid_focal_loss_forward_f64(const double* logits, const int64_t* targets, int num, int classes, float gamma,
                                       float alpha, double* losses, void* stream) {
  if (num < 0 || classes < 0) return ORP_EINVAL;
  if (num == 0 || classes == 0) return ORP_OK;
  if (!logits || !targets || !losses) return ORP_EINVAL;
  const long total = (long)num * classes;
  hipLaunchKernelGGL(focal_fwd_kernel_f64, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     total, classes, gamma, alpha, losses);
  return done();
}
int orp_sigmoid_focal_loss_backward_f64(const double* logits, const int64_t* targets, const double* d_losses, int num,
                                        int classes, float gamma, float alpha, double* d_logits, void* stream) {
  if (num < 0 || classes < 0) return ORP_EINVAL;
  if (num == 0 || classes == 0) return ORP_OK;
  if (!logits || !targets || !d_losses || !d_logits) return ORP_EINVAL;
  const long total = (long)num * classes;
  hipLaunchKernelGGL(focal_bwd_kernel_f64, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, logits, targets,
                     d_losses, total, classes, gamma, alpha, d_logits);
  return done();
}
}
