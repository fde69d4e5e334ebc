// orp_train.hip -- the batched glue of the training path (gfx950): what the reference does with per-image / per-level
// Python loops of small tensor operations between its compiled operators.
//
//   orp_pointset_target      init_pointset_target_single / refine_pointset_target_single + unmap + images_to_levels
//                            (mmdet/core/bbox/pointset_target.py:61-121, 171-230): ONE launch writes labels, label weights,
//                            the gt box of every positive, proposal weights, gt indices and the positive / negative counts
//                            for all images at their full-N positions
//   orp_points_from_offsets  offset_to_pts (orientedreppoints_head.py:204-222) for every location of every level and image,
//                            and the refine-stage proposals of loss() (:378-381, which add the (y, x) offsets to the (x, y)
//                            centres WITHOUT the swap -- reproduced)
//   orp_gather_levels (+ _backward)   rows of [B, C, H, W] level tensors at selected locations -> [P, C] (optionally as
//                            image-space point sets): the head's losses only ever read the positives, so neither the
//                            [B, N, C] re-layouts (levels_to_images) nor their backward passes are materialised
//   orp_outline_samples      sampling_points (:250-292): n points on each edge of a quad
// HBM-bound, one thread per element or row; no atomics on floating-point data (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxLevels = 8;

inline int done() { hipError_t e = hipGetLastError(); return e == hipSuccess ? ORP_OK : (int)e; }

struct LevelTable {
  const float* ptr[kMaxLevels];   // [B, C, H, W]
  float* grad[kMaxLevels];
  int hw[kMaxLevels];             // H * W
  int width[kMaxLevels];
  int first[kMaxLevels + 1];      // first location of the level in the concatenated [N] order
  float stride[kMaxLevels];
  int nlev, N, B, C;
};

__device__ __forceinline__ int level_of(const LevelTable& T, int i) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < kMaxLevels; k++) l = (k < T.nlev && i >= T.first[k]) ? k : l;
  return l;
}

// ---- targets ------------------------------------------------------------------------------------------------------------------
__global__ void pointset_target_kernel(const int64_t* __restrict__ gt_inds, const uint8_t* __restrict__ valid, int B, int N,
                                       const float* __restrict__ gt_boxes, const int64_t* __restrict__ gt_labels,
                                       const int32_t* __restrict__ gt_offset, const float* __restrict__ proposals, int D,
                                       float pos_weight, int64_t* __restrict__ labels, float* __restrict__ label_weights,
                                       float* __restrict__ rbbox_gt, float* __restrict__ pos_proposals,
                                       float* __restrict__ proposal_weights, int64_t* __restrict__ gt_inds_out,
                                       int32_t* __restrict__ counts) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * N;
  int pos = 0, neg = 0, b = 0;
  if (t < total) {
    b = (int)(t / N);
    const bool ok = valid ? valid[t] != 0 : true;
    const int64_t g = ok ? gt_inds[t] : 0;                          // unmap(fill = 0) of the reference for invalid locations
    const bool is_pos = ok && g > 0, is_neg = ok && g == 0;
    pos = is_pos; neg = is_neg;
    int64_t lab = 0;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (is_pos) {
      const long k = (long)gt_offset[b] + (long)(g - 1);
      lab = gt_labels ? gt_labels[k] : 1;
      q0 = *reinterpret_cast<const float4*>(gt_boxes + k * 8);
      q1 = *reinterpret_cast<const float4*>(gt_boxes + k * 8 + 4);
    }
    labels[t] = lab;
    label_weights[t] = is_pos ? (pos_weight <= 0.f ? 1.f : pos_weight) : (is_neg ? 1.f : 0.f);
    *reinterpret_cast<float4*>(rbbox_gt + t * 8) = q0;
    *reinterpret_cast<float4*>(rbbox_gt + t * 8 + 4) = q1;
    proposal_weights[t] = is_pos ? 1.f : 0.f;
    gt_inds_out[t] = g;
    if (pos_proposals) {
      for (int d = 0; d < D; d++) pos_proposals[t * D + d] = is_pos ? proposals[t * D + d] : 0.f;
    }
  }
  // integer counts per image: wave-aggregated (a wave never straddles more than two images when N >= 64; general: by ballot per image of lane 0 / last lane)
  if (counts) {
    const int b0 = __builtin_amdgcn_readfirstlane(b);
    const unsigned long long same = __ballot(t < total && b == b0);
    const unsigned long long pm = __ballot(pos != 0), nm = __ballot(neg != 0);
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
      const int p0 = __popcll(pm & same), n0 = __popcll(nm & same);
      if (p0) atomicAdd(&counts[2 * b0], p0);
      if (n0) atomicAdd(&counts[2 * b0 + 1], n0);
    }
    if (t < total && b != b0) {                                      // the (rare) lanes of the next image: one by one
      if (pos) atomicAdd(&counts[2 * b], 1);
      if (neg) atomicAdd(&counts[2 * b + 1], 1);
    }
  }
}

// ---- point sets of every location ---------------------------------------------------------------------------------------------
// out[b, i, :] for i in the concatenated level order.  mode 0: offset_to_pts -- (x, y) pairs: x = pred[2k+1] * stride + cx,
// y = pred[2k] * stride + cy; mode 1: the refine-stage proposals of loss(): element 2k = cx + pred[2k] * stride, element
// 2k+1 = cy + pred[2k+1] * stride (no swap).  Centres: (w * stride, h * stride) of the location (PointGenerator.grid_points).
__global__ void points_from_offsets_kernel(const LevelTable T, int mode, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (b, i, k): one point (two floats) per thread
  const int K = T.C / 2;
  const long total = (long)T.B * T.N * K;
  if (t >= total) return;
  const int k = (int)(t % K);
  const long bi = t / K;
  const int i = (int)(bi % T.N), b = (int)(bi / T.N);
  const int l = level_of(T, i);
  const int loc = i - T.first[l];
  const float s = T.stride[l];
  const float cx = (float)(loc % T.width[l]) * s, cy = (float)(loc / T.width[l]) * s;
  const float* p = T.ptr[l] + ((size_t)b * T.C + 2 * k) * T.hw[l] + loc;
  const float a = p[0], c = p[T.hw[l]];                              // channels 2k (y offset) and 2k+1 (x offset)
  float2 o;
  if (mode == 0) { o.x = c * s + cx; o.y = a * s + cy; }
  else { o.x = cx + a * s; o.y = cy + c * s; }
  *reinterpret_cast<float2*>(out + (size_t)bi * T.C + 2 * k) = o;
}

// ---- rows at selected locations ------------------------------------------------------------------------------------------------
// idx[p] = b * N + i.  mode 0: raw channels -> out[p, c]; mode 1: image-space point sets (offset_to_pts) -> out[p, 2k] = x, [2k+1] = y
__global__ void gather_levels_kernel(const LevelTable T, const int64_t* __restrict__ idx, int P, int mode, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (p, c)
  if (t >= (long)P * T.C) return;
  const int c = (int)(t % T.C), p = (int)(t / T.C);
  const long g = idx[p];
  const int i = (int)(g % T.N), b = (int)(g / T.N);
  const int l = level_of(T, i);
  const int loc = i - T.first[l];
  if (mode == 0) {
    out[t] = T.ptr[l][((size_t)b * T.C + c) * T.hw[l] + loc];
  } else {
    const float s = T.stride[l];
    const int src = c ^ 1;                                           // x (even output) comes from channel 2k+1, y from 2k
    const float v = T.ptr[l][((size_t)b * T.C + src) * T.hw[l] + loc];
    const float ctr = (c & 1) ? (float)(loc / T.width[l]) * s : (float)(loc % T.width[l]) * s;
    out[t] = v * s + ctr;
  }
}
// grad[level][b, c', loc] = grad_out[p, c] (x stride in mode 1); the level gradients are zero-filled by the caller and the
// selected locations are distinct, so plain stores suffice
__global__ void gather_levels_backward_kernel(const LevelTable T, const int64_t* __restrict__ idx, int P, int mode,
                                              const float* __restrict__ grad_out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)P * T.C) return;
  const int c = (int)(t % T.C), p = (int)(t / T.C);
  const long g = idx[p];
  const int i = (int)(g % T.N), b = (int)(g / T.N);
  const int l = level_of(T, i);
  const int loc = i - T.first[l];
  if (mode == 0) T.grad[l][((size_t)b * T.C + c) * T.hw[l] + loc] = grad_out[t];
  else T.grad[l][((size_t)b * T.C + (c ^ 1)) * T.hw[l] + loc] = grad_out[t] * T.stride[l];
}

// ---- outline samples ------------------------------------------------------------------------------------------------------------
// corners [P, 8] -> out [P, 4 * n, 2]: on edge e (corner e -> corner e+1) the points ratio_j * next + (1 - ratio_j) * cur;
// the n ratios are the caller's torch.linspace(0, 1, n) values (the same floats the reference multiplies with)
__global__ void outline_samples_kernel(const float* __restrict__ corners, int P, int n, const float* __restrict__ ratios,
                                       float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (p, e, j)
  if (t >= (long)P * 4 * n) return;
  const int j = (int)(t % n), e = (int)((t / n) % 4);
  const long p = t / (4 * n);
  const float r = ratios[j];
  const float* q = corners + p * 8;
  const float cx = q[2 * e], cy = q[2 * e + 1], nx = q[2 * ((e + 1) & 3)], ny = q[2 * ((e + 1) & 3) + 1];
  float2 o;
  o.x = r * nx + (1.f - r) * cx;
  o.y = r * ny + (1.f - r) * cy;
  *reinterpret_cast<float2*>(out + t * 2) = o;
}

int fill_table(const orp_level_desc* lv, int nlevels, int batch, int channels, LevelTable& T) {
  if (!lv || nlevels <= 0 || nlevels > kMaxLevels || batch <= 0 || channels <= 0) return ORP_EINVAL;
  T.nlev = nlevels; T.B = batch; T.C = channels;
  int n = 0;
  for (int i = 0; i < kMaxLevels; i++) {
    if (i < nlevels) {
      if (lv[i].height <= 0 || lv[i].width <= 0 || !lv[i].data) return ORP_EINVAL;
      T.ptr[i] = lv[i].data; T.grad[i] = lv[i].grad; T.hw[i] = lv[i].height * lv[i].width; T.width[i] = lv[i].width;
      T.first[i] = n; T.stride[i] = lv[i].stride;
      n += T.hw[i];
    } else {
      T.ptr[i] = nullptr; T.grad[i] = nullptr; T.hw[i] = 0; T.width[i] = 1; T.first[i] = 0x7fffffff; T.stride[i] = 1.f;
    }
  }
  T.first[kMaxLevels] = n;
  T.N = n;
  return ORP_OK;
}

}  // namespace

extern "C" {

int orp_pointset_target(const int64_t* gt_inds, const uint8_t* valid, int batch, int n, const float* gt_boxes,
                        const int64_t* gt_labels, const int32_t* gt_offset, const float* proposals, int dim,
                        float pos_weight, int64_t* labels, float* label_weights, float* rbbox_gt, float* pos_proposals,
                        float* proposal_weights, int64_t* gt_inds_out, int32_t* counts, void* stream) {
  if (batch < 0 || n < 0 || !gt_inds || !gt_offset || !labels || !label_weights || !rbbox_gt || !proposal_weights ||
      !gt_inds_out || (pos_proposals && (!proposals || dim <= 0)))
    return ORP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (counts) {
    hipError_t e = orp::fill_async(counts, 0, sizeof(int32_t) * 2 * (size_t)(batch > 0 ? batch : 1), st);
    if (e != hipSuccess) return (int)e;
  }
  const long total = (long)batch * n;
  if (total == 0) return ORP_OK;
  hipLaunchKernelGGL(pointset_target_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                     gt_inds, valid, batch, n, gt_boxes, gt_labels, gt_offset, proposals, dim, pos_weight, labels,
                     label_weights, rbbox_gt, pos_proposals, proposal_weights, gt_inds_out, counts);
  return done();
}

int orp_points_from_offsets(const orp_level_desc* levels_host, int nlevels, int batch, int channels, int mode, float* out,
                            void* stream) {
  LevelTable T;
  int rc = fill_table(levels_host, nlevels, batch, channels, T);
  if (rc != ORP_OK) return rc;
  if (!out || (channels & 1) || (mode != 0 && mode != 1)) return ORP_EINVAL;
  const long total = (long)batch * T.N * (channels / 2);
  hipLaunchKernelGGL(points_from_offsets_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, T, mode, out);
  return done();
}

int orp_gather_levels(const orp_level_desc* levels_host, int nlevels, int batch, int channels, const int64_t* index, int p,
                      int mode, float* out, void* stream) {
  LevelTable T;
  int rc = fill_table(levels_host, nlevels, batch, channels, T);
  if (rc != ORP_OK) return rc;
  if (p < 0 || (p > 0 && (!index || !out)) || (mode != 0 && mode != 1) || (mode == 1 && (channels & 1))) return ORP_EINVAL;
  if (p == 0) return ORP_OK;
  const long total = (long)p * channels;
  hipLaunchKernelGGL(gather_levels_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, T, index, p, mode, out);
  return done();
}

int orp_gather_levels_backward(const orp_level_desc* levels_host, int nlevels, int batch, int channels, const int64_t* index,
                               int p, int mode, const float* grad_out, void* stream) {
  LevelTable T;
  int rc = fill_table(levels_host, nlevels, batch, channels, T);
  if (rc != ORP_OK) return rc;
  if (p < 0 || (p > 0 && (!index || !grad_out)) || (mode != 0 && mode != 1)) return ORP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < nlevels; i++) {
    if (!levels_host[i].grad) return ORP_EINVAL;
    hipError_t e = orp::fill_async(levels_host[i].grad, 0, sizeof(float) * (size_t)batch * channels * T.hw[i], st);
    if (e != hipSuccess) return (int)e;
  }
  if (p == 0) return ORP_OK;
  const long total = (long)p * channels;
  hipLaunchKernelGGL(gather_levels_backward_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                     T, index, p, mode, grad_out);
  return done();
}

int orp_outline_samples(const float* corners, int p, int n, const float* ratios, float* out, void* stream) {
  if (p < 0 || n < 2 || !ratios || (p > 0 && (!corners || !out))) return ORP_EINVAL;
  if (p == 0) return ORP_OK;
  const long total = (long)p * 4 * n;
  hipLaunchKernelGGL(outline_samples_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, corners, p, n, ratios, out);
  return done();
}

}  // extern "C"
