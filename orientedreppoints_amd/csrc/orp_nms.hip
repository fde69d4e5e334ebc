// orp_nms.hip -- rotated / polygon NMS for gfx950 (MI355X), fully on device.
//
// Replaces (reference = LiWentomng/OrientedRepPoints):
//   mmdet/ops/nms/src/rnms_kernel.cu:149-265   rnms_kernel + rnms_cuda   (device mask, D2H, SERIAL HOST sweep)
//   DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:214-329  poly_nms_kernel + _poly_nms
//
// MI355X design (not a translation of the 64-thread CUDA tiling):
//   1. stable radix sort of (segment, score desc) keys (rocPRIM), gather boxes into an 8-float row layout;
//   2. mask kernel: one WAVE owns R rows x 64 columns of an upper-triangular tile; the 64 lanes hold the 64
//      column boxes in registers, the row box is wave-uniform (scalar loads), and the 64-bit suppression word
//      of a row is ONE wavefront ballot -- no LDS box tile, no atomics, lower-triangle tiles never launched
//      past an early exit.  Clipping scratch is a per-lane LDS column (orp_geom.hpp), not 8 KB of private stack;
//   3. sweep kernel: one workgroup per segment walks the 64-row blocks: a scalar (readlane) pass resolves the
//      diagonal word, then all 1024 lanes OR the kept rows' words into the LDS `removed` bitmap; the same
//      kernel scatters keep flags back to original indices and compacts them in ascending order (ballot-free
//      popcount scan), so the host never sees the mask.
// The IoU arithmetic is bit-identical to the reference's fp32 devrIoU / devPolyIoU (see orp_geom.hpp).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"
#include "orp_prof.hpp"

namespace {

using orp::Pt;
typedef unsigned long long u64;

constexpr int kMaskThreads = 256;   // 4 waves per workgroup
constexpr int kSweepThreads = 1024;

__device__ __forceinline__ unsigned int float_flip_desc(float f) {
  // order-preserving float -> uint map, then inverted so that an ASCENDING radix sort yields scores DESCENDING
  unsigned int u = __float_as_uint(f);
  unsigned int mask = (u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
  return ~(u ^ mask);
}

// keys[i] = (segment << 32) | flipped score ; vals[i] = i
__global__ void make_keys_kernel(const float* __restrict__ dets, int n, const int32_t* __restrict__ seg_off, int nseg,
                                 u64* __restrict__ keys, int32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // binary search the segment of row i
  int lo = 0, hi = nseg;   // seg_off[lo] <= i < seg_off[hi]
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
  keys[i] = ((u64)(unsigned)lo << 32) | (u64)float_flip_desc(dets[(size_t)i * 9 + 8]);
  vals[i] = i;
}

__global__ void iota_kernel(int32_t* v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

__global__ void set_single_segment_kernel(int32_t* seg_off, int n) { seg_off[0] = 0; seg_off[1] = n; }

// sorted[i][0..7] = dets[order[i]][0..7]
__global__ void gather_boxes_kernel(const float* __restrict__ dets, const int32_t* __restrict__ order, int n,
                                    float4* __restrict__ sorted) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* s = dets + (size_t)order[i] * 9;
  sorted[2 * i] = make_float4(s[0], s[1], s[2], s[3]);
  sorted[2 * i + 1] = make_float4(s[4], s[5], s[6], s[7]);
}

// ---- mask kernel -------------------------------------------------------------------------------------------
// grid = (max_cb, row_groups, nseg); block = 256.  Wave w of block (c, g, s) owns rows
// [g*4R + w*R, +R) of segment s against columns [64c, 64c+64).
template <bool GUARD>
__global__ void __launch_bounds__(kMaskThreads)
nms_mask_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ seg_off, int rows_per_wave,
                int mask_stride, float thr, u64* __restrict__ mask) {
  __shared__ Pt<float> scratch[2 * orp::ORP_CLIP_CAP][kMaskThreads];
  const int seg = blockIdx.z;
  const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
  const int c = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rpb = rows_per_wave * (kMaskThreads / 64);
  const int row_base = blockIdx.y * rpb;
  if (row_base >= n || c * 64 >= n) return;
  if ((row_base >> 6) > c) return;                       // lower-triangular tile: never read by the sweep

  orp::PolyLds<float> P{&scratch[0][threadIdx.x], kMaskThreads};
  orp::PolyLds<float> Q{&scratch[orp::ORP_CLIP_CAP][threadIdx.x], kMaskThreads};

  const int col = c * 64 + lane;
  float q8[8];
  if (col < n) {
    float4 a = boxes[2 * (size_t)(s0 + col)], b = boxes[2 * (size_t)(s0 + col) + 1];
    q8[0] = a.x; q8[1] = a.y; q8[2] = a.z; q8[3] = a.w; q8[4] = b.x; q8[5] = b.y; q8[6] = b.z; q8[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) q8[i] = 0.f;
  }
  const int r_first = __builtin_amdgcn_readfirstlane(row_base + wave * rows_per_wave);
  for (int rr = 0; rr < rows_per_wave; rr++) {
    const int r = r_first + rr;                          // wave-uniform
    if (r >= n) break;
    const float* rp = reinterpret_cast<const float*>(boxes + 2 * (size_t)(s0 + r));
    float p8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p8[i] = rp[i];           // uniform address -> scalar loads
    bool hit = false;
    if (col < n && col > r) {
      float iou = orp::quad_iou<float, GUARD>(P, Q, p8, q8);
      hit = iou > thr;
    }
    u64 bits = __ballot(hit);
    if (lane == 0) mask[(size_t)(s0 + r) * mask_stride + c] = bits;
  }
}

// ---- sweep + compaction kernel --------------------------------------------------------------------------------
// one workgroup per segment.  All LDS is dynamic (16-B aligned carve, cdna guide G17):
//   [0,8) kept word | [16, 16+4096) scan scratch | removed[cb] u64 | keepbits[cb] u64 | origbits[cb] u64
constexpr size_t kSweepHdr = 16 + sizeof(int) * kSweepThreads;

// exclusive prefix (over the whole block, chunk by chunk) of popcounts of words[0..nw); calls emit(i, word, offset)
template <typename Emit>
__device__ __forceinline__ int popc_scan_emit(const u64* words, int nw, int* tmp, Emit emit) {
  const int tid = threadIdx.x;
  int running = 0;
  for (int base = 0; base < nw; base += kSweepThreads) {
    const int i = base + tid;
    const int cnt = (i < nw) ? __popcll(words[i]) : 0;
    tmp[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < kSweepThreads; off <<= 1) {
      int v = (tid >= off) ? tmp[tid - off] : 0;
      __syncthreads();
      tmp[tid] += v;
      __syncthreads();
    }
    if (i < nw) emit(i, words[i], running + tmp[tid] - cnt);
    const int chunk_total = tmp[kSweepThreads - 1];
    __syncthreads();
    running += chunk_total;
  }
  return running;
}

__global__ void __launch_bounds__(kSweepThreads)
nms_sweep_kernel(const u64* __restrict__ mask, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_off,
                 int mask_stride, int order_out, int64_t* __restrict__ keep_out, int32_t* __restrict__ num_keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int seg = blockIdx.x;
  const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n <= 0) { if (tid == 0) num_keep[seg] = 0; return; }
  const int cb = (n + 63) >> 6;
  u64* s_kept = reinterpret_cast<u64*>(smem);
  int* tmp = reinterpret_cast<int*>(smem + 16);
  u64* removed = reinterpret_cast<u64*>(smem + kSweepHdr);
  u64* keepbits = removed + cb;
  u64* origbits = keepbits + cb;             // keep flags in ORIGINAL-index space (segment-local)

  for (int i = tid; i < cb; i += kSweepThreads) { removed[i] = 0; keepbits[i] = 0; origbits[i] = 0; }
  __syncthreads();

  for (int blk = 0; blk < cb; blk++) {
    if (wave == 0) {
      const int row = blk * 64 + lane;
      const u64 d = (row < n) ? mask[(size_t)(s0 + row) * mask_stride + blk] : 0ull;
      const u64 cur0 = removed[blk];
      unsigned clo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur0);
      unsigned chi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur0 >> 32));
      u64 cur = ((u64)chi << 32) | clo;
      const int valid = __builtin_amdgcn_readfirstlane(min(64, n - blk * 64));
      const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
      u64 kept = 0;
      for (int k = 0; k < valid; k++) {          // wave-uniform serial pass: scalar unit + v_readlane
        if (!((cur >> k) & 1ull)) {
          kept |= (1ull << k);
          cur |= ((u64)(unsigned)__builtin_amdgcn_readlane(dhi, k) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane(dlo, k);
        }
      }
      if (lane == 0) { *s_kept = kept; keepbits[blk] = kept; }
    }
    __syncthreads();
    const u64 kept = *s_kept;
    const int ncols = cb - (blk + 1);
    if (ncols > 0 && kept != 0) {
      // thread -> (column word, row slice): spread the <=64 kept rows over the otherwise idle lanes
      int slices = kSweepThreads / ncols; if (slices < 1) slices = 1; if (slices > 64) slices = 64;
      const int rows_per_slice = (64 + slices - 1) / slices;
      for (int w = tid; w < ncols * slices; w += kSweepThreads) {
        const int cidx = blk + 1 + (w % ncols);
        const int k0 = (w / ncols) * rows_per_slice;
        u64 acc = 0;
        for (int k = k0; k < k0 + rows_per_slice && k < 64; k++)
          if ((kept >> k) & 1ull) acc |= mask[(size_t)(s0 + blk * 64 + k) * mask_stride + cidx];
        if (acc) atomicOr(&removed[cidx], acc);
      }
    }
    __syncthreads();
  }

  int total;
  if (order_out == 1) {
    // visiting (score) order: positions -> original indices through `order`
    total = popc_scan_emit(keepbits, cb, tmp, [&](int i, u64 w, int o) {
      while (w) { int k = __ffsll((long long)w) - 1; w &= w - 1; keep_out[s0 + o++] = (int64_t)order[s0 + i * 64 + k]; }
    });
  } else {
    // ascending original index: scatter the flags into original-index space, then compact
    for (int i = tid; i < n; i += kSweepThreads) {
      if ((keepbits[i >> 6] >> (i & 63)) & 1ull) {
        const int o = order[s0 + i] - s0;
        atomicOr(&origbits[o >> 6], 1ull << (o & 63));
      }
    }
    __syncthreads();
    total = popc_scan_emit(origbits, cb, tmp, [&](int i, u64 w, int o) {
      while (w) { int k = __ffsll((long long)w) - 1; w &= w - 1; keep_out[s0 + o++] = (int64_t)(s0 + i * 64 + k); }
    });
  }
  if (tid == 0) num_keep[seg] = total;
}

// ---- host-side plumbing ----------------------------------------------------------------------------------------
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct NmsLayout {
  size_t off_seg, off_keys_in, off_keys_out, off_vals_in, off_order, off_boxes, off_mask, off_cub, cub_bytes, total;
};

NmsLayout nms_layout(int n_total, int nseg, int max_seg) {
  NmsLayout L;
  size_t o = 0;
  const size_t n = (size_t)(n_total > 0 ? n_total : 1);
  const size_t cb = (size_t)((max_seg + 63) / 64 > 0 ? (max_seg + 63) / 64 : 1);
  L.off_seg = o; o += align256(sizeof(int32_t) * (size_t)(nseg + 1));
  L.off_keys_in = o; o += align256(sizeof(u64) * n);
  L.off_keys_out = o; o += align256(sizeof(u64) * n);
  L.off_vals_in = o; o += align256(sizeof(int32_t) * n);
  L.off_order = o; o += align256(sizeof(int32_t) * n);
  L.off_boxes = o; o += align256(sizeof(float) * 8 * n);
  L.off_mask = o; o += align256(sizeof(u64) * n * cb);
  size_t cub = 0;
  hipcub::DeviceRadixSort::SortPairs((void*)nullptr, cub, (const u64*)nullptr, (u64*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, (int)n, 0, 64, (hipStream_t)0);
  L.cub_bytes = cub;
  L.off_cub = o; o += align256(cub);
  L.total = o;
  return L;
}

int pick_rows_per_wave(int max_seg, int nseg) {
  const long cb = (max_seg + 63) / 64;
  const long tiles = cb * (cb + 1) / 2 * (nseg > 0 ? nseg : 1);
  long r = tiles * 64 / 8192;          // aim at >= 8192 waves (8 per SIMD) when the problem is big enough
  int R = 1;
  while (R * 2 <= r && R < 16) R *= 2;
  return R;
}

int launch_nms(const float* dets, int n_total, const int32_t* seg_off_dev, int nseg, int max_seg, float thr, int flavor,
               int presorted, int order_out, int64_t* keep_out, int32_t* num_keep, void* ws, size_t ws_bytes,
               hipStream_t st, bool single_segment) {
  if (n_total < 0 || nseg < 0 || (!dets && n_total > 0) || !keep_out || !num_keep) return ORP_EINVAL;
  if (flavor != 0 && flavor != 1) return ORP_EINVAL;
  if (max_seg > ORP_NMS_MAX_BOXES) return ORP_ETOOBIG;
  if (nseg == 0) return ORP_OK;
  NmsLayout L = nms_layout(n_total, nseg, max_seg);
  if (!ws || ws_bytes < L.total) return ORP_EWORKSPACE;
  char* base = reinterpret_cast<char*>(ws);
  int32_t* seg = reinterpret_cast<int32_t*>(base + L.off_seg);
  u64* keys_in = reinterpret_cast<u64*>(base + L.off_keys_in);
  u64* keys_out = reinterpret_cast<u64*>(base + L.off_keys_out);
  int32_t* vals_in = reinterpret_cast<int32_t*>(base + L.off_vals_in);
  int32_t* order = reinterpret_cast<int32_t*>(base + L.off_order);
  float4* boxes = reinterpret_cast<float4*>(base + L.off_boxes);
  u64* mask = reinterpret_cast<u64*>(base + L.off_mask);
  void* cub = base + L.off_cub;

  if (single_segment) {
    hipLaunchKernelGGL(set_single_segment_kernel, dim3(1), dim3(1), 0, st, seg, n_total);
  } else {
    hipError_t e = hipMemcpyAsync(seg, seg_off_dev, sizeof(int32_t) * (size_t)(nseg + 1), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  if (n_total == 0) {
    hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int32_t) * (size_t)nseg, st);
    return e == hipSuccess ? ORP_OK : (int)e;
  }
  const int tb = 256, nb = (n_total + tb - 1) / tb;
  if (presorted) {
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(tb), 0, st, order, n_total);
  } else {
    hipLaunchKernelGGL(make_keys_kernel, dim3(nb), dim3(tb), 0, st, dets, n_total, seg, nseg, keys_in, vals_in);
    size_t cub_bytes = L.cub_bytes;
    int end_bit = 32;
    { int s = nseg - 1; while (s > 0) { end_bit++; s >>= 1; } }
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, keys_in, keys_out, vals_in, order, n_total, 0,
                                                      end_bit, st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(gather_boxes_kernel, dim3(nb), dim3(tb), 0, st, dets, order, n_total, boxes);

  const int max_cb = (max_seg + 63) / 64;
  const int R = pick_rows_per_wave(max_seg, nseg);
  const int rpb = R * (kMaskThreads / 64);
  dim3 grid(max_cb, (max_seg + rpb - 1) / rpb, nseg);
  {
    OrpProfScope prof(ORP_PROF_NMS_MASK, st);
    if (flavor == 0)
      hipLaunchKernelGGL(nms_mask_kernel<false>, grid, dim3(kMaskThreads), 0, st, boxes, seg, R, max_cb, thr, mask);
    else
      hipLaunchKernelGGL(nms_mask_kernel<true>, grid, dim3(kMaskThreads), 0, st, boxes, seg, R, max_cb, thr, mask);
  }

  const size_t smem = kSweepHdr + (size_t)max_cb * 3 * sizeof(u64);
  {
    OrpProfScope prof(ORP_PROF_NMS_SWEEP, st);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(nseg), dim3(kSweepThreads), smem, st, mask, order, seg, max_cb, order_out,
                       keep_out, num_keep);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // namespace

extern "C" {

size_t orp_rnms_workspace_bytes(int n) { return nms_layout(n, 1, n).total; }

int orp_rnms(const float* dets, int n, float iou_thr, int flavor, int presorted, int order_out, int64_t* keep_out,
             int32_t* num_keep, void* workspace, size_t workspace_bytes, void* stream) {
  return launch_nms(dets, n, nullptr, 1, n, iou_thr, flavor, presorted, order_out, keep_out, num_keep, workspace,
                    workspace_bytes, (hipStream_t)stream, true);
}

size_t orp_rnms_batched_workspace_bytes(int n_total, int nseg, int max_seg) {
  return nms_layout(n_total, nseg, max_seg).total;
}

int orp_rnms_batched(const float* dets, int n_total, const int32_t* seg_offsets, int nseg, int max_seg, float iou_thr,
                     int flavor, int64_t* keep_out, int32_t* num_keep, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (!seg_offsets && nseg > 0) return ORP_EINVAL;
  return launch_nms(dets, n_total, seg_offsets, nseg, max_seg, iou_thr, flavor, 0, 0, keep_out, num_keep, workspace,
                    workspace_bytes, (hipStream_t)stream, false);
}

// Host-pointer API of DOTA_devkit/poly_nms_gpu/poly_nms.hpp:9-10 -- polys_host is ALREADY sorted by the caller
// (poly_nms.pyx:18-22); keep_out_host receives positions in that order.
void _poly_nms(int* keep_out_host, int* num_out_host, const float* polys_host, int polys_num, int polys_dim,
               float nms_overlap_thresh, int device_id) {
  *num_out_host = 0;
  if (polys_num <= 0) return;
  if (polys_dim != 9) { fprintf(stderr, "_poly_nms: polys_dim must be 9 (got %d)\n", polys_dim); return; }
#define ORP_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "_poly_nms: %s\n", hipGetErrorString(e_)); goto done; } } while (0)
  float* d_polys = nullptr; int64_t* d_keep = nullptr; int32_t* d_num = nullptr; void* d_ws = nullptr;
  int64_t* h_keep = nullptr;
  size_t wsb = orp_rnms_workspace_bytes(polys_num);
  int32_t h_num = 0; int rc;
  ORP_CHK(hipSetDevice(device_id));
  ORP_CHK(hipMalloc(&d_polys, sizeof(float) * 9 * (size_t)polys_num));
  ORP_CHK(hipMalloc(&d_keep, sizeof(int64_t) * (size_t)polys_num));
  ORP_CHK(hipMalloc(&d_num, sizeof(int32_t)));
  ORP_CHK(hipMalloc(&d_ws, wsb));
  ORP_CHK(hipMemcpy(d_polys, polys_host, sizeof(float) * 9 * (size_t)polys_num, hipMemcpyHostToDevice));
  rc = orp_rnms(d_polys, polys_num, nms_overlap_thresh, 1, 1, 1, d_keep, d_num, d_ws, wsb, nullptr);
  if (rc != ORP_OK) { fprintf(stderr, "_poly_nms: orp_rnms failed (%d)\n", rc); goto done; }
  ORP_CHK(hipMemcpy(&h_num, d_num, sizeof(int32_t), hipMemcpyDeviceToHost));
  h_keep = (int64_t*)malloc(sizeof(int64_t) * (size_t)(h_num > 0 ? h_num : 1));
  ORP_CHK(hipMemcpy(h_keep, d_keep, sizeof(int64_t) * (size_t)h_num, hipMemcpyDeviceToHost));
  for (int i = 0; i < h_num; i++) keep_out_host[i] = (int)h_keep[i];
  *num_out_host = h_num;
done:
  free(h_keep);
  if (d_polys) (void)hipFree(d_polys);
  if (d_keep) (void)hipFree(d_keep);
  if (d_num) (void)hipFree(d_num);
  if (d_ws) (void)hipFree(d_ws);
#undef ORP_CHK
}

}  // extern "C"
