// orp_postproc.hip -- fused test-time decode / selection / packing around the rotated NMS (gfx950).
//
// Replaces the per-level tensor-op chains of OrientedRepPointsHead.get_bboxes_single (mmdet/models/anchor_heads/
// orientedreppoints_head.py:707-779), multiclass_rnms (mmdet/core/post_processing/bbox_nms.py:93-182) and rbbox2result
// (mmdet/core/bbox/transforms.py:356-375) -- ~180 small framework kernels per image, 1.1 ms of GPU time plus launch gaps
// in the round-1 profile -- by three kernels around the existing min-area-rect and NMS kernels.  Semantics are the
// reference's, order included:
//   candidates  level-major; inside a level the top-`nms_pre` points in the order torch.topk returned them (computed by
//               the caller on the sigmoid maxima, exactly as the reference) or all points in grid order;
//   detections  every (candidate, class) pair with score > score_thr, row-major over (candidate, class) -- the order of
//               `valid_mask.nonzero()`; one NMS over coords + label * (max_coordinate + 1) (fp32, separate multiply and
//               add: this file is built with -ffp-contract=off);
//   output      rows kept by the NMS in ascending index order, or, when more than max_num survive, the max_num highest
//               scores in descending order; packed as [reppoints(18) | corners(8) | score | label].
// Everything is fixed-shape and stream-ordered: the box count lives in device memory (orp_rnms_batched reads it), so the
// whole chain is hipGraph-capturable and costs one D2H copy per image.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"

namespace {

constexpr int kMaxLevels = 8;
constexpr int kScanThreads = 1024;

struct PpParams { int off[kMaxLevels + 1]; int width[kMaxLevels]; float stride[kMaxLevels]; int nlev; };

// ---- candidates: gather the 9 refined points (y,x)-interleaved NCHW -> (x,y) rows, centres, strides, image-space points
__global__ void pp_gather_kernel(const float* __restrict__ pts_all, const int64_t* __restrict__ cand, int m0, int n,
                                 PpParams P, float* __restrict__ pts_xy, float* __restrict__ centers,
                                 float* __restrict__ strides, float* __restrict__ reppoints) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m0) return;
  const int g = (int)cand[j];
  int l = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (g >= P.off[i]) l = i;
  const int local = g - P.off[l];
  const int y = local / P.width[l], x = local - y * P.width[l];
  const float st = P.stride[l];
  const float cx = (float)x * st, cy = (float)y * st;        // PointGenerator.grid_points: arange(W) * stride
  centers[2 * j] = cx; centers[2 * j + 1] = cy;
  strides[j] = st;
#pragma unroll
  for (int t = 0; t < 9; t++) {
    const float yv = pts_all[(size_t)(2 * t) * n + g], xv = pts_all[(size_t)(2 * t + 1) * n + g];
    pts_xy[(size_t)j * 18 + 2 * t] = xv;
    pts_xy[(size_t)j * 18 + 2 * t + 1] = yv;
    reppoints[(size_t)j * 18 + 2 * t] = xv * st + cx;        // pts * stride + centre: two roundings, as two tensor ops
    reppoints[(size_t)j * 18 + 2 * t + 1] = yv * st + cy;
  }
}

// order-preserving float <-> uint map (for an atomicMax over floats of any sign)
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// max that PROPAGATES NaN like Tensor.max() does (fmaxf drops it): a NaN coordinate from a diverged model must poison
// max_coordinate -- and with it every class offset -- exactly as in the reference (bbox_nms.py:156-158)
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? __uint_as_float(0x7fc00000u) : fmaxf(a, b); }

// ---- detections, step 1 (parallel): per candidate the bit set of classes above the threshold, and the max coordinate of
// the boxes that own at least one detection (bboxes.max() of the expanded set) through one atomicMax per workgroup
__global__ void __launch_bounds__(256)
pp_flags_kernel(const float* __restrict__ sig_all, const int64_t* __restrict__ cand, int m0, int n, int num_cls,
                const float* __restrict__ boxes, float thr, unsigned* __restrict__ bits, unsigned* __restrict__ max_ord,
                int cap, float* __restrict__ dets, int32_t* __restrict__ sel_cand, int32_t* __restrict__ sel_label) {
  __shared__ float red[4];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  // every detection row starts as padding (score -inf: sorts last, never evaluated); the compaction kernel behind this
  // one overwrites the first `count` rows -- filling 8 K rows is parallel work, not something for its single workgroup
  for (int r = j; r < cap; r += gridDim.x * blockDim.x) {
#pragma unroll
    for (int k = 0; k < 8; k++) dets[(size_t)r * 9 + k] = 0.f;
    dets[(size_t)r * 9 + 8] = -INFINITY;
    sel_cand[r] = 0; sel_label[r] = 0;
  }
  float mx = -INFINITY;
  if (j < m0) {
    const int g = (int)cand[j];
    unsigned b = 0;
    for (int c = 0; c < num_cls; c++) b |= (sig_all[(size_t)c * n + g] > thr) ? (1u << c) : 0u;
    bits[j] = b;
    if (b) {
#pragma unroll
      for (int k = 0; k < 8; k++) mx = max_nan(mx, boxes[(size_t)j * 8 + k]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max_nan(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = max_nan(max_nan(red[0], red[1]), max_nan(red[2], red[3]));
    if (mx > -INFINITY || mx != mx) atomicMax(max_ord, f2ord(mx));      // +NaN maps above +inf in the ordered space
  }
}

// inclusive scan over the workgroup (wave shuffles + one LDS hop): returns the EXCLUSIVE prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  __syncthreads();                                         // wsum may still be read from the previous chunk
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; w++) { const int t = wsum[w]; if (w < wave) base += t; tot += t; }
  *total = tot;
  return base + inc - v;
}

// ---- detections, step 2 (one workgroup): ordered compaction, row-major over (candidate, class), with the class-offset
// coordinates.  dets rows >= count get score -inf (they sort last and are never evaluated).
__global__ void __launch_bounds__(kScanThreads)
pp_compact_kernel(const float* __restrict__ sig_all, const int64_t* __restrict__ cand, int m0, int n,
                  const unsigned* __restrict__ bits, const unsigned* __restrict__ max_ord,
                  const float* __restrict__ boxes, int cap, float* __restrict__ dets, int32_t* __restrict__ sel_cand,
                  int32_t* __restrict__ sel_label, int32_t* __restrict__ seg, int32_t* __restrict__ total_out, int fill_here) {
  __shared__ int wsum[kScanThreads / 64];
  const int tid = threadIdx.x;
  const float span = ord2f(*max_ord) + 1.0f;                 // max_coordinate + 1
  int running = 0;
  for (int base = 0; base < m0; base += kScanThreads) {
    const int j = base + tid;
    const unsigned b = (j < m0) ? bits[j] : 0u;
    const int cnt = __popc(b);
    int chunk_total;
    int pos = running + block_excl_scan(cnt, wsum, &chunk_total);
    if (cnt > 0) {
      const int g = (int)cand[j];
      float b8[8];
#pragma unroll
      for (int k = 0; k < 8; k++) b8[k] = boxes[(size_t)j * 8 + k];
      unsigned rem = b;
      while (rem) {
        const int c = __ffs((int)rem) - 1;
        rem &= rem - 1;
        if (pos < cap) {
          const float offs = (float)c * span;                // labels.to(bboxes) * (max_coordinate + 1)
#pragma unroll
          for (int k = 0; k < 8; k++) dets[(size_t)pos * 9 + k] = b8[k] + offs;
          dets[(size_t)pos * 9 + 8] = sig_all[(size_t)c * n + g];
          sel_cand[pos] = j; sel_label[pos] = c;
        }
        pos++;
      }
    }
    running += chunk_total;
  }
  const int count = running < cap ? running : cap;
  if (fill_here) for (int r = count + tid; r < cap; r += kScanThreads) {      // (no candidates: the flags kernel did not run)
#pragma unroll
    for (int k = 0; k < 8; k++) dets[(size_t)r * 9 + k] = 0.f;
    dets[(size_t)r * 9 + 8] = -INFINITY;
    sel_cand[r] = 0; sel_label[r] = 0;
  }
  if (tid == 0) { seg[0] = 0; seg[1] = count; total_out[0] = running; }
}

// ---- packing: NMS survivors -> [reppoints | corners | score | label] rows + (count, overflow) tail row -----------------
__global__ void pp_pack_kernel(const int64_t* __restrict__ keep, const int32_t* __restrict__ num_keep,
                               const float* __restrict__ dets, const int32_t* __restrict__ sel_cand,
                               const int32_t* __restrict__ sel_label, const float* __restrict__ boxes,
                               const float* __restrict__ reppoints, const int32_t* __restrict__ total, int cap,
                               int max_out, float* __restrict__ packed) {
  const int kn = num_keep[0];
  const int count = kn < max_out ? kn : max_out;
  const int W = 28;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    float* tail = packed + (size_t)max_out * W;
    tail[0] = (float)count;
    tail[1] = (total[0] > cap) ? 1.f : 0.f;
    for (int k = 2; k < W; k++) tail[k] = 0.f;
  }
  // rows past the count are zero
  if (i < max_out && i >= count) {
    for (int k = 0; k < W; k++) packed[(size_t)i * W + k] = 0.f;
  }
  if (i >= kn) return;
  int row = i;                                               // <= max_out survivors: ascending-index order
  const int e = (int)keep[i];
  if (kn > max_out) {
    // more survivors than max_num: the max_num highest scores in descending order (bbox_nms.py:175-180); rank by
    // counting -- O(kn^2) over <= capacity elements, and only on this (rare) branch
    const float si = dets[(size_t)e * 9 + 8];
    int rank = 0;
    for (int k = 0; k < kn; k++) {
      const float sk = dets[(size_t)keep[k] * 9 + 8];
      rank += (sk > si || (sk == si && k < i)) ? 1 : 0;
    }
    row = rank;
  }
  if (row >= max_out) return;
  const int j = sel_cand[e];
  float* o = packed + (size_t)row * W;
#pragma unroll
  for (int k = 0; k < 18; k++) o[k] = reppoints[(size_t)j * 18 + k];
#pragma unroll
  for (int k = 0; k < 8; k++) o[18 + k] = boxes[(size_t)j * 8 + k];
  o[26] = dets[(size_t)e * 9 + 8];
  o[27] = (float)sel_label[e];
}


// ---- candidate selection: per level the top-k points by the class-maximum score, in descending score order ------------
// Replaces `scores.max(dim=1)` + `max_scores.topk(nms_pre)` per level (head :730-737) -- 15 framework kernels, ~150 us per
// image (two library sorts of 16 384 / 4 096 keys to keep 2 000) -- by an exact radix SELECT (three histogram passes over
// the order-preserving integer image of the fp32 scores find the k-th key; an ordered compaction keeps everything above
// it plus the first ties by index) and a counting rank of the k survivors (descending score, ties by ascending index: what
// a stable descending sort gives; torch.topk leaves the order of exact ties unspecified).  NaN scores rank highest, as in
// torch.  Levels with at most k points pass through in grid order.
struct SelParams {
  int off[kMaxLevels + 1];       // first point of each level
  int out_off[kMaxLevels + 1];   // first candidate slot of each level
  int sel_slot[kMaxLevels];      // index among the levels that need a selection, or -1
  int blk0[kMaxLevels + 1];      // pp_keys_kernel: first block of each level
  int nlev, k;
};

__device__ __forceinline__ unsigned ord_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// all 1024 threads: the highest bin b with count(bins > b) < need <= count(bins >= b); on return need -= count(bins > b).
// Thread t owns bins 2t, 2t+1 (nbins = 2048) or bin t (nbins = 1024): suffix scan inside the wave, wave totals through LDS.
__device__ inline void find_bin(const unsigned* hist, int nbins, int& need, int& bin, int* s_out, int* wtot) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int v0 = nbins == 2048 ? (int)hist[2 * t] : (int)hist[t];
  const int v1 = nbins == 2048 ? (int)hist[2 * t + 1] : 0;
  const int pair = v0 + v1;
  int s = pair;                                                     // inclusive suffix sum over the lanes >= mine
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_down(s, o, 64); if (lane + o < 64) s += u; }
  if (lane == 0) wtot[wave] = s;
  __syncthreads();
  int above = s - pair;
  for (int w = wave + 1; w < 16; w++) above += wtot[w];
  if (above < need && need <= above + pair) {
    if (nbins == 2048) {
      if (above + v1 >= need) { s_out[0] = 2 * t + 1; s_out[1] = need - above; }
      else { s_out[0] = 2 * t; s_out[1] = need - above - v1; }
    } else { s_out[0] = t; s_out[1] = need - above; }
  }
  __syncthreads();
  bin = s_out[0]; need = s_out[1];
  __syncthreads();
}

// pass 0, all CUs: class maximum -> order-preserving key, and the level's histogram of the top 11 key bits (per-block LDS
// histogram, non-zero bins flushed with one atomic each: a scene whose scores share one coarse bin costs one atomic per block)
constexpr int kKeyThreads = 256;
__global__ void pp_zero_kernel(unsigned* p, int per_block) {        // (a kernel, not a memset node: replay-safe everywhere)
  for (int i = threadIdx.x; i < per_block; i += blockDim.x) p[(size_t)blockIdx.x * per_block + i] = 0u;
}
__global__ void __launch_bounds__(kKeyThreads)
pp_keys_kernel(const float* __restrict__ sig, int C, int N, SelParams P, unsigned* __restrict__ keys,
               unsigned* __restrict__ hist1, int64_t* __restrict__ cand) {
  __shared__ unsigned hist[2048];
  int l = 0;
  for (int i = 1; i < P.nlev; i++) if ((int)blockIdx.x >= P.blk0[i]) l = i;
  const int base = P.off[l], n = P.off[l + 1] - base;
  const int i = ((int)blockIdx.x - P.blk0[l]) * kKeyThreads + threadIdx.x;
  if (P.sel_slot[l] < 0) {                                          // every point of this level is a candidate, grid order
    if (i < n) cand[P.out_off[l] + i] = base + i;
    return;
  }
  for (int h = threadIdx.x; h < 2048; h += kKeyThreads) hist[h] = 0;
  __syncthreads();
  if (i < n) {
    float m = sig[base + i];
    for (int c = 1; c < C; c++) m = max_nan(m, sig[(size_t)c * N + base + i]);
    if (m != m) m = __uint_as_float(0x7fc00000u);                  // one NaN pattern: the key 0 stays reserved for padding
    const unsigned key = ord_key(m);
    keys[base + i] = key;
    atomicAdd(&hist[key >> 21], 1u);
  }
  __syncthreads();
  unsigned* gh = hist1 + (size_t)P.sel_slot[l] * 2048;
  for (int h = threadIdx.x; h < 2048; h += kKeyThreads) if (hist[h]) atomicAdd(&gh[h], hist[h]);
}

// one workgroup per selected level: the keys stay in registers (thread t owns the contiguous keys [t*KPT, (t+1)*KPT)), two more
// histogram passes pin the k-th key, two block scans place the survivors
constexpr int kMaxKpt = 40;                                         // n <= 40 960 points per level
template <int KPT>
__global__ void __launch_bounds__(kScanThreads)
pp_select_kernel(SelParams P, int slot0, const unsigned* __restrict__ keys, const unsigned* __restrict__ hist1,
                 unsigned* __restrict__ selkey, int* __restrict__ selidx) {
  __shared__ unsigned hist[2048];
  __shared__ int s_out[2];
  __shared__ int wsum[16];
  const int slot = slot0 + (int)blockIdx.x;
  int l = 0;
  for (int i = 0; i < P.nlev; i++) if (P.sel_slot[i] == slot) l = i;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = P.off[l], n = P.off[l + 1] - base, k = P.k;
  unsigned* sk = selkey + (size_t)slot * k;
  int* si = selidx + (size_t)slot * k;
  unsigned key[KPT];                                               // 0 = padding: below every real key (ord_key sets or flips the top bit)
  if ((KPT & 3) == 0 && (base & 3) == 0) {
#pragma unroll
    for (int j = 0; j < KPT; j += 4) {
      const int i = tid * KPT + j;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (i + 3 < n) v = *reinterpret_cast<const uint4*>(keys + base + i);
      else {
        if (i < n) v.x = keys[base + i];
        if (i + 1 < n) v.y = keys[base + i + 1];
        if (i + 2 < n) v.z = keys[base + i + 2];
      }
      key[j] = v.x; key[j + 1] = v.y; key[j + 2] = v.z; key[j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < KPT; j++) { const int i = tid * KPT + j; key[j] = i < n ? keys[base + i] : 0u; }
  }
  for (int h = tid; h < 2048; h += kScanThreads) hist[h] = hist1[(size_t)slot * 2048 + h];
  __syncthreads();
  int need = k, b1, b2, b3;
  find_bin(hist, 2048, need, b1, s_out, wsum);
  for (int h = tid; h < 2048; h += kScanThreads) hist[h] = 0;
  __syncthreads();
  {                                                                 // neighbouring points score alike: runs of one bin -> one atomic
    int run_bin = -1, run = 0;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
      const int bn = (key[j] != 0u && (int)(key[j] >> 21) == b1) ? (int)((key[j] >> 10) & 2047u) : -1;
      if (bn != run_bin) { if (run_bin >= 0) atomicAdd(&hist[run_bin], (unsigned)run); run_bin = bn; run = 0; }
      run++;
    }
    if (run_bin >= 0) atomicAdd(&hist[run_bin], (unsigned)run);
  }
  __syncthreads();
  find_bin(hist, 2048, need, b2, s_out, wsum);
  for (int h = tid; h < 1024; h += kScanThreads) hist[h] = 0;
  __syncthreads();
  const unsigned hi22 = ((unsigned)b1 << 11) | (unsigned)b2;
#pragma unroll
  for (int j = 0; j < KPT; j++)
    if (key[j] != 0u && (key[j] >> 10) == hi22) atomicAdd(&hist[key[j] & 1023u], 1u);
  __syncthreads();
  find_bin(hist, 1024, need, b3, s_out, wsum);
  const unsigned T = (hi22 << 10) | (unsigned)b3;                   // the k-th largest key; the first `need` keys == T are kept

  auto block_exclusive = [&](int v) {                               // exclusive scan of a per-thread count over the 1024 threads
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; w++) off += wsum[w];
    __syncthreads();
    return off + incl - v;
  };
  int n_eq = 0;
#pragma unroll
  for (int j = 0; j < KPT; j++) n_eq += key[j] == T;
  int eq_rank = block_exclusive(n_eq);
  int n_sel = 0;
  unsigned long long selmask = 0;
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const bool eq = key[j] == T;
    const bool sel = key[j] > T || (eq && eq_rank < need);
    eq_rank += eq;
    n_sel += sel;
    selmask |= sel ? (1ull << j) : 0ull;
  }
  int pos = block_exclusive(n_sel);
#pragma unroll
  for (int j = 0; j < KPT; j++)
    if ((selmask >> j) & 1ull) {
      if (pos < k) { sk[pos] = key[j]; si[pos] = tid * KPT + j; }
      pos++;
    }
}

// rank of each survivor among the k of its level: 64 survivors per workgroup, 16 lanes each over a sixteenth of the keys
__global__ void __launch_bounds__(kScanThreads)
pp_rank_kernel(SelParams P, const unsigned* __restrict__ selkey, const int* __restrict__ selidx, int64_t* __restrict__ cand) {
  extern __shared__ unsigned char smem_rank[];
  unsigned* lk = reinterpret_cast<unsigned*>(smem_rank);
  int* li = reinterpret_cast<int*>(lk + P.k);
  int l = 0;
  for (int i = 0; i < P.nlev; i++) if (P.sel_slot[i] == (int)blockIdx.y) l = i;
  const int k = P.k, tid = threadIdx.x;
  const unsigned* sk = selkey + (size_t)blockIdx.y * k;
  const int* si = selidx + (size_t)blockIdx.y * k;
  for (int i = tid; i < k; i += kScanThreads) { lk[i] = sk[i]; li[i] = si[i]; }
  __syncthreads();
  const int j = blockIdx.x * 64 + (tid >> 4), part = tid & 15;
  int cnt = 0;
  if (j < k) {
    const unsigned kj = lk[j];
    const int ij = li[j];
    for (int i = part; i < k; i += 16) {
      const unsigned ki = lk[i];
      cnt += (ki > kj) | ((ki == kj) & (li[i] < ij));
    }
  }
  for (int o = 8; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (j < k && part == 0) cand[P.out_off[l] + cnt] = P.off[l] + li[j];
}

inline int done() { hipError_t e = hipGetLastError(); return e == hipSuccess ? ORP_OK : (int)e; }

}  // namespace

extern "C" {

size_t orp_pp_compact_scratch_bytes(int m0) { return 256 + sizeof(unsigned) * (size_t)(m0 > 0 ? m0 : 1); }

size_t orp_pp_select_scratch_bytes(int n, int nms_pre, int nlevels) {
  const size_t nn = n > 0 ? n : 1, kk = nms_pre > 0 ? nms_pre : 1, ll = nlevels > 0 ? nlevels : 1;
  return 512 + sizeof(unsigned) * (nn + 64) + (sizeof(unsigned) + sizeof(int)) * kk * ll + sizeof(unsigned) * 2048 * ll;
}

int orp_pp_select(const float* sig_all, int num_classes, int n, const int* level_offsets_host, int nlevels, int nms_pre,
                  int64_t* cand, void* scratch, size_t scratch_bytes, void* stream) {
  if (!sig_all || !level_offsets_host || !cand || !scratch || num_classes <= 0 || n <= 0 || nlevels <= 0 ||
      nlevels > kMaxLevels || nms_pre <= 0 || nms_pre > 4096)
    return ORP_EINVAL;
  if (scratch_bytes < orp_pp_select_scratch_bytes(n, nms_pre, nlevels)) return ORP_EWORKSPACE;
  SelParams P;
  P.nlev = nlevels; P.k = nms_pre;
  int out = 0, nsel = 0, blk = 0;
  for (int i = 0; i < kMaxLevels; i++) {
    P.off[i] = i < nlevels ? level_offsets_host[i] : n;
    P.out_off[i] = out;
    P.blk0[i] = i < nlevels ? blk : 0x7fffffff;
    P.sel_slot[i] = -1;
    if (i < nlevels) {
      const int n_l = (i + 1 < nlevels ? level_offsets_host[i + 1] : n) - level_offsets_host[i];
      if (n_l < 0) return ORP_EINVAL;
      if (n_l > kMaxKpt * kScanThreads && n_l > nms_pre) return ORP_ETOOBIG;
      if (n_l > nms_pre) P.sel_slot[i] = nsel++;
      out += n_l > nms_pre ? nms_pre : n_l;
      blk += (n_l + kKeyThreads - 1) / kKeyThreads;
    }
  }
  P.off[nlevels] = n; P.off[kMaxLevels] = n; P.out_off[kMaxLevels] = out; P.blk0[kMaxLevels] = 0x7fffffff;
  for (int i = nlevels; i < kMaxLevels; i++) P.out_off[i] = out;
  unsigned* keys = reinterpret_cast<unsigned*>(scratch);
  unsigned* selkey = keys + (((size_t)n + 63) & ~(size_t)63);
  int* selidx = reinterpret_cast<int*>(selkey + (size_t)nms_pre * nlevels);
  unsigned* hist1 = reinterpret_cast<unsigned*>(selidx + (size_t)nms_pre * nlevels);
  hipStream_t st = (hipStream_t)stream;
  if (nsel > 0) hipLaunchKernelGGL(pp_zero_kernel, dim3(nsel), dim3(kScanThreads), 0, st, hist1, 2048);
  hipLaunchKernelGGL(pp_keys_kernel, dim3(blk), dim3(kKeyThreads), 0, st, sig_all, num_classes, n, P, keys, hist1, cand);
  if (nsel > 0) {
    // the register array is sized at compile time: one launch with the largest selected level's keys-per-thread count
    int kpt = 1;
    for (int i = 0; i < nlevels; i++) {
      if (P.sel_slot[i] < 0) continue;
      const int n_l = (i + 1 < nlevels ? level_offsets_host[i + 1] : n) - level_offsets_host[i];
      const int kk = (n_l + kScanThreads - 1) / kScanThreads;
      if (kk > kpt) kpt = kk;
    }
    if (kpt <= 4) hipLaunchKernelGGL(pp_select_kernel<4>, dim3(nsel), dim3(kScanThreads), 0, st, P, 0, keys, hist1, selkey, selidx);
    else if (kpt <= 8) hipLaunchKernelGGL(pp_select_kernel<8>, dim3(nsel), dim3(kScanThreads), 0, st, P, 0, keys, hist1, selkey, selidx);
    else if (kpt <= 16) hipLaunchKernelGGL(pp_select_kernel<16>, dim3(nsel), dim3(kScanThreads), 0, st, P, 0, keys, hist1, selkey, selidx);
    else if (kpt <= 24) hipLaunchKernelGGL(pp_select_kernel<24>, dim3(nsel), dim3(kScanThreads), 0, st, P, 0, keys, hist1, selkey, selidx);
    else hipLaunchKernelGGL(pp_select_kernel<kMaxKpt>, dim3(nsel), dim3(kScanThreads), 0, st, P, 0, keys, hist1, selkey, selidx);
    hipLaunchKernelGGL(pp_rank_kernel, dim3((nms_pre + 63) / 64, nsel), dim3(kScanThreads),
                       (sizeof(unsigned) + sizeof(int)) * (size_t)nms_pre, st, P, selkey, selidx, cand);
  }
  return done();
}

int orp_pp_gather(const float* pts_all, const int64_t* cand, int m0, int n, const int* level_offsets_host,
                  const int* level_widths_host, const float* level_strides_host, int nlevels, float* pts_xy,
                  float* centers, float* strides, float* reppoints, void* stream) {
  if (m0 < 0 || n <= 0 || nlevels <= 0 || nlevels > kMaxLevels || !level_offsets_host || !level_widths_host ||
      !level_strides_host)
    return ORP_EINVAL;
  if (m0 == 0) return ORP_OK;
  if (!pts_all || !cand || !pts_xy || !centers || !strides || !reppoints) return ORP_EINVAL;
  PpParams P;
  P.nlev = nlevels;
  for (int i = 0; i < kMaxLevels; i++) {
    const int k = i < nlevels ? i : nlevels - 1;
    P.off[i] = i < nlevels ? level_offsets_host[i] : 0x7fffffff;
    P.width[i] = level_widths_host[k]; P.stride[i] = level_strides_host[k];
  }
  P.off[kMaxLevels] = 0x7fffffff;
  hipLaunchKernelGGL(pp_gather_kernel, dim3((m0 + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts_all, cand, m0, n,
                     P, pts_xy, centers, strides, reppoints);
  return done();
}

int orp_pp_compact(const float* sig_all, const int64_t* cand, int m0, int n, int num_classes, const float* boxes,
                   float score_thr, int capacity, float* dets, int32_t* sel_cand, int32_t* sel_label, int32_t* seg2,
                   int32_t* total, void* scratch, size_t scratch_bytes, void* stream) {
  if (m0 < 0 || n <= 0 || num_classes <= 0 || num_classes > 32 || capacity <= 0 || !dets || !sel_cand || !sel_label ||
      !seg2 || !total)
    return ORP_EINVAL;
  if (m0 > 0 && (!sig_all || !cand || !boxes)) return ORP_EINVAL;
  if (!scratch || scratch_bytes < orp_pp_compact_scratch_bytes(m0)) return ORP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned* max_ord = reinterpret_cast<unsigned*>(scratch);
  unsigned* bits = max_ord + 64;
  // ordered encoding of -inf = ~0xff800000 = 0x007fffff
  hipError_t e = orp::fill_async(max_ord, 0, sizeof(unsigned), st);     // 0 < f2ord(x) for every float x: "no box yet"
  if (e != hipSuccess) return (int)e;
  if (m0 > 0)
    hipLaunchKernelGGL(pp_flags_kernel, dim3((m0 + 255) / 256), dim3(256), 0, st, sig_all, cand, m0, n, num_classes, boxes,
                       score_thr, bits, max_ord, capacity, dets, sel_cand, sel_label);
  hipLaunchKernelGGL(pp_compact_kernel, dim3(1), dim3(kScanThreads), 0, st, sig_all, cand, m0, n, bits, max_ord, boxes,
                     capacity, dets, sel_cand, sel_label, seg2, total, m0 > 0 ? 0 : 1);
  return done();
}

int orp_pp_pack(const int64_t* keep, const int32_t* num_keep, const float* dets, const int32_t* sel_cand,
                const int32_t* sel_label, const float* boxes, const float* reppoints, const int32_t* total, int capacity,
                int max_out, float* packed, void* stream) {
  if (capacity <= 0 || max_out <= 0 || !keep || !num_keep || !dets || !sel_cand || !sel_label || !boxes || !reppoints ||
      !total || !packed)
    return ORP_EINVAL;
  const int threads = capacity > max_out ? capacity : max_out;
  hipLaunchKernelGGL(pp_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, keep, num_keep, dets,
                     sel_cand, sel_label, boxes, reppoints, total, capacity, max_out, packed);
  return done();
}

}  // extern "C"
