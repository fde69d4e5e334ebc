// orp_assign.hip -- the assignment side of the APAA training path on gfx950: everything the reference does with
// Python loops over ground truths (hundreds of tiny launches + host syncs per image) as a handful of stream-ordered
// kernels.
//
//   orp_point_assign       PointAssigner.assign            mmdet/core/bbox/assigners/point_assigner.py:22-145
//   orp_max_iou_assign     MaxIoUAssigner.assign_wrt_overlaps   mmdet/core/bbox/assigners/max_iou_assigner.py:88-152
//   orp_apaa_feature_dissimilarity   get_adaptive_points_feature + feature_cosine_similarity
//                                    mmdet/models/anchor_heads/orientedreppoints_head.py:495-520, 576-600
//                                    (positives only: the reference samples ALL N x 9 points -> 201 MB / image)
//   orp_apaa_select        point_samples_selection          orientedreppoints_head.py:602-671
//
// Tie rules the reference leaves to torch.topk / torch.sort / torch.max are fixed here as: smaller index first.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"

namespace {

typedef unsigned long long u64;
// Total order of the quality values in the FINAL ranking of orp_apaa_select (both formulations below use it, so a gt with more
// than kSelCap positives and one with fewer cannot rank a NaN differently -- round-5 advisor): the float order for ordinary values
// (+0 == -0), NaN above everything (where torch.sort, which the reference ranks with, puts it).
__device__ __forceinline__ unsigned apaa_rank_key(float v) {
  if (v != v) return 0xffffffffu;
  if (v == 0.f) v = 0.f;
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
constexpr int kThreads = 256;

__device__ __forceinline__ u64 pack_key(float v, int idx) {
  // v >= 0: IEEE bits are order preserving; (value, index) lexicographic in one 64-bit word
  return ((u64)__float_as_uint(v) << 32) | (u64)(unsigned)idx;
}

// block-wide minimum of a u64 key (all threads get the result)
__device__ __forceinline__ u64 block_min_u64(u64 v, u64* sbuf) {
  for (int off = 32; off > 0; off >>= 1) {
    u64 o = __shfl_down(v, off, 64);
    v = o < v ? o : v;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sbuf[wave] = v;
  __syncthreads();
  u64 r = sbuf[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = sbuf[i] < r ? sbuf[i] : r;
  return r;
}

// ---- PointAssigner -----------------------------------------------------------------------------------------------
__global__ void init_minmax_kernel(int* minmax) { minmax[0] = 0x7fffffff; minmax[1] = -0x7fffffff; }

__global__ void level_minmax_kernel(const float* __restrict__ points, int n, int* __restrict__ minmax) {
  int lo = 0x7fffffff, hi = -0x7fffffff;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int l = (int)log2f(points[3 * i + 2]);
    lo = min(lo, l); hi = max(hi, l);
  }
  atomicMin(&minmax[0], lo);
  atomicMax(&minmax[1], hi);
}

__global__ void __launch_bounds__(kThreads)
point_assign_gt_kernel(const float* __restrict__ points, int n, const float* __restrict__ gts, int k, float scale,
                       int pos_num, const int* __restrict__ minmax, u64* __restrict__ best) {
  __shared__ u64 sbuf[kThreads / 64];
  const int g = blockIdx.x;
  const float* q = gts + (size_t)g * 8;
  float xmin = q[0], xmax = q[0], ymin = q[1], ymax = q[1];
#pragma unroll
  for (int t = 1; t < 4; t++) {
    xmin = fminf(xmin, q[2 * t]); xmax = fmaxf(xmax, q[2 * t]);
    ymin = fminf(ymin, q[2 * t + 1]); ymax = fmaxf(ymax, q[2 * t + 1]);
  }
  const float cx = (xmin + xmax) / 2, cy = (ymin + ymax) / 2;
  const float w = fmaxf(xmax - xmin, 1e-6f), h = fmaxf(ymax - ymin, 1e-6f);
  int lvl = (int)((log2f(w / scale) + log2f(h / scale)) / 2);
  lvl = max(minmax[0], min(minmax[1], lvl));
  u64 last = 0;                     // keys are strictly increasing across the pos_num rounds
  bool first = true;
  for (int r = 0; r < pos_num; r++) {
    u64 mine = ~0ull;
    for (int i = threadIdx.x; i < n; i += kThreads) {
      const float s = points[3 * i + 2];
      if ((int)log2f(s) != lvl) continue;
      const float dx = (points[3 * i] - cx) / w, dy = (points[3 * i + 1] - cy) / h;
      const float d = sqrtf(dx * dx + dy * dy);
      if (!(d >= 0.f)) continue;    // NaN
      const u64 key = pack_key(d, i);
      if ((first || key > last) && key < mine) mine = key;
    }
    const u64 sel = block_min_u64(mine, sbuf);
    if (sel == ~0ull) break;
    if (threadIdx.x == 0) {
      const int pi = (int)(unsigned)(sel & 0xffffffffu);
      // (distance, gt index): smaller distance wins, equal distance -> the EARLIER gt keeps the point (strict <)
      atomicMin(&best[pi], ((sel >> 32) << 32) | (u64)(unsigned)g);
    }
    last = sel; first = false;
  }
}

__global__ void point_assign_finish_kernel(const u64* __restrict__ best, int n, int64_t* __restrict__ gt_inds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gt_inds[i] = (best[i] == ~0ull) ? 0 : (int64_t)(best[i] & 0xffffffffu) + 1;
}

// ---- MaxIoUAssigner ------------------------------------------------------------------------------------------------
// overlaps are point-major [n, k] (what convex_iou produces).  torch.max semantics: NaN wins, first index on ties.
__device__ __forceinline__ bool better(float v, float cur) { return (v > cur) || (v != v && cur == cur); }

__global__ void __launch_bounds__(kThreads)
gt_max_kernel(const float* __restrict__ ov, int n, int k, float* __restrict__ gt_max) {
  __shared__ float s[kThreads];
  const int g = blockIdx.x;
  float m = -INFINITY;
  bool any = false;
  for (int i = threadIdx.x; i < n; i += kThreads) {
    const float v = ov[(size_t)i * k + g];
    if (!any || better(v, m)) { m = v; any = true; }
  }
  s[threadIdx.x] = any ? m : -INFINITY;
  __syncthreads();
  for (int off = kThreads / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) { const float o = s[threadIdx.x + off]; if (better(o, s[threadIdx.x])) s[threadIdx.x] = o; }
    __syncthreads();
  }
  if (threadIdx.x == 0) gt_max[g] = s[0];
}

// k <= 256 (the usual case): the matrix is read ONCE, row-major -- thread t takes column t % k of rows t / k, t / k + 256 / k, ... of
// its block's row range (consecutive threads read consecutive floats), the block's per-column maxima go to partial[block][k];
// gt_max_finish_kernel folds the blocks.  (One workgroup per gt reading its column at stride k touched every cache line of the
// matrix k times: 43 us for 21 824 x 64 at 0.6 % VALU busy.)
constexpr int kGtMaxBlocks = 64;
__global__ void __launch_bounds__(kThreads)
gt_max_rows_kernel(const float* __restrict__ ov, int n, int k, float* __restrict__ partial) {
  __shared__ float s[kThreads];
  const int rpp = kThreads / k;                                   // rows per pass
  const int col = threadIdx.x % k, rsub = threadIdx.x / k;
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(n, r0 + per);
  float m = -INFINITY;
  bool any = false;
  if (rsub < rpp) {
#pragma unroll 8
    for (int r = r0 + rsub; r < r1; r += rpp) {                   // (independent loads: unrolled, eight in flight per thread)
      const float v = ov[(size_t)r * k + col];
      if (!any || better(v, m)) { m = v; any = true; }
    }
  }
  s[threadIdx.x] = any ? m : -INFINITY;
  __syncthreads();
  if (threadIdx.x < k) {
    float b = s[threadIdx.x];
    for (int j = 1; j < rpp; j++) { const float o = s[j * k + threadIdx.x]; if (better(o, b)) b = o; }
    partial[(size_t)blockIdx.x * k + threadIdx.x] = b;
  }
}
__global__ void __launch_bounds__(kThreads)
gt_max_finish_kernel(const float* __restrict__ partial, int nblk, int k, float* __restrict__ gt_max) {
  __shared__ float s[kThreads];
  const int rpp = kThreads / k, col = threadIdx.x % k, rsub = threadIdx.x / k;
  float b = -INFINITY;
  if (rsub < rpp) {
#pragma unroll 8
    for (int j = rsub; j < nblk; j += rpp) { const float o = partial[(size_t)j * k + col]; if (better(o, b)) b = o; }
  }
  s[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x < k) {
    float m = s[threadIdx.x];
    for (int j = 1; j < rpp; j++) { const float o = s[j * k + threadIdx.x]; if (better(o, m)) m = o; }
    gt_max[threadIdx.x] = m;
  }
}

__global__ void max_iou_assign_kernel(const float* __restrict__ ov, int n, int k, const float* __restrict__ gt_max,
                                      float pos_thr, float neg_lo, float neg_hi, float min_pos_iou, int assign_all,
                                      int64_t* __restrict__ gt_inds, float* __restrict__ max_overlaps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* row = ov + (size_t)i * k;
  float m = row[0]; int arg = 0;
  for (int g = 1; g < k; g++) { const float v = row[g]; if (better(v, m)) { m = v; arg = g; } }
  int64_t a = -1;
  if (m >= neg_lo && m < neg_hi) a = 0;
  if (m >= pos_thr) a = arg + 1;
  if (assign_all) {
    for (int g = 0; g < k; g++) { const float gm = gt_max[g]; if (gm >= min_pos_iou && row[g] == gm) a = g + 1; }
  }
  gt_inds[i] = a;
  if (max_overlaps) max_overlaps[i] = m;
}

// gt_max_assign_all == False: assigned[gt_argmax[g]] = g + 1 for g in order (later gt overwrites)
__global__ void gt_argmax_assign_kernel(const float* __restrict__ ov, int n, int k, const float* __restrict__ gt_max,
                                        float min_pos_iou, int64_t* __restrict__ gt_inds) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int g = 0; g < k; g++) {
    const float gm = gt_max[g];
    if (!(gm >= min_pos_iou)) continue;
    for (int i = 0; i < n; i++) if (ov[(size_t)i * k + g] == gm || (gm != gm && ov[(size_t)i * k + g] != ov[(size_t)i * k + g])) { gt_inds[i] = g + 1; break; }
  }
}

// ---- APAA: feature dissimilarity of the 9 refined points of each positive -------------------------------------------
struct FeatLevels { const float* feat[8]; int H[8], W[8]; float stride[8]; };

// one wave per positive; lanes over channels.  F.grid_sample(bilinear, zeros padding, align_corners=False) of the
// level's [B,C,H,W] map at the 9 points, then max_k (1 - cos(f_k/|f_k|_c, mean/|mean|_c)) with norms clamped at 1e-2.
__global__ void __launch_bounds__(kThreads)
feature_dissimilarity_kernel(const FeatLevels L, int C, const float* __restrict__ pts18, const int32_t* __restrict__ img,
                             const int32_t* __restrict__ lvl, int p, float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * (kThreads / 64) + wave;
  if (idx >= p) return;
  const int l = lvl[idx], b = img[idx];
  const int H = L.H[l], W = L.W[l];
  const float hh = (float)H * L.stride[l], ww = (float)W * L.stride[l];
  const float* fb = L.feat[l] + (size_t)b * C * H * W;
  float wgt[9][4]; int off[9][4];
#pragma unroll
  for (int t = 0; t < 9; t++) {
    float gx = pts18[(size_t)idx * 18 + 2 * t] / (ww / 2.f) - 1.f;
    float gy = pts18[(size_t)idx * 18 + 2 * t + 1] / (hh / 2.f) - 1.f;
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = 1.f - wx1, wy1 = iy - y0f, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    wgt[t][0] = (vx0 && vy0) ? wx0 * wy0 : 0.f; off[t][0] = (vx0 && vy0) ? y0 * W + x0 : 0;   // nw
    wgt[t][1] = (vx1 && vy0) ? wx1 * wy0 : 0.f; off[t][1] = (vx1 && vy0) ? y0 * W + x1 : 0;   // ne
    wgt[t][2] = (vx0 && vy1) ? wx0 * wy1 : 0.f; off[t][2] = (vx0 && vy1) ? y1 * W + x0 : 0;   // sw
    wgt[t][3] = (vx1 && vy1) ? wx1 * wy1 : 0.f; off[t][3] = (vx1 && vy1) ? y1 * W + x1 : 0;   // se
  }
  float nk[9], dk[9], nm = 0.f;      // |f_k|^2, f_k . mean, |mean|^2 partial sums over this lane's channels
#pragma unroll
  for (int t = 0; t < 9; t++) { nk[t] = 0.f; dk[t] = 0.f; }
  for (int c = lane; c < C; c += 64) {
    const float* fc = fb + (size_t)c * H * W;
    float f[9], mean = 0.f;
#pragma unroll
    for (int t = 0; t < 9; t++) {
      f[t] = fc[off[t][0]] * wgt[t][0] + fc[off[t][1]] * wgt[t][1] + fc[off[t][2]] * wgt[t][2] + fc[off[t][3]] * wgt[t][3];
      mean += f[t];
    }
    mean = mean / 9.f;
    nm += mean * mean;
#pragma unroll
    for (int t = 0; t < 9; t++) { nk[t] += f[t] * f[t]; dk[t] += f[t] * mean; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    nm += __shfl_xor(nm, o, 64);
#pragma unroll
    for (int t = 0; t < 9; t++) { nk[t] += __shfl_xor(nk[t], o, 64); dk[t] += __shfl_xor(dk[t], o, 64); }
  }
  if (lane == 0) {
    const float norm_m = sqrtf(nm), cm = fmaxf(norm_m, 1e-2f);
    float worst = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const float norm_k = sqrtf(nk[t]), ck = fmaxf(norm_k, 1e-2f);
      // unit vectors u = f_k / ck, v = mean / cm;  CosineSimilarity(eps = 1e-6): u.v / (max(|u|,eps) * max(|v|,eps))
      const float uv = dk[t] / (ck * cm), nu = norm_k / ck, nv = norm_m / cm;
      const float cs = uv / (fmaxf(nu, 1e-6f) * fmaxf(nv, 1e-6f));
      worst = fmaxf(worst, 1.f - cs);
    }
    out[idx] = worst;
  }
}

// ---- APAA: per-gt sample selection -------------------------------------------------------------------------------------
// one workgroup per gt: per level the <= per_level_k smallest-Q positives -> merge -> ascending -> keep ceil(ratio * n).
// ONE pass over the positives collects the gt's own (a few dozen of several thousand) in LDS; the per-level selection and the final
// order are then ranks by counting -- every item against every other, all threads at once -- instead of per_level_k * num_level
// sequential scans of the whole array with a block-wide minimum each (199 us at 0.2 waves per SIMD: pure latency).  The order is
// the reference's: inside a level ascending (Q, index) -- torch.topk(largest=False) on distinct keys --, levels concatenated, then
// a STABLE sort by Q (ties keep the concatenation order), i.e. ascending (Q, level, index).  A gt with more positives than the
// LDS list holds takes the sequential form (same results).
constexpr int kSelCap = 1024;
__global__ void __launch_bounds__(kThreads)
apaa_select_kernel(const float* __restrict__ q, const int64_t* __restrict__ pos_gt, const int32_t* __restrict__ pos_lvl,
                   int p, int num_level, int per_level_k, double top_ratio, uint8_t* __restrict__ keep) {
  __shared__ u64 sbuf[kThreads / 64];
  __shared__ float cand_q[64];
  __shared__ int cand_i[64];
  __shared__ int ncand;
  __shared__ u64 it_key[kSelCap];                // (order-preserving Q bits, index)
  __shared__ unsigned char it_lvl[kSelCap];
  __shared__ int nitem, ncand2;
  __shared__ u64 c_key[64];
  __shared__ unsigned char c_lvl[64];
  const int g = blockIdx.x + 1;
  if (threadIdx.x == 0) { ncand = 0; nitem = 0; ncand2 = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < p; i += kThreads) {
    if (pos_gt[i] != g) continue;
    const int lv = pos_lvl[i];
    if (lv < 0 || lv >= num_level) continue;
    const int slot = atomicAdd(&nitem, 1);
    if (slot < kSelCap) {
      // Q is a sum of non-negative losses; map to an order-preserving unsigned key for any sign anyway
      unsigned u = __float_as_uint(q[i]); u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      it_key[slot] = ((u64)u << 32) | (u64)(unsigned)i;
      it_lvl[slot] = (unsigned char)lv;
    }
  }
  __syncthreads();
  const int m = nitem;
  if (m <= kSelCap && num_level <= 255) {
    // (1) per level: rank among the level's items (keys are distinct: they carry the index); rank < per_level_k -> candidate
    for (int a = threadIdx.x; a < m; a += kThreads) {
      const u64 ka = it_key[a]; const int la = it_lvl[a];
      int rank = 0;
      for (int b = 0; b < m; b++) rank += (it_lvl[b] == la && it_key[b] < ka) ? 1 : 0;
      if (rank < per_level_k) {
        const int c = atomicAdd(&ncand2, 1);
        if (c < 64) { c_key[c] = ka; c_lvl[c] = (unsigned char)la; }
      }
    }
    __syncthreads();
    const int n = min(ncand2, 64);                // (the sequential form also stops at 64 candidates: per_level_k * levels <= 64)
    if (ncand2 <= 64) {
      if (n < 2) {
        if ((int)threadIdx.x < n) keep[(int)(unsigned)(c_key[threadIdx.x] & 0xffffffffu)] = 1;
      } else if ((int)threadIdx.x < n) {
        // (2) position in the stable sort by Q of the level-major, key-ascending concatenation = rank under (Q, level, key)
        const u64 ka = c_key[threadIdx.x]; const int la = c_lvl[threadIdx.x];
        const int ia = (int)(unsigned)(ka & 0xffffffffu);
        const float qa = q[ia];
        int pos = 0;
        for (int b = 0; b < n; b++) {
          const u64 kb = c_key[b]; const int lb = c_lvl[b];
          const float qb = q[(int)(unsigned)(kb & 0xffffffffu)];
          const unsigned ob = apaa_rank_key(qb), oa = apaa_rank_key(qa);
          const bool before = (ob < oa) || (ob == oa && (lb < la || (lb == la && kb < ka)));
          pos += before ? 1 : 0;
        }
        const int topk = (int)ceil((double)n * top_ratio);
        if (pos < topk) keep[ia] = 1;
      }
      return;
    }
  }
  __syncthreads();
  for (int lv = 0; lv < num_level; lv++) {
    u64 last = 0; bool first = true;
    for (int r = 0; r < per_level_k; r++) {
      u64 mine = ~0ull;
      for (int i = threadIdx.x; i < p; i += kThreads) {
        if (pos_gt[i] != g || pos_lvl[i] != lv) continue;
        const float v = q[i];
        unsigned u = __float_as_uint(v); u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        const u64 key = ((u64)u << 32) | (u64)(unsigned)i;
        if ((first || key > last) && key < mine) mine = key;
      }
      const u64 sel = block_min_u64(mine, sbuf);
      if (sel == ~0ull) break;
      if (threadIdx.x == 0 && ncand < 64) {
        const int i = (int)(unsigned)(sel & 0xffffffffu);
        cand_q[ncand] = q[i]; cand_i[ncand] = i; ncand++;
      }
      last = sel; first = false;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = ncand;
    if (n < 2) {
      for (int a = 0; a < n; a++) keep[cand_i[a]] = 1;
    } else {
      // stable insertion sort by Q ascending (ties keep concatenation order)
      for (int a = 1; a < n; a++) {
        const float vq = cand_q[a]; const int vi = cand_i[a];
        int b = a - 1;
        while (b >= 0 && apaa_rank_key(cand_q[b]) > apaa_rank_key(vq)) { cand_q[b + 1] = cand_q[b]; cand_i[b + 1] = cand_i[b]; b--; }
        cand_q[b + 1] = vq; cand_i[b + 1] = vi;
      }
      const int topk = (int)ceil((double)n * top_ratio);
      for (int a = 0; a < topk && a < n; a++) keep[cand_i[a]] = 1;
    }
  }
}

inline int done() { hipError_t e = hipGetLastError(); return e == hipSuccess ? ORP_OK : (int)e; }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

extern "C" {

size_t orp_point_assign_workspace_bytes(int n) { return align256(sizeof(u64) * (size_t)(n > 0 ? n : 1)) + 256; }

int orp_point_assign(const float* points, int n, const float* gts, int k, float scale, int pos_num, int64_t* gt_inds,
                     void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || k < 0 || pos_num < 1 || (!gt_inds && n > 0)) return ORP_EINVAL;
  if (n == 0) return ORP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0) { hipError_t e = orp::fill_async(gt_inds, 0, sizeof(int64_t) * (size_t)n, st); return e == hipSuccess ? ORP_OK : (int)e; }
  if (!points || !gts) return ORP_EINVAL;
  if (!workspace || workspace_bytes < orp_point_assign_workspace_bytes(n)) return ORP_EWORKSPACE;
  u64* best = reinterpret_cast<u64*>(workspace);
  int* minmax = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + align256(sizeof(u64) * (size_t)n));
  hipError_t e = orp::fill_async(best, 0xff, sizeof(u64) * (size_t)n, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(init_minmax_kernel, dim3(1), dim3(1), 0, st, minmax);
  hipLaunchKernelGGL(level_minmax_kernel, dim3(64), dim3(kThreads), 0, st, points, n, minmax);
  hipLaunchKernelGGL(point_assign_gt_kernel, dim3(k), dim3(kThreads), 0, st, points, n, gts, k, scale, pos_num, minmax,
                     best);
  hipLaunchKernelGGL(point_assign_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, best, n, gt_inds);
  return done();
}

size_t orp_max_iou_assign_workspace_bytes(int k) {
  const size_t kk = (size_t)(k > 0 ? k : 1);
  return align256(sizeof(float) * kk) + align256(sizeof(float) * kk * kGtMaxBlocks);      // gt_max | per-block partial maxima
}

int orp_max_iou_assign(const float* overlaps_nk, int n, int k, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
                       float min_pos_iou, int gt_max_assign_all, int64_t* gt_inds, float* max_overlaps,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || k < 0 || (n > 0 && !gt_inds)) return ORP_EINVAL;
  if (n == 0) return ORP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0) {      // no gt: everything background, max_overlaps = 0 (max_iou_assigner.py:103-118)
    hipError_t e = orp::fill_async(gt_inds, 0, sizeof(int64_t) * (size_t)n, st);
    if (e == hipSuccess && max_overlaps) e = orp::fill_async(max_overlaps, 0, sizeof(float) * (size_t)n, st);
    return e == hipSuccess ? ORP_OK : (int)e;
  }
  if (!overlaps_nk) return ORP_EINVAL;
  if (!workspace || workspace_bytes < orp_max_iou_assign_workspace_bytes(k)) return ORP_EWORKSPACE;
  float* gt_max = reinterpret_cast<float*>(workspace);
  if (k <= kThreads) {
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align256(sizeof(float) * (size_t)k));
    const int nblk = n < kGtMaxBlocks * 64 ? (n + 63) / 64 : kGtMaxBlocks;
    hipLaunchKernelGGL(gt_max_rows_kernel, dim3(nblk), dim3(kThreads), 0, st, overlaps_nk, n, k, partial);
    hipLaunchKernelGGL(gt_max_finish_kernel, dim3(1), dim3(kThreads), 0, st, partial, nblk, k, gt_max);
  } else {
    hipLaunchKernelGGL(gt_max_kernel, dim3(k), dim3(kThreads), 0, st, overlaps_nk, n, k, gt_max);
  }
  hipLaunchKernelGGL(max_iou_assign_kernel, dim3((n + 255) / 256), dim3(256), 0, st, overlaps_nk, n, k, gt_max,
                     pos_iou_thr, neg_iou_lo, neg_iou_hi, min_pos_iou, gt_max_assign_all, gt_inds, max_overlaps);
  if (!gt_max_assign_all)
    hipLaunchKernelGGL(gt_argmax_assign_kernel, dim3(1), dim3(64), 0, st, overlaps_nk, n, k, gt_max, min_pos_iou, gt_inds);
  return done();
}

int orp_apaa_feature_dissimilarity(const float* const* feats_host, const int* heights_host, const int* widths_host,
                                   const float* strides_host, int num_levels, int channels, const float* pts18,
                                   const int32_t* img_index, const int32_t* level_index, int p, float* out,
                                   void* stream) {
  if (p < 0 || num_levels < 1 || num_levels > 8 || channels < 1) return ORP_EINVAL;
  if (p == 0) return ORP_OK;
  if (!feats_host || !heights_host || !widths_host || !strides_host || !pts18 || !img_index || !level_index || !out)
    return ORP_EINVAL;
  FeatLevels L;
  for (int i = 0; i < 8; i++) {
    const int j = i < num_levels ? i : 0;
    L.feat[i] = feats_host[j]; L.H[i] = heights_host[j]; L.W[i] = widths_host[j]; L.stride[i] = strides_host[j];
  }
  hipLaunchKernelGGL(feature_dissimilarity_kernel, dim3((p + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream, L,
                     channels, pts18, img_index, level_index, p, out);
  return done();
}

int orp_apaa_select(const float* quality, const int64_t* pos_gt_inds, const int32_t* pos_level, int p, int num_gt,
                    int num_level, int per_level_topk, double top_ratio, uint8_t* keep, void* stream) {
  if (p < 0 || num_gt < 0 || num_level < 1 || per_level_topk < 1 || per_level_topk * num_level > 64) return ORP_EINVAL;
  if (p == 0) return ORP_OK;
  if (!quality || !pos_gt_inds || !pos_level || !keep) return ORP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = orp::fill_async(keep, 0, (size_t)p, st);
  if (e != hipSuccess) return (int)e;
  if (num_gt == 0) return ORP_OK;
  hipLaunchKernelGGL(apaa_select_kernel, dim3(num_gt), dim3(kThreads), 0, st, quality, pos_gt_inds, pos_level, p,
                     num_level, per_level_topk, top_ratio, keep);
  return done();
}

}  // extern "C"
