// orp_nms.hip -- rotated / polygon NMS for gfx950 (MI355X), fully on device.
//
// Replaces (reference = LiWentomng/OrientedRepPoints):
//   mmdet/ops/nms/src/rnms_kernel.cu:149-265   rnms_kernel + rnms_cuda   (device mask, D2H, SERIAL HOST sweep)
//   DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:214-329  poly_nms_kernel + _poly_nms
//
// MI355X design (not a translation of the 64-thread CUDA tiling):
//   1. stable radix sort of (segment, score desc) keys (rocPRIM);
//   2. per-box pre-pass (QuadPrep: orientation, oriented origin-fan triangles, signs, |area|), then the mask
//      kernel: a workgroup owns a (<=64 rows x 64 columns) upper-triangular tile.  Phase A (lane = column, row
//      wave-uniform through scalar loads) proves for ~84 % of the pairs of a dense scene, without a division,
//      that every fan term is exactly 0 (orp_quadfast.hpp pair_is_far) and emits their bits by one wavefront
//      ballot; the remaining pairs are queued in LDS and drained in phase B as a TERM queue (orp_tile.hpp): a
//      per-term exact-zero screen drops half of their 16 fan terms, the others run the register decision tree one
//      term per lane (no per-lane polygon storage) and are summed per pair in the reference's order.  The box
//      count is read from device memory (exact or capacity callers alike): a bounded grid of workgroups loops over
//      the upper-triangular tiles of the actual count.  Non-zero mask words are also appended to a side list;
//   3. sweep kernel: one workgroup per segment.  If the side list fits in LDS (<= 8192 words) the whole greedy
//      pass runs out of LDS (per 64-row block: one readlane-based diagonal pass, one list scan, two barriers);
//      otherwise the dense pass walks the block rows with the next row's mask words prefetched.  Either way the
//      keep flags are scattered back to original indices and compacted in ascending order (popcount scan), so
//      the host never sees the mask;
//   4. fp64 instantiation of the same core for the merge NMS of the DOTA evaluation workflow (orp_poly_nms_f64).
// The IoU arithmetic is bit-identical to the reference's fp32 devrIoU / devPolyIoU (see orp_geom.hpp).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"
#include "orp_quadfast.hpp"
#include "orp_tile.hpp"
#include "orp_prof.hpp"

namespace {

using orp::Pt;
typedef unsigned long long u64;
using orp_tile::TileLds;
using orp_tile::TermLds;
using orp_tile::pack_signs;

constexpr int kMaskThreads = 256;   // 4 waves per workgroup
constexpr int kSweepThreads = 1024;
constexpr int kNzCap = 8192;        // sparse sweep: non-zero mask words kept in LDS (12 B each); more -> dense sweep

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

// prep[i] = quad_prepare(dets[order[i]][0..7]): orientation, oriented fan triangles, signs, |area| -- once per box
__global__ void prep_boxes_kernel(const float* __restrict__ dets, const int32_t* __restrict__ order, int n,
                                  orp::QuadPrep* __restrict__ prep, int* __restrict__ nz_count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && nz_count) *nz_count = 0;             // list of non-zero mask words, appended by the mask kernel
  if (i >= n) return;
  const float* s = dets + (size_t)order[i] * 9;
  float q8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) q8[k] = s[k];
  orp::QuadPrep p;
  orp::quad_prepare(q8, p);
  prep[i] = p;
}

// ---- mask kernel -------------------------------------------------------------------------------------------
// grid = (max_cb, row_groups, nseg); block = 256.  Block (c, g, s) owns the tile rows [g*RB, +RB) x columns
// [64c, 64c+64) of segment s (RB = 4 * rows_per_wave <= 64), in two phases:
//   A  lane = column, the wave's row is uniform (scalar loads of its QuadPrep): orp::pair_is_far decides -- without a
//      division -- that every fan term of the pair is exactly 0 (84 % of the pairs of a dense DOTA scene).  Resolved
//      pairs set their bit by one wavefront ballot; the others go to a workgroup queue in LDS;
//   B  the queue is drained by all 256 lanes (orp_tile::tile_drain_terms; row and column records come from the tile's
//      LDS copy): B1 one lane per pair -- per-term exact-zero screen, surviving terms appended to a term queue; B2 one
//      lane per TERM -- the register decision tree of orp_quadfast.hpp (generic polygon loop only for the ~1e-4 of
//      terms the tree does not cover); B3 one lane per pair -- ordered sum, threshold, atomicOr into the row's word.
//      Heavy work is thus packed densely into wavefronts instead of idling next to resolved pairs.
// one (rpb rows x 64 columns) tile: phase A, queue, phase B, mask words out
template <bool GUARD>
__device__ __forceinline__ void mask_tile(TileLds& T, TermLds& X, const orp::QuadPrep* __restrict__ prep, int s0, int n, int c,
                                          int row_base, int rpb, int rows_per_wave, int mask_stride, float thr,
                                          u64* __restrict__ mask, int dbg, int* __restrict__ nz_count,
                                          unsigned* __restrict__ nz_rc, u64* __restrict__ nz_w) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

  // ---- stage the tile's row / column records in LDS (phase B reads them with per-lane indices) ----------------
  const int col = c * 64 + lane;
  orp::FarCol fc;
  {
    orp::QuadPrep cp;
    if (col < n) {
      cp = prep[s0 + col];
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) { cp.ax[k] = cp.ay[k] = cp.bx[k] = cp.by[k] = cp.vx[k] = cp.vy[k] = 0.f; cp.s[k] = 0; }
      cp.area_abs = 0.f; cp.force_slow = 0; cp.mabs = 0.f;
    }
    fc = orp::far_col(cp);
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) T.colE[k][lane] = make_float4(cp.ax[k], cp.ay[k], cp.bx[k], cp.by[k]);
      T.colS[lane] = pack_signs(cp);
      T.colArea[lane] = cp.area_abs;
      X.colM[lane] = cp.mabs;
    }
    if (tid < rpb) {
      const int r = row_base + tid;
      if (r < n) {
        const orp::QuadPrep rp = prep[s0 + r];
#pragma unroll
        for (int k = 0; k < 4; k++) T.rowE[k][tid] = make_float4(rp.ax[k], rp.ay[k], rp.bx[k], rp.by[k]);
        T.rowS[tid] = pack_signs(rp);
        T.rowArea[tid] = rp.area_abs;
        X.rowM[tid] = rp.mabs;
      }
      T.words[tid] = 0ull;
    }
    if (tid == 0) T.qcount = 0;
    orp_tile::term_lds_reset(X, tid);
  }
  __syncthreads();
  const bool cslow = (T.colS[lane] >> 8) != 0;
  const float carea = T.colArea[lane];

  // ---- phase A ---------------------------------------------------------------------------------------------------
  const int rl_first = __builtin_amdgcn_readfirstlane(wave * rows_per_wave);
  for (int rr = 0; rr < rows_per_wave; rr++) {
    const int rl = rl_first + rr;                        // wave-uniform
    const int r = row_base + rl;
    if (r >= n) break;
    const orp::QuadPrep* rp = prep + (s0 + r);           // uniform address -> scalar loads
    const bool valid = (col < n) & (col > r);
    bool resolved = false;
    if (valid && !(cslow | (rp->force_slow != 0))) {
      float rvx[4], rvy[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { rvx[k] = rp->vx[k]; rvy[k] = rp->vy[k]; }
      resolved = (dbg & 2) ? true : orp::pair_is_far(rvx, rvy, rp->mabs, fc);
    }
    const bool hit0 = resolved && (orp::iou_of_zero_inter<GUARD>(rp->area_abs, carea) > thr);
    const u64 bits = __ballot(hit0);
    const bool pend = valid && !resolved;
    const u64 pmask = __ballot(pend);
    if (pmask) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&T.qcount, __popcll(pmask));
      base = __builtin_amdgcn_readfirstlane(base);
      if (pend) {
        const int pos = base + __popcll(pmask & ((1ull << lane) - 1ull));
        T.queue[pos] = (unsigned short)((rl << 6) | lane);
      }
    }
    if (lane == 0 && bits) T.words[rl] = bits;          // this wave owns row rl in phase A
  }
  __syncthreads();

  // ---- phase B: per-term screen, one surviving fan term per lane, ordered sum per pair (orp_tile.hpp) ----------
  const int nq = (dbg & 1) ? 0 : T.qcount;
  orp_tile::tile_drain_terms<GUARD>(T, X, nq, [&](int rl, int cl, float iou) {
    if (iou > thr) atomicOr(&T.words[rl], 1ull << cl);
  }, dbg);
  __syncthreads();
  if (tid < rpb && row_base + tid < n) {
    const u64 w = T.words[tid];
    mask[(size_t)(s0 + row_base + tid) * mask_stride + c] = w;
    if (nz_count && w) {                                 // sparse side list for the LDS-resident sweep (single segment)
      const int pos = atomicAdd(nz_count, 1);
      if (pos < kNzCap) { nz_rc[pos] = ((unsigned)(row_base + tid) << 11) | (unsigned)c; nz_w[pos] = w; }
    }
  }
}

// The box count is read from DEVICE memory (seg_off): the host may only know an upper bound (sync-free / hipGraph callers:
// an 8 K capacity holding 2 K boxes must not pay for 60 K empty workgroups).  A bounded grid of workgroups loops over the
// UPPER-TRIANGULAR tiles of the actual count (enumerated in closed form; lower-triangular tiles are never read by the
// sweep), which also balances the load: 117 -> 83 us at 2000 boxes against one workgroup per tile of the full grid.
template <bool GUARD>
__global__ void __launch_bounds__(kMaskThreads, 4)
nms_mask_loop_kernel(const orp::QuadPrep* __restrict__ prep, const int32_t* __restrict__ seg_off, int rows_per_wave,
                     int mask_stride, float thr, u64* __restrict__ mask, int dbg, int* __restrict__ nz_count,
                     unsigned* __restrict__ nz_rc, u64* __restrict__ nz_w) {
  __shared__ TileLds T;
  __shared__ TermLds X;
  const int seg = blockIdx.z;
  const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
  const int rpb = rows_per_wave * (kMaskThreads / 64);
  if (n <= 0) return;
  const int cbn = (n + 63) >> 6, ngroups = (n + rpb - 1) / rpb;
  // only the upper-triangular tiles exist in this enumeration: the q = 64 / rpb row groups of 64-row batch j own the
  // columns j .. cbn-1, so before(j) = q * (j * cbn - j (j - 1) / 2) tiles precede batch j (the last batch may be short)
  const int q = 64 / rpb, last = cbn - 1;
  auto before = [&](int j) { return (long)q * ((long)j * cbn - (long)j * (j - 1) / 2); };
  const long t_last = before(last);
  const long total_tiles = t_last + (ngroups - last * q);
  for (long t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    int g, c;
    if (t >= t_last) {
      g = last * q + (int)(t - t_last); c = last;
    } else {
      const double b = 2.0 * cbn + 1.0;
      int j = (int)((b - sqrt(b * b - 8.0 * (double)t / (double)q)) * 0.5);
      j = j < 0 ? 0 : (j > last - 1 ? last - 1 : j);
      while (j > 0 && before(j) > t) j--;
      while (j + 1 < last && before(j + 1) <= t) j++;
      const int r = (int)(t - before(j)), w = cbn - j;
      g = j * q + r / w; c = j + r % w;
    }
    mask_tile<GUARD>(T, X, prep, s0, n, c, g * rpb, rpb, rows_per_wave, mask_stride, thr, mask, dbg, nz_count, nz_rc, nz_w);
    __syncthreads();                                     // the LDS tile is reused by the next tile
  }
}

// ---- fp64 mask kernel: the merge NMS of DOTA_devkit/ResultMerge.py (polyiou.cpp arithmetic) -------------------------
// Same triangle-fan core instantiated in double (orp_quadfast.hpp is templated on the precision): classifier first,
// register decision tree for what it leaves, generic polygon loop as the fallback -- one pair per lane, no queue
// (merge sets are a few thousand boxes per (class, image); the CPU reference spends minutes in python here).
// Suppression follows ResultMerge.py:38 `inds = np.where(ovr <= thresh)`: a box is dropped unless iou <= thr, so a NaN
// IoU (degenerate boxes) suppresses -- the opposite of the `iou > thr` test of rnms.
__global__ void prep_boxes_f64_kernel(const double* __restrict__ dets, int n, orp::QuadPrepT<double>* __restrict__ prep) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double q8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) q8[k] = dets[(size_t)i * 9 + k];
  orp::QuadPrepT<double> p;
  orp::quad_prepare<double>(q8, p);
  prep[i] = p;
}

__global__ void __launch_bounds__(kMaskThreads)
nms_mask_f64_kernel(const orp::QuadPrepT<double>* __restrict__ prep, int n, int rows_per_wave, int mask_stride,
                    double thr, u64* __restrict__ mask) {
  const int c = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rpb = rows_per_wave * (kMaskThreads / 64);
  const int row_base = blockIdx.y * rpb;
  if (row_base >= n || c * 64 >= n) return;
  if ((row_base >> 6) > c) return;
  const int col = c * 64 + lane;
  orp::QuadPrepT<double> cp;
  if (col < n) {
    cp = prep[col];
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) { cp.ax[k] = cp.ay[k] = cp.bx[k] = cp.by[k] = cp.vx[k] = cp.vy[k] = 0.0; cp.s[k] = 0; }
    cp.area_abs = 0.0; cp.force_slow = 0; cp.mabs = 0.0; cp.pad0 = 0.0;
  }
  const int r_first = __builtin_amdgcn_readfirstlane(row_base + wave * rows_per_wave);
  for (int rr = 0; rr < rows_per_wave; rr++) {
    const int r = r_first + rr;
    if (r >= n) break;
    bool hit = false;
    if (col < n && col > r) {
      const double iou = orp::quad_iou_two_phase_t<double, false>(prep + r, &cp);
      hit = !(iou <= thr);
    }
    const u64 bits = __ballot(hit);
    if (lane == 0) mask[(size_t)r * mask_stride + c] = bits;
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
                 int mask_stride, int order_out, int64_t* __restrict__ keep_out, int32_t* __restrict__ num_keep,
                 const int* __restrict__ nz_count, const unsigned* __restrict__ nz_rc, const u64* __restrict__ nz_w) {
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

  // ---- sparse sweep: the mask of a detection scene is mostly zeros; the mask kernel appended every non-zero word to a
  // side list.  If the list fits (<= kNzCap words) the whole greedy pass runs out of LDS: per 64-row block one diagonal
  // pass (wave 0) and one scan of the list that ORs the kept rows' words into `removed` and files the NEXT block's
  // diagonal words -- two barriers and no HBM / L2 round trip per block.
  const int nnz = nz_count ? *nz_count : (kNzCap + 1);
  const bool sparse = nnz <= kNzCap;
  if (sparse) {
    u64* nzw = origbits + cb;
    unsigned* nzrc = reinterpret_cast<unsigned*>(nzw + kNzCap);
    u64* diag = reinterpret_cast<u64*>(nzrc + kNzCap);              // [2][64]
    for (int i = tid; i < nnz; i += kSweepThreads) { nzw[i] = nz_w[i]; nzrc[i] = nz_rc[i]; }
    if (tid < 128) diag[tid] = 0ull;
    __syncthreads();
    for (int i = tid; i < nnz; i += kSweepThreads) {
      const unsigned rc = nzrc[i];
      if ((rc >> 17) == 0u && (rc & 2047u) == 0u) diag[(rc >> 11) & 63u] = nzw[i];      // row block 0, column block 0
    }
    __syncthreads();
    for (int blk = 0; blk < cb; blk++) {
      if (wave == 0) {
        const u64 d = diag[(blk & 1) * 64 + lane];
        diag[(blk & 1) * 64 + lane] = 0ull;                       // this buffer is refilled for block blk + 2
        const u64 cur0 = removed[blk];
        unsigned clo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur0);
        unsigned chi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur0 >> 32));
        u64 cur = ((u64)chi << 32) | clo;
        const int valid = __builtin_amdgcn_readfirstlane(min(64, n - blk * 64));
        const u64 vmask = (valid >= 64) ? ~0ull : ((1ull << valid) - 1ull);
        const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
        u64 todo = __ballot(d != 0ull) & vmask;
        while (todo) {
          const int kk = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          if (!((cur >> kk) & 1ull))
            cur |= ((u64)(unsigned)__builtin_amdgcn_readlane(dhi, kk) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane(dlo, kk);
        }
        const u64 kept = ~cur & vmask;
        if (lane == 0) { *s_kept = kept; keepbits[blk] = kept; }
      }
      __syncthreads();
      const u64 kept = *s_kept;
      const unsigned ublk = (unsigned)blk;
      for (int i = tid; i < nnz; i += kSweepThreads) {
        const unsigned rc = nzrc[i];
        const unsigned rb = rc >> 17, cc = rc & 2047u, rl = (rc >> 11) & 63u;
        if (rb == ublk && cc > ublk && ((kept >> rl) & 1ull)) atomicOr(&removed[cc], nzw[i]);
        if (rb == ublk + 1u && cc == ublk + 1u) diag[((blk + 1) & 1) * 64 + rl] = nzw[i];
      }
      __syncthreads();
    }
  }

  // Block-row sweep, software-pipelined: at the top of iteration blk every thread ISSUES the loads of its share of
  // block-row blk (all 64 rows x the column words right of the diagonal, kept or not -- at most kPre words per thread)
  // and wave 0 the loads of the NEXT diagonal words; the HBM/L2 latency then overlaps the diagonal pass and the
  // barrier instead of following them.  The diagonal pass only visits rows whose diagonal word is non-zero (rows with
  // an empty word cannot suppress anything inside the block), which in practice is a handful per block.
  constexpr int kPre = 8;
  u64 d_next = 0ull;
  if (!sparse && wave == 0) { const int row = lane; d_next = (row < n) ? mask[(size_t)(s0 + row) * mask_stride + 0] : 0ull; }
  for (int blk = 0; !sparse && blk < cb; blk++) {
    const int ncols = cb - (blk + 1);
    int slices = 1, rows_per_slice = 64;
    if (ncols > 0) {
      slices = kSweepThreads / ncols; if (slices < 1) slices = 1; if (slices > 64) slices = 64;
      rows_per_slice = (64 + slices - 1) / slices;
    }
    // thread -> (column word, row slice) of its first work item; prefetch up to kPre rows of it
    u64 pre[kPre];
    const int w0 = tid;
    const bool has = (ncols > 0) && (w0 < ncols * slices);
    const int cidx0 = blk + 1 + (has ? (w0 % ncols) : 0);
    const int k00 = has ? (w0 / ncols) * rows_per_slice : 0;
#pragma unroll
    for (int u = 0; u < kPre; u++) {
      const int kk = k00 + u;
      const int row = blk * 64 + kk;
      pre[u] = (has && u < rows_per_slice && kk < 64 && row < n) ? mask[(size_t)(s0 + row) * mask_stride + cidx0] : 0ull;
    }
    if (wave == 0) {
      const u64 d = d_next;
      const int nrow = (blk + 1) * 64 + lane;
      d_next = (blk + 1 < cb && nrow < n) ? mask[(size_t)(s0 + nrow) * mask_stride + blk + 1] : 0ull;
      const u64 cur0 = removed[blk];
      unsigned clo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur0);
      unsigned chi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur0 >> 32));
      u64 cur = ((u64)chi << 32) | clo;
      const int valid = __builtin_amdgcn_readfirstlane(min(64, n - blk * 64));
      const u64 vmask = (valid >= 64) ? ~0ull : ((1ull << valid) - 1ull);
      const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
      u64 todo = __ballot(d != 0ull) & vmask;          // rows that can suppress inside this block
      while (todo) {                                     // wave-uniform, ascending row order
        const int kk = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        if (!((cur >> kk) & 1ull))
          cur |= ((u64)(unsigned)__builtin_amdgcn_readlane(dhi, kk) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane(dlo, kk);
      }
      const u64 kept = ~cur & vmask;
      if (lane == 0) { *s_kept = kept; keepbits[blk] = kept; }
    }
    __syncthreads();
    const u64 kept = *s_kept;
    if (ncols > 0 && kept != 0) {
      for (int w = tid; w < ncols * slices; w += kSweepThreads) {
        const int cidx = blk + 1 + (w % ncols);
        const int k0 = (w / ncols) * rows_per_slice;
        u64 acc = 0;
        if (w == w0) {
#pragma unroll
          for (int u = 0; u < kPre; u++)
            if (u < rows_per_slice && k0 + u < 64 && ((kept >> (k0 + u)) & 1ull)) acc |= pre[u];
          for (int kk = k0 + kPre; kk < k0 + rows_per_slice && kk < 64; kk++)
            if ((kept >> kk) & 1ull) acc |= mask[(size_t)(s0 + blk * 64 + kk) * mask_stride + cidx];
        } else {
          for (int kk = k0; kk < k0 + rows_per_slice && kk < 64; kk++)
            if ((kept >> kk) & 1ull) acc |= mask[(size_t)(s0 + blk * 64 + kk) * mask_stride + cidx];
        }
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
  size_t off_seg, off_keys_in, off_keys_out, off_vals_in, off_order, off_boxes, off_mask, off_nzc, off_nzrc, off_nzw, off_cub,
      cub_bytes, total;
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
  L.off_boxes = o; o += align256(sizeof(orp::QuadPrep) * n);
  L.off_mask = o; o += align256(sizeof(u64) * n * cb);
  L.off_nzc = o; o += align256(sizeof(int));
  L.off_nzrc = o; o += align256(sizeof(unsigned) * kNzCap);
  L.off_nzw = o; o += align256(sizeof(u64) * kNzCap);
  size_t cub = 0;
  hipcub::DeviceRadixSort::SortPairs((void*)nullptr, cub, (const u64*)nullptr, (u64*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, (int)n, 0, 64, (hipStream_t)0);
  L.cub_bytes = cub;
  L.off_cub = o; o += align256(cub);
  L.total = o;
  return L;
}

// dynamic LDS of the sweep: header | removed, keepbits, origbits [cb] | sparse list (12 B x kNzCap) | diag [2][64]
inline size_t sweep_smem_bytes(int max_cb) {
  return kSweepHdr + (size_t)max_cb * 3 * sizeof(u64) + (size_t)kNzCap * (sizeof(u64) + sizeof(unsigned)) + 128 * sizeof(u64);
}
inline hipError_t sweep_attr() {     // > 64 KB of dynamic LDS needs the attribute; once (not allowed during stream capture)
  static const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_sweep_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

int pick_rows_per_wave(int max_seg, int nseg) {
  static const int forced = getenv("ORP_NMS_ROWS") ? atoi(getenv("ORP_NMS_ROWS")) : 0;   // dev aid
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
  const long cb = (max_seg + 63) / 64;
  const long tiles = cb * (cb + 1) / 2 * (nseg > 0 ? nseg : 1);
  // one workgroup per (4R rows x 64 cols) tile; aim at >= 2048 workgroups (8 per CU) when the problem is big enough
  long r = tiles * 16 / 2048;
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
  orp::QuadPrep* boxes = reinterpret_cast<orp::QuadPrep*>(base + L.off_boxes);
  u64* mask = reinterpret_cast<u64*>(base + L.off_mask);
  // the sparse side list serves single-segment launches (the inference path); batched segments use the dense sweep
  int* nz_count = (nseg == 1) ? reinterpret_cast<int*>(base + L.off_nzc) : nullptr;
  unsigned* nz_rc = reinterpret_cast<unsigned*>(base + L.off_nzrc);
  u64* nz_w = reinterpret_cast<u64*>(base + L.off_nzw);
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
  hipLaunchKernelGGL(prep_boxes_kernel, dim3(nb), dim3(tb), 0, st, dets, order, n_total, boxes, nz_count);

  const int max_cb = (max_seg + 63) / 64;
  // exact_n: max_seg IS the box count (orp_rnms) -> one workgroup per tile.  Otherwise max_seg is only a capacity (the
  // count lives in device memory: batched / sync-free callers): 16-row tiles and a bounded grid whose workgroups loop
  // over the tiles of the actual count -- an 8 K capacity holding 2 K boxes must not pay for 60 K empty workgroups.
  // exact_n: max_seg IS the box count (orp_rnms); otherwise it is only a capacity (batched / sync-free callers)
  const bool exact_n = single_segment;
  const int R = exact_n ? pick_rows_per_wave(max_seg, nseg) : (max_seg <= 8192 ? (max_seg <= 256 ? 1 : 4) : 16);
  const int rpb = R * (kMaskThreads / 64);
  long ntile = ((long)max_cb * ((max_seg + rpb - 1) / rpb)) / 2 + max_cb;       // ~ the upper-triangular tiles at max_seg
  const long cap_wg = 4096 / (nseg < 8 ? nseg : 8);
  if (ntile > cap_wg) ntile = cap_wg;
  const dim3 grid((unsigned)ntile, 1, nseg);
  static const int dbg = getenv("ORP_NMS_DBG") ? atoi(getenv("ORP_NMS_DBG")) : 0;   // dev aid (timing): 1 = skip phase B, 2 = skip classifier, 4/8/16 = see tile_drain_terms
  {
    OrpProfScope prof(ORP_PROF_NMS_MASK, st);
    if (flavor == 0) hipLaunchKernelGGL(nms_mask_loop_kernel<false>, grid, dim3(kMaskThreads), 0, st, boxes, seg, R, max_cb, thr, mask, dbg, nz_count, nz_rc, nz_w);
    else hipLaunchKernelGGL(nms_mask_loop_kernel<true>, grid, dim3(kMaskThreads), 0, st, boxes, seg, R, max_cb, thr, mask, dbg, nz_count, nz_rc, nz_w);
  }

  const size_t smem = sweep_smem_bytes(max_cb);
  if (sweep_attr() != hipSuccess) return (int)sweep_attr();
  {
    OrpProfScope prof(ORP_PROF_NMS_SWEEP, st);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(nseg), dim3(kSweepThreads), smem, st, mask, order, seg, max_cb, order_out,
                       keep_out, num_keep, nz_count, nz_rc, nz_w);
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

// fp64 greedy polygon NMS over PRE-SORTED dets [n,9] (device, double): ResultMerge.py:18-41 semantics; keep_out receives
// the kept POSITIONS (ascending = visiting order), num_keep[0] their count.
size_t orp_poly_nms_f64_workspace_bytes(int n) {
  const size_t nn = (size_t)(n > 0 ? n : 1), cb = (nn + 63) / 64;
  return align256(sizeof(int32_t) * 2) + align256(sizeof(int32_t) * nn) + align256(sizeof(orp::QuadPrepT<double>) * nn) +
         align256(sizeof(u64) * nn * cb);
}

int orp_poly_nms_f64(const double* dets_sorted, int n, double iou_thr, int64_t* keep_out, int32_t* num_keep,
                     void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || (!dets_sorted && n > 0) || !keep_out || !num_keep) return ORP_EINVAL;
  if (n > ORP_NMS_MAX_BOXES) return ORP_ETOOBIG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e0 = hipMemsetAsync(num_keep, 0, sizeof(int32_t), st);
    return e0 == hipSuccess ? ORP_OK : (int)e0;
  }
  if (!workspace || workspace_bytes < orp_poly_nms_f64_workspace_bytes(n)) return ORP_EWORKSPACE;
  char* base = reinterpret_cast<char*>(workspace);
  const size_t nn = (size_t)n, cb = (nn + 63) / 64;
  int32_t* seg = reinterpret_cast<int32_t*>(base); base += align256(sizeof(int32_t) * 2);
  int32_t* order = reinterpret_cast<int32_t*>(base); base += align256(sizeof(int32_t) * nn);
  orp::QuadPrepT<double>* prep = reinterpret_cast<orp::QuadPrepT<double>*>(base); base += align256(sizeof(orp::QuadPrepT<double>) * nn);
  u64* mask = reinterpret_cast<u64*>(base);
  const int tb = 256, nb = (n + tb - 1) / tb;
  hipLaunchKernelGGL(set_single_segment_kernel, dim3(1), dim3(1), 0, st, seg, n);
  hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(tb), 0, st, order, n);
  hipLaunchKernelGGL(prep_boxes_f64_kernel, dim3(nb), dim3(tb), 0, st, dets_sorted, n, prep);
  const int max_cb = (int)cb;
  int R = 1;
  { const long tiles = (long)cb * (cb + 1) / 2; long r = tiles * 64 / 8192; while (R * 2 <= r && R < 16) R *= 2; }
  const int rpb = R * (kMaskThreads / 64);
  hipLaunchKernelGGL(nms_mask_f64_kernel, dim3(max_cb, (n + rpb - 1) / rpb), dim3(kMaskThreads), 0, st, prep, n, R,
                     max_cb, iou_thr, mask);
  const size_t smem = sweep_smem_bytes(max_cb);
  if (sweep_attr() != hipSuccess) return (int)sweep_attr();
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(kSweepThreads), smem, st, mask, order, seg, max_cb, 1, keep_out,
                     num_keep, (const int*)nullptr, (const unsigned*)nullptr, (const u64*)nullptr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
