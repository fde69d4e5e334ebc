// orp_nms.hip -- rotated / polygon NMS for gfx950 (MI355X), fully on device.
//
// Replaces (reference = LiWentomng/OrientedRepPoints):
//   mmdet/ops/nms/src/rnms_kernel.cu:149-265   rnms_kernel + rnms_cuda   (device mask, D2H, SERIAL HOST sweep)
//   DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:214-329  poly_nms_kernel + _poly_nms
//
// MI355X design (not a translation of the 64-thread CUDA tiling):
//   1. rank + prepare in ONE launch over the whole chip (<= 8192 boxes per segment): the visiting order (score
//      descending, index ascending -- stable) by rank counting against the segment's keys staged in LDS, 1..64 lanes per
//      box, and the per-box record (QuadPrep: orientation, oriented origin-fan triangles, signs, |area|) written straight
//      at its rank.  The box count is read from device memory.  Larger segments use a device-wide radix sort (rocPRIM)
//      + a prepare launch;
//   2. mask kernel: a workgroup owns a (<= 64 rows x 64 columns) upper-triangular tile.  Phase A (lane = column, row
//      wave-uniform through scalar loads) proves for ~80 % of the pairs of a dense scene, without a division,
//      that every fan term is exactly 0 (orp_quadfast.hpp pair_is_far) and emits their bits by one wavefront
//      ballot; the remaining pairs are queued in LDS and drained in phase B as a TERM queue (orp_tile.hpp): a
//      per-term exact-zero screen drops half of their 16 fan terms, the others (~8 per pair, ~350 VALU instructions
//      each: 3.4 M terms for 2000 dense boxes, the arithmetic floor of the stage) run the register decision tree one
//      term per lane out of LDS and are summed per pair in the reference's order.  The box count is read from device
//      memory (exact or capacity callers alike): a bounded grid of workgroups loops over the upper-triangular tiles of
//      the actual count.  Non-zero mask words are also filed in a per-segment side list.
//      (Measured alternative, round 2: classify / screen / terms as three flat launches over global pair and term
//      queues -- perfectly balanced, but the queue traffic (8 live terms per pair) and ~4 k same-address reservation
//      atomics cost more than the tile kernel's barriers: 100-140 us against 84 us.  Not kept.)
//   3. sweep kernel: one workgroup per segment.  If the segment's side list fits in LDS (<= 8192 words) it is bucketed
//      by 64-row block and the whole greedy pass runs out of LDS (per block: one readlane-based diagonal pass + that
//      block's words only); otherwise the dense pass walks the block rows with the next row's mask words prefetched.
//      Either way the keep flags are scattered back to original indices and compacted in ascending order (popcount
//      scan), so the host never sees the mask;
//   4. fp64 instantiation of the same core for the merge NMS of the DOTA evaluation workflow (orp_poly_nms_f64).
// The IoU arithmetic is bit-identical to the reference's fp32 devrIoU / devPolyIoU (see orp_geom.hpp).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"
#include "orp_quadfast.hpp"
#include "orp_tile.hpp"
#include "orp_launch.hpp"
#include "orp_prof.hpp"

namespace {

using orp::Pt;
typedef unsigned long long u64;
using orp_tile::TileLds;
using orp_tile::TermLds;
using orp_tile::pack_signs;

constexpr int kMaskThreads = 256;   // 4 waves per workgroup

// Development aid (-DORP_NMS_PHASE_PROF, tests/checks/nms_phase_prof.py; macros in orp_tile.hpp): shader-clock cycles
// thread 0 of every workgroup spends in each phase of a tile, summed over all tiles: [0] staging, [1] phase A,
// [2] drain (B1..B3: slots 8..12), [3] mask words out, [4] tiles, [5] whole kernel per workgroup, [6] workgroups.
constexpr int kSweepThreads = 1024;
constexpr int kSortMax = 8192;      // boxes per segment the one-launch rank + prepare handles (keys staged in 32 KB of LDS)
constexpr int kNzCap = 8192;        // sparse sweep: non-zero mask words kept in LDS per segment; more -> dense sweep

__device__ __forceinline__ unsigned int float_flip_desc(float f) {
  // order-preserving float -> uint map, then inverted so that an ASCENDING radix sort yields scores DESCENDING
  unsigned int u = __float_as_uint(f);
  unsigned int mask = (u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
  return ~(u ^ mask);
}

// prep[i] = quad_prepare(box): orientation, oriented origin-fan triangles, signs, |area| -- once per box
__device__ __forceinline__ void prepare_box(const float* __restrict__ s, orp::QuadPrep* __restrict__ prep, int i) {
  float q8[8];
#pragma unroll
  for (int k = 0; k < 8; k++) q8[k] = s[k];
  orp::QuadPrep p;
  orp::quad_prepare(q8, p);
  prep[i] = p;
}

// ---- stage 1, large segments: keys[i] = (segment << 32) | flipped score ; vals[i] = i, device-wide radix sort ----
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

__global__ void prep_boxes_kernel(const float* __restrict__ dets, const int32_t* __restrict__ order, int n,
                                  orp::QuadPrep* __restrict__ prep, int* __restrict__ nz_count, int nseg) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (nz_count && i < nseg) nz_count[i] = 0;           // per-segment lists of non-zero mask words (filled by the mask kernel)
  if (i >= n) return;
  prepare_box(dets + (size_t)order[i] * 9, prep, i);
}

// ---- stage 1, the usual case: ONE launch ranks and prepares the segments (<= 8192 boxes each) --------------------------
// Visiting order by RANK COUNTING over the whole chip instead of a sort inside one workgroup: the rank of box i is the
// number of boxes that precede it in (score descending, index ascending) order -- the stable tie rule of SURVEY A3.  A
// workgroup stages the segment's keys in LDS (<= 32 KB); S = 1..64 lanes share one box, each counting over 1/S of the
// keys with ds_read_b128, then a shuffle reduction; lane 0 of the group writes order[rank] and the box's QuadPrep record
// straight at its rank.  2000 boxes: 32 workgroups x ~125 LDS reads per lane instead of a 4-pass block radix sort on one
// CU (30 us -> ~6 us); no second launch has to wait for a complete order[].
constexpr int kRankThreads = 256;
__global__ void __launch_bounds__(kRankThreads)
nms_rankprep_kernel(const float* __restrict__ dets, int32_t* __restrict__ seg_off, int single_n, int presorted,
                    int32_t* __restrict__ order, orp::QuadPrep* __restrict__ prep, int* __restrict__ nz_count) {
  __shared__ __attribute__((aligned(16))) unsigned keys[kSortMax];
  const int seg = blockIdx.y, tid = threadIdx.x;
  int s0, n;
  if (single_n >= 0) {                                     // orp_rnms: the segment table is written here (one launch less)
    s0 = 0; n = single_n;
    if (blockIdx.x == 0 && tid == 0) { seg_off[0] = 0; seg_off[1] = single_n; }
  } else {
    s0 = seg_off[seg]; n = seg_off[seg + 1] - s0;
  }
  if (blockIdx.x == 0 && tid == 0) nz_count[seg] = 0;       // this segment's list of non-zero mask words (mask kernel)
  if (n <= 0) return;
  // lanes per box: the largest power of two S <= 64 with S * n <= threads launched for this segment
  const long T = (long)gridDim.x * kRankThreads;
  int S = 1;
  while (S < 64 && (long)(2 * S) * n <= T) S <<= 1;
  const int boxes_per_block = kRankThreads / S;
  const int first = blockIdx.x * boxes_per_block;
  if (first >= n) return;
  const int npad = (n + 3) & ~3;
  for (int j = tid; j < npad; j += kRankThreads)
    keys[j] = (j < n) ? float_flip_desc(dets[(size_t)(s0 + j) * 9 + 8]) : 0xFFFFFFFFu;
  __syncthreads();
  const int i = first + tid / S, part = tid % S;
  if (i >= n) return;                                      // whole groups leave together (S divides the wave size)
  int rank = i;
  if (!presorted) {
    const unsigned ki = keys[i];
    // this lane counts over key quads [q0, q1)
    const int nq = npad >> 2;
    const int q0 = (int)((long)nq * part / S), q1 = (int)((long)nq * (part + 1) / S);
    const uint4* k4 = reinterpret_cast<const uint4*>(keys);
    int cnt = 0;
    for (int q = q0; q < q1; q++) {
      const uint4 k = k4[q];
      const int j = q << 2;
      cnt += (int)((k.x < ki) | ((k.x == ki) & (j < i)));
      cnt += (int)((k.y < ki) | ((k.y == ki) & (j + 1 < i)));
      cnt += (int)((k.z < ki) | ((k.z == ki) & (j + 2 < i)));
      cnt += (int)((k.w < ki) | ((k.w == ki) & (j + 3 < i)));
    }
    for (int off = S >> 1; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    rank = cnt;                                            // padding keys (0xFFFFFFFF, index >= n) never precede a real box
  }
  if (part == 0) {
    order[s0 + rank] = s0 + i;
    prepare_box(dets + (size_t)(s0 + i) * 9, prep, s0 + rank);
  }
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
// Compile-time switches of the round-3 changes (tools/build_variant.py builds A/B variants with -D...=0 / 1; measured on the
// 2 000-box, 15-class scene of the bench, profiles/r03_nms_variants_*.log, r03_nms_decomp_a.log):
//   ORP_NMS_ROWLDS    (off) phase A reads its wave-uniform row (vertices, max |coordinate|, |area|, flags) from the tile's LDS
//                     copy with the NEXT row's reads issued before the current row's arithmetic, instead of ~11 scalar
//                     global loads per row in front of it; the unresolved columns of a row are parked as one mask word
//                     and filed after the row loop with ONE LDS reservation per wave.  Measured slower: the row operands
//                     move from SGPRs to VGPRs and the VALU count, which is what bounds the kernel, goes up
//   ORP_NMS_DIAGLAST  (on) tiles on the diagonal (half of their pairs are below it: half the work) are enumerated last, so
//                     the final partial round of workgroups is made of the cheap tiles
//   ORP_NMS_XCD       (on, where it applies) XCD-aware tile map for single segments whose column-block count is a multiple
//                     of 16 (2 048-box capacity launches): workgroup b (XCD b % 8 on gfx950) only visits the column blocks of
//                     its XCD's set, so a column record is fetched into ONE L2 (FETCH 2.7 -> 2.2 MB per launch)
//   ORP_NMS_AGGAPPEND (on) the non-zero words of a tile are appended to the segment's side list with one reservation per tile
//                     (WRITE 2.4 -> 1.4 MB per launch)
// Together with the padded edge tables (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.46 -> 0.33): 83 -> 75 us.  What bounds the
// kernel is fp32 VALU issue: per-workgroup timeline (ORP_NMS_PHASE_PROF) shows 4 resident workgroups per CU for the first
// 45 us of the launch, phase A 22 k + drain 28 k of the 57 k cycles of a tile, and the timing-only variants (ORP_NMS_DBG) put
// the classifier alone at 34 us, the term evaluation at another 47 us.
#ifndef ORP_NMS_ROWLDS
#define ORP_NMS_ROWLDS 0
#endif
#ifndef ORP_NMS_DIAGLAST
#define ORP_NMS_DIAGLAST 1
#endif
#ifndef ORP_NMS_XCD
#define ORP_NMS_XCD 1
#endif
#ifndef ORP_NMS_AGGAPPEND
#define ORP_NMS_AGGAPPEND 1
#endif

template <bool GUARD>
__device__ __forceinline__ void mask_tile(TileLds& T, TermLds& X, const orp::QuadPrep* __restrict__ prep, int s0, int n, int c,
                                          int row_base, int rpb, int rows_per_wave, int mask_stride, float thr,
                                          u64* __restrict__ mask, int dbg, int* __restrict__ nz_count,
                                          unsigned* __restrict__ nz_rc) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  ORP_PHASE_T(pt0);

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
#if ORP_NMS_ROWLDS
        T.rowV[0][tid] = make_float4(rp.vx[0], rp.vx[1], rp.vx[2], rp.vx[3]);
        T.rowV[1][tid] = make_float4(rp.vy[0], rp.vy[1], rp.vy[2], rp.vy[3]);
#endif
      }
      T.words[tid] = 0ull;
    }
    if (tid == 0) T.qcount = 0;
    orp_tile::term_lds_reset(X, tid);
  }
  __syncthreads();
  ORP_PHASE_T(pt1);
  const bool cslow = (T.colS[lane] >> 8) != 0;
  const float carea = T.colArea[lane];

  // ---- phase A ---------------------------------------------------------------------------------------------------
  const int rl_first = __builtin_amdgcn_readfirstlane(wave * rows_per_wave);
#if ORP_NMS_ROWLDS
  {
    // rows of this wave that exist (uniform); the row record comes from LDS (same address in every lane: broadcast)
    int nrows = n - (row_base + rl_first);
    nrows = nrows < 0 ? 0 : (nrows > rows_per_wave ? rows_per_wave : nrows);
    float4 vx4 = make_float4(0.f, 0.f, 0.f, 0.f), vy4 = vx4;
    float rm = 0.f, rarea = 0.f;
    int rflags = 0;
    if (nrows > 0) {
      vx4 = T.rowV[0][rl_first]; vy4 = T.rowV[1][rl_first];
      rm = X.rowM[rl_first]; rarea = T.rowArea[rl_first]; rflags = T.rowS[rl_first];
    }
    int wave_total = 0;
    for (int rr = 0; rr < nrows; rr++) {
      const int rl = rl_first + rr;                      // wave-uniform
      const int r = row_base + rl;
      const float rvx[4] = {vx4.x, vx4.y, vx4.z, vx4.w}, rvy[4] = {vy4.x, vy4.y, vy4.z, vy4.w};
      const float crm = rm, crarea = rarea;
      const bool rslow = (rflags >> 8) != 0;
      if (rr + 1 < nrows) {                              // next row's reads fly during this row's arithmetic
        vx4 = T.rowV[0][rl + 1]; vy4 = T.rowV[1][rl + 1];
        rm = X.rowM[rl + 1]; rarea = T.rowArea[rl + 1]; rflags = T.rowS[rl + 1];
      }
      const bool valid = (col < n) & (col > r);
      bool resolved = false;
      if (valid && !(cslow | rslow)) resolved = (dbg & 2) ? true : orp::pair_is_far(rvx, rvy, crm, fc);
      const bool hit0 = resolved && (orp::iou_of_zero_inter<GUARD>(crarea, carea) > thr);
      const u64 bits = __ballot(hit0);
      const u64 pmask = __ballot(valid && !resolved);
      wave_total += __popcll(pmask);
      if (lane == 0) { T.pend[rl] = pmask; if (bits) T.words[rl] = bits; }   // this wave owns row rl in phase A
    }
    // file the wave's unresolved pairs: one reservation, rows in order, columns ascending within a row
    if (wave_total > 0) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&T.qcount, wave_total);
      base = __builtin_amdgcn_readfirstlane(base);
      for (int rr = 0; rr < nrows; rr++) {
        const int rl = rl_first + rr;
        const u64 pm = T.pend[rl];                       // written by this wave's lane 0 above (same wave: in order)
        if ((pm >> lane) & 1ull) T.queue[base + __popcll(pm & ((1ull << lane) - 1ull))] = (unsigned short)((rl << 6) | lane);
        base += __popcll(pm);
      }
    }
  }
#else
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
#endif
  __syncthreads();
  ORP_PHASE_T(pt2);

  // ---- phase B: per-term screen, one surviving fan term per lane, ordered sum per pair (orp_tile.hpp) ----------
  const int nq = (dbg & 1) ? 0 : T.qcount;
  orp_tile::tile_drain_terms<GUARD>(T, X, nq, [&](int rl, int cl, float iou) {
    if (iou > thr) atomicOr(&T.words[rl], 1ull << cl);
  }, dbg);
  __syncthreads();
  ORP_PHASE_T(pt3);
#if ORP_NMS_AGGAPPEND
  if (wave == 0) {                                       // rpb <= 64: the tile's rows are the lanes of wave 0
    const bool have = (lane < rpb) && (row_base + lane < n);
    const u64 w = have ? T.words[lane] : 0ull;
    if (have) mask[(size_t)(c) * mask_stride + (s0 + row_base + lane)] = w;
    const u64 nzm = __ballot(w != 0ull);
    if (nzm) {                                           // one reservation in the segment's side list per tile
      int pos0 = 0;
      if (lane == 0) pos0 = atomicAdd(nz_count, __popcll(nzm));
      pos0 = __builtin_amdgcn_readfirstlane(pos0);
      const int pos = pos0 + __popcll(nzm & ((1ull << lane) - 1ull));
      if (w != 0ull && pos < kNzCap) nz_rc[pos] = ((unsigned)(row_base + lane) << 11) | (unsigned)c;
    }
  }
#else
  if (tid < rpb && row_base + tid < n) {
    const u64 w = T.words[tid];
    mask[(size_t)(c) * mask_stride + (s0 + row_base + tid)] = w;
    if (w) {                                             // sparse side list of this segment for the LDS-resident sweep
      const int pos = atomicAdd(nz_count, 1);
      if (pos < kNzCap) nz_rc[pos] = ((unsigned)(row_base + tid) << 11) | (unsigned)c;
    }
  }
#endif
#ifdef ORP_NMS_PHASE_PROF
  { ORP_PHASE_T(pt4);
    ORP_PHASE_ADD(0, pt0, pt1); ORP_PHASE_ADD(1, pt1, pt2); ORP_PHASE_ADD(2, pt2, pt3); ORP_PHASE_ADD(3, pt3, pt4);
    ORP_PHASE_ADD(4, 0ull, 1ull); }
#endif
}

// ---- tile enumeration ---------------------------------------------------------------------------------------------------
// Only the upper-triangular tiles exist.  With q = 64 / rpb row groups per 64-row batch, batch j (rows [64j, 64j+64)) has
// FULL tiles in the column blocks j+1 .. cbn-1 and DIAGONAL tiles in column block j (half of their pairs lie below the
// diagonal: about half the work).  The last batch may have fewer than q row groups.
struct TileMap {
  int cbn, ngroups, q, last;
  long n_full, n_diag;
  __device__ __forceinline__ long full_before(int j) const {       // full tiles of batches < j
    return (long)q * ((long)j * (cbn - 1) - (long)j * (j - 1) / 2);
  }
  __device__ __forceinline__ void init(int n, int rpb) {
    cbn = (n + 63) >> 6; ngroups = (n + rpb - 1) / rpb; q = 64 / rpb; last = cbn - 1;
    n_full = full_before(last);                                       // the last batch has no full tile
    n_diag = ngroups;                                                 // one diagonal tile per row group
  }
  // t in [0, n_full + n_diag): full tiles first (batch by batch, row group by row group), then the diagonal tiles
  __device__ __forceinline__ void decode_diag_last(long t, int& g, int& c) const {
    if (t >= n_full) { g = (int)(t - n_full); c = g / q; return; }
    const double b = 2.0 * cbn - 1.0;
    int j = (int)((b - sqrt(b * b - 8.0 * (double)t / (double)q)) * 0.5);
    j = j < 0 ? 0 : (j > last - 1 ? last - 1 : j);
    while (j > 0 && full_before(j) > t) j--;
    while (j + 1 < last && full_before(j + 1) <= t) j++;
    const int r = (int)(t - full_before(j)), w = cbn - 1 - j;
    g = j * q + r / w; c = j + 1 + r % w;
  }
  // round-2 order: batch by batch, the diagonal tile first in every row group
  __device__ __forceinline__ long any_before(int j) const { return (long)q * ((long)j * cbn - (long)j * (j - 1) / 2); }
  __device__ __forceinline__ void decode_row_major(long t, int& g, int& c) const {
    const long t_last = any_before(last);
    if (t >= t_last) { g = last * q + (int)(t - t_last); c = last; return; }
    const double b = 2.0 * cbn + 1.0;
    int j = (int)((b - sqrt(b * b - 8.0 * (double)t / (double)q)) * 0.5);
    j = j < 0 ? 0 : (j > last - 1 ? last - 1 : j);
    while (j > 0 && any_before(j) > t) j--;
    while (j + 1 < last && any_before(j + 1) <= t) j++;
    const int r = (int)(t - any_before(j)), w = cbn - j;
    g = j * q + r / w; c = j + r % w;
  }
  // XCD-aware lists: XCD x owns the column blocks c_m = 8m + (m even ? x : 7 - x) (boustrophedon: the tile counts of
  // the eight lists differ by at most one column's worth).  List x: the full tiles of its columns (column c has q*c of
  // them: row groups 0 .. q*c-1), column by column, then its diagonal tiles.  Returns false past the end of list x.
  __device__ __forceinline__ int xcd_col(int x, int m) const { return 8 * m + ((m & 1) ? 7 - x : x); }
  __device__ __forceinline__ bool decode_xcd(int x, long k, int& g, int& c) const {
    for (int m = 0;; m++) {                                           // full tiles
      const int cc = xcd_col(x, m);
      if (cc >= cbn) break;
      const long cnt = (long)q * cc;
      if (k < cnt) { c = cc; g = (int)k; return true; }
      k -= cnt;
    }
    for (int m = 0;; m++) {                                           // diagonal tiles
      const int cc = xcd_col(x, m);
      if (cc >= cbn) break;
      const int cnt = (cc == last) ? (ngroups - last * q) : q;
      if (k < cnt) { c = cc; g = cc * q + (int)k; return true; }
      k -= cnt;
    }
    return false;
  }
};

// The box count is read from DEVICE memory (seg_off): the host may only know an upper bound (sync-free / hipGraph callers:
// an 8 K capacity holding 2 K boxes must not pay for 60 K empty workgroups).  A bounded grid of workgroups loops over the
// UPPER-TRIANGULAR tiles of the actual count (enumerated in closed form; lower-triangular tiles are never read by the
// sweep), which also balances the load: 117 -> 83 us at 2000 boxes against one workgroup per tile of the full grid.
#ifndef ORP_MASK_WGS
#define ORP_MASK_WGS 4
#endif
template <bool GUARD>
__global__ void __launch_bounds__(kMaskThreads, ORP_MASK_WGS)
nms_mask_loop_kernel(const orp::QuadPrep* __restrict__ prep, const int32_t* __restrict__ seg_off, int rows_per_wave,
                     int mask_stride, float thr, u64* __restrict__ mask, int dbg, int* __restrict__ nz_count,
                     unsigned* __restrict__ nz_rc, int xcd_map) {
  __shared__ TileLds T;
  __shared__ TermLds X;
  ORP_PHASE_T(pk0);
#ifdef ORP_NMS_PHASE_PROF
  if (threadIdx.x == 0) orp_tile::g_phase_cycles[(blockIdx.x % orp_tile::kPhaseRows) * 16 + 13] = wall_clock64();   // 100 MHz, chip-wide
#endif
  const int seg = blockIdx.z;
  const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
  const int rpb = rows_per_wave * (kMaskThreads / 64);
  if (n <= 0) return;
  TileMap M;
  M.init(n, rpb);
  const long total_tiles = M.n_full + M.n_diag;
  // gfx950 places workgroup b on XCD b % 8 (observed, used for speed only: any placement gives the same mask)
  // (only when the eight lists are equally long: a multiple of 16 column blocks)
  const bool xcd = ORP_NMS_XCD && (xcd_map & 1) && M.cbn >= 16 && (M.cbn & 15) == 0 && (gridDim.x & 7) == 0;
  const int x = blockIdx.x & 7;
  const long step = xcd ? (long)(gridDim.x >> 3) : (long)gridDim.x;
  for (long k = xcd ? (long)(blockIdx.x >> 3) : (long)blockIdx.x;; k += step) {
    int g, c, row_base;
    if (xcd) {
      if (!M.decode_xcd(x, k, g, c)) break;
      row_base = g * rpb;
    } else {
      if (k >= total_tiles) break;
      if (ORP_NMS_DIAGLAST) M.decode_diag_last(k, g, c); else M.decode_row_major(k, g, c);
      row_base = g * rpb;
    }
    mask_tile<GUARD>(T, X, prep, s0, n, c, row_base, rpb, rows_per_wave, mask_stride, thr, mask, dbg, nz_count + seg,
                     nz_rc + (size_t)seg * kNzCap);
    __syncthreads();                                     // the LDS tile is reused by the next tile
  }
#ifdef ORP_NMS_PHASE_PROF
  { ORP_PHASE_T(pk1); ORP_PHASE_ADD(5, pk0, pk1); ORP_PHASE_ADD(6, 0ull, 1ull);
    if (threadIdx.x == 0) orp_tile::g_phase_cycles[(blockIdx.x % orp_tile::kPhaseRows) * 16 + 14] = wall_clock64(); }
#endif
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
    if (lane == 0) mask[(size_t)(c) * mask_stride + r] = bits;
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

// ---- sweep of a small segment (<= 4096 boxes, sparse list fits): the serial part by ONE wave ---------------------------
// LDS (own carve of the dynamic buffer): removed | keepbits | origbits | word prefix [64 each] | bucket ends [64] | list
// words / coordinates [kNzCap] | diagonal words [64][64].  The whole workgroup fetches the listed words and buckets them
// by 64-row block (counting sort, 4 barriers); the greedy pass over the blocks is then run by wave 0 alone WITHOUT any
// workgroup barrier (per block: one broadcast read of removed[blk], the readlane-based diagonal pass, one or two rounds
// over that block's words); the workgroup comes back for the compaction.  ~8 barriers per call instead of ~110.
constexpr int kSmallCb = 64;
inline size_t sweep_small_smem_bytes() {
  return 16 + sizeof(u64) * 3 * kSmallCb + sizeof(int) * 2 * kSmallCb + (size_t)kNzCap * (sizeof(u64) + sizeof(unsigned)) +
         sizeof(u64) * kSmallCb * 64;
}
__device__ __forceinline__ void wave_lds_sync() {         // orders this wave's LDS writes before its later LDS reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void sweep_small(unsigned char* smem, const u64* __restrict__ mask,
                                            const int32_t* __restrict__ order, int s0, int n, int cb, int mask_stride,
                                            int order_out, int64_t* __restrict__ keep_out, int32_t* __restrict__ num_keep,
                                            int seg, int nnz, const unsigned* __restrict__ my_rc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u64* removed = reinterpret_cast<u64*>(smem + 16);
  u64* keepbits = removed + kSmallCb;
  u64* origbits = keepbits + kSmallCb;
  int* wpre = reinterpret_cast<int*>(origbits + kSmallCb);
  int* bend = wpre + kSmallCb;
  u64* nzw = reinterpret_cast<u64*>(bend + kSmallCb);
  unsigned* nzrc = reinterpret_cast<unsigned*>(nzw + kNzCap);
  u64* diagw = reinterpret_cast<u64*>(nzrc + kNzCap);            // [cb][64]: word (row, row >> 6) of every row

  if (tid < kSmallCb) { removed[tid] = 0ull; keepbits[tid] = 0ull; origbits[tid] = 0ull; bend[tid] = 0; }
  for (int i = tid; i < cb * 64; i += kSweepThreads) diagw[i] = 0ull;
  __syncthreads();
  for (int i = tid; i < nnz; i += kSweepThreads) atomicAdd(&bend[my_rc[i] >> 17], 1);
  __syncthreads();
  if (wave == 0) {                                         // exclusive scan of the 64 bucket sizes: one shuffle scan
    const int v = bend[lane];
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      incl += (lane >= off) ? u : 0;
    }
    bend[lane] = incl - v;                                 // start of bucket `lane` (becomes its end after the scatter)
  }
  __syncthreads();
  for (int i = tid; i < nnz; i += kSweepThreads) {
    const unsigned rc = my_rc[i];
    const unsigned row = rc >> 11, cc = rc & 2047u;
    const u64 w = mask[(size_t)(cc) * mask_stride + (s0 + (int)row)];
    const int pos = atomicAdd(&bend[row >> 6], 1);
    nzrc[pos] = rc; nzw[pos] = w;
    if (cc == (row >> 6)) diagw[row] = w;
  }
  __syncthreads();
  if (wave == 0) {
    for (int blk = 0; blk < cb; blk++) {
      const u64 d = diagw[blk * 64 + lane];
      const u64 cur0 = removed[blk];                       // same address for the whole wave: broadcast
      unsigned clo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur0);
      unsigned chi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur0 >> 32));
      u64 cur = ((u64)chi << 32) | clo;
      const int valid = min(64, n - blk * 64);
      const u64 vmask = (valid >= 64) ? ~0ull : ((1ull << valid) - 1ull);
      const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
      u64 todo = __ballot(d != 0ull) & vmask;              // rows that can suppress inside this block
      while (todo) {                                       // wave-uniform, ascending row order
        const int kk = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        if (!((cur >> kk) & 1ull))
          cur |= ((u64)(unsigned)__builtin_amdgcn_readlane(dhi, kk) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane(dlo, kk);
      }
      const u64 kept = ~cur & vmask;
      if (lane == 0) keepbits[blk] = kept;
      const int b0 = blk ? bend[blk - 1] : 0, b1 = bend[blk];
      for (int i = b0 + lane; i < b1; i += 64) {
        const unsigned rc = nzrc[i];
        const unsigned cc = rc & 2047u, rl = (rc >> 11) & 63u;
        if (cc > (unsigned)blk && ((kept >> rl) & 1ull)) atomicOr(&removed[cc], nzw[i]);
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  // ---- compaction ----------------------------------------------------------------------------------------------------------
  const u64* bits = keepbits;                               // visiting order: positions are the answer's order
  if (order_out != 1) {                                    // ascending original index: scatter the flags first
    for (int i = tid; i < n; i += kSweepThreads) {
      if ((keepbits[i >> 6] >> (i & 63)) & 1ull) {
        const int o = order[s0 + i] - s0;
        atomicOr(&origbits[o >> 6], 1ull << (o & 63));
      }
    }
    __syncthreads();
    bits = origbits;
  }
  if (wave == 0) {
    const int cnt = __popcll(bits[lane]);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      incl += (lane >= off) ? u : 0;
    }
    wpre[lane] = incl - cnt;
    if (lane == 63) num_keep[seg] = incl;
  }
  __syncthreads();
  for (int i = tid; i < n; i += kSweepThreads) {
    const u64 w = bits[i >> 6];
    if ((w >> (i & 63)) & 1ull) {
      const int pos = wpre[i >> 6] + __popcll(w & ((1ull << (i & 63)) - 1ull));
      keep_out[s0 + pos] = (order_out == 1) ? (int64_t)order[s0 + i] : (int64_t)(s0 + i);
    }
  }
}

__global__ void __launch_bounds__(kSweepThreads)
nms_sweep_kernel(const u64* __restrict__ mask, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_off,
                 int mask_stride, int order_out, int64_t* __restrict__ keep_out, int32_t* __restrict__ num_keep,
                 const int* __restrict__ nz_count, const unsigned* __restrict__ nz_rc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int seg = blockIdx.x;
  const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n <= 0) { if (tid == 0) num_keep[seg] = 0; return; }
  const int cb = (n + 63) >> 6;
  if (cb <= kSmallCb && nz_count && nz_count[seg] <= kNzCap) {     // small segment (uniform branch)
    sweep_small(smem, mask, order, s0, n, cb, mask_stride, order_out, keep_out, num_keep, seg, nz_count[seg],
                nz_rc + (size_t)seg * kNzCap);
    return;
  }
  u64* s_kept = reinterpret_cast<u64*>(smem);
  int* tmp = reinterpret_cast<int*>(smem + 16);
  u64* removed = reinterpret_cast<u64*>(smem + kSweepHdr);
  u64* keepbits = removed + cb;
  u64* origbits = keepbits + cb;             // keep flags in ORIGINAL-index space (segment-local)

  for (int i = tid; i < cb; i += kSweepThreads) { removed[i] = 0; keepbits[i] = 0; origbits[i] = 0; }
  __syncthreads();

  // ---- sparse sweep: the mask of a detection scene is mostly zeros; the classify / pairs launches filed the coordinates
  // of every non-zero word in a per-segment list.  If the list fits (<= kNzCap words) the whole greedy pass runs out of
  // LDS: the words are fetched once, bucketed by 64-row block (counting sort), and block blk then only touches ITS
  // words: the diagonal ones feed the readlane-based pass of wave 0, the others are ORed into `removed` for the kept
  // rows -- two barriers and no HBM / L2 round trip per block.
  const int nnz = nz_count ? nz_count[seg] : (kNzCap + 1);
  const bool sparse = nnz <= kNzCap;
  if (sparse) {
    u64* nzw = origbits + cb;
    unsigned* nzrc = reinterpret_cast<unsigned*>(nzw + kNzCap);
    u64* diag = reinterpret_cast<u64*>(nzrc + kNzCap);              // [2][64]
    int* bend = reinterpret_cast<int*>(diag + 128);                 // [cb]: end of bucket blk after the scatter
    const unsigned* my_rc = nz_rc + (size_t)seg * kNzCap;
    for (int i = tid; i < cb; i += kSweepThreads) bend[i] = 0;
    if (tid < 128) diag[tid] = 0ull;
    __syncthreads();
    for (int i = tid; i < nnz; i += kSweepThreads) atomicAdd(&bend[my_rc[i] >> 17], 1);
    __syncthreads();
    // exclusive scan of the bucket sizes (cb <= 2048), chunk by chunk through tmp
    {
      int running = 0;
      for (int base = 0; base < cb; base += kSweepThreads) {
        const int i = base + tid;
        const int v = (i < cb) ? bend[i] : 0;
        tmp[tid] = v;
        __syncthreads();
        for (int off = 1; off < kSweepThreads; off <<= 1) {
          const int u = (tid >= off) ? tmp[tid - off] : 0;
          __syncthreads();
          tmp[tid] += u;
          __syncthreads();
        }
        if (i < cb) bend[i] = running + tmp[tid] - v;               // start of bucket i (becomes its end below)
        const int chunk_total = tmp[kSweepThreads - 1];
        __syncthreads();
        running += chunk_total;
      }
    }
    for (int i = tid; i < nnz; i += kSweepThreads) {
      const unsigned rc = my_rc[i];
      const int pos = atomicAdd(&bend[rc >> 17], 1);
      nzrc[pos] = rc;
      nzw[pos] = mask[(size_t)((rc & 2047u)) * mask_stride + (s0 + (int)(rc >> 11))];
    }
    __syncthreads();
    // block 0's diagonal words
    for (int i = tid; i < bend[0]; i += kSweepThreads)
      if ((nzrc[i] & 2047u) == 0u) diag[(nzrc[i] >> 11) & 63u] = nzw[i];
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
      // this block's words (right of the diagonal) of the kept rows; the next block's diagonal words
      const int b0 = blk ? bend[blk - 1] : 0, b1 = bend[blk], b2 = (blk + 1 < cb) ? bend[blk + 1] : b1;
      for (int i = b0 + tid; i < b2; i += kSweepThreads) {
        const unsigned rc = nzrc[i];
        const unsigned cc = rc & 2047u, rl = (rc >> 11) & 63u;
        if (i < b1) { if (cc > ublk && ((kept >> rl) & 1ull)) atomicOr(&removed[cc], nzw[i]); }
        else if (cc == ublk + 1u) diag[((blk + 1) & 1) * 64 + rl] = nzw[i];
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
  if (!sparse && wave == 0) { const int row = lane; d_next = (row < n) ? mask[(size_t)(0) * mask_stride + (s0 + row)] : 0ull; }
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
      pre[u] = (has && u < rows_per_slice && kk < 64 && row < n) ? mask[(size_t)(cidx0) * mask_stride + (s0 + row)] : 0ull;
    }
    if (wave == 0) {
      const u64 d = d_next;
      const int nrow = (blk + 1) * 64 + lane;
      d_next = (blk + 1 < cb && nrow < n) ? mask[(size_t)(blk + 1) * mask_stride + (s0 + nrow)] : 0ull;
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
            if ((kept >> kk) & 1ull) acc |= mask[(size_t)(cidx) * mask_stride + (s0 + blk * 64 + kk)];
        } else {
          for (int kk = k0; kk < k0 + rows_per_slice && kk < 64; kk++)
            if ((kept >> kk) & 1ull) acc |= mask[(size_t)(cidx) * mask_stride + (s0 + blk * 64 + kk)];
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
  size_t off_seg, off_keys_in, off_keys_out, off_vals_in, off_order, off_boxes, off_mask, off_nzc, off_nzrc, off_cub,
      cub_bytes, total;
};

NmsLayout nms_layout(int n_total, int nseg, int max_seg) {
  NmsLayout L;
  size_t o = 0;
  const size_t n = (size_t)(n_total > 0 ? n_total : 1);
  const size_t ns = (size_t)(nseg > 0 ? nseg : 1);
  const size_t cb = (size_t)((max_seg + 63) / 64 > 0 ? (max_seg + 63) / 64 : 1);
  L.off_seg = o; o += align256(sizeof(int32_t) * (ns + 1));
  const bool big = max_seg > kSortMax;                     // device-wide radix sort needed
  L.off_keys_in = o; o += big ? align256(sizeof(u64) * n) : 0;
  L.off_keys_out = o; o += big ? align256(sizeof(u64) * n) : 0;
  L.off_vals_in = o; o += big ? align256(sizeof(int32_t) * n) : 0;
  L.off_order = o; o += align256(sizeof(int32_t) * n);
  L.off_boxes = o; o += align256(sizeof(orp::QuadPrep) * n);
  L.off_mask = o; o += align256(sizeof(u64) * n * cb);
  L.off_nzc = o; o += align256(sizeof(int) * ns);
  L.off_nzrc = o; o += align256(sizeof(unsigned) * kNzCap * ns);
  size_t cub = 0;
  if (big)
    hipcub::DeviceRadixSort::SortPairs((void*)nullptr, cub, (const u64*)nullptr, (u64*)nullptr, (const int32_t*)nullptr,
                                       (int32_t*)nullptr, (int)n, 0, 64, (hipStream_t)0);
  L.cub_bytes = cub;
  L.off_cub = o; o += align256(cub);
  L.total = o;
  return L;
}

// dynamic LDS of the sweep: header | removed, keepbits, origbits [cb] | sparse list (12 B x kNzCap) | diag [2][64] | bucket ends [cb]
inline size_t sweep_smem_bytes(int max_cb) {
  const size_t general = kSweepHdr + (size_t)max_cb * 3 * sizeof(u64) + (size_t)kNzCap * (sizeof(u64) + sizeof(unsigned)) +
                         128 * sizeof(u64) + (size_t)max_cb * sizeof(int);
  const size_t small = sweep_small_smem_bytes();           // the one-wave path carves the same buffer its own way
  return general > small ? general : small;
}
inline hipError_t sweep_attr() {     // > 64 KB of dynamic LDS needs the attribute: once per device, never during a capture
  struct Tag {};
  return orp::set_max_dynamic_lds_once<Tag>(reinterpret_cast<const void*>(&nms_sweep_kernel), 160 * 1024);
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
  int32_t* order = reinterpret_cast<int32_t*>(base + L.off_order);
  orp::QuadPrep* boxes = reinterpret_cast<orp::QuadPrep*>(base + L.off_boxes);
  u64* mask = reinterpret_cast<u64*>(base + L.off_mask);
  int* nz_count = reinterpret_cast<int*>(base + L.off_nzc);
  unsigned* nz_rc = reinterpret_cast<unsigned*>(base + L.off_nzrc);

  const bool one_launch = max_seg <= kSortMax;            // rank + prepare in one launch (also writes the segment table)
  if (!single_segment) {
    hipError_t e = hipMemcpyAsync(seg, seg_off_dev, sizeof(int32_t) * (size_t)(nseg + 1), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  } else if (!one_launch || n_total == 0) {
    hipLaunchKernelGGL(set_single_segment_kernel, dim3(1), dim3(1), 0, st, seg, n_total);
  }
  if (n_total == 0) {
    hipError_t e = orp::fill_async(num_keep, 0, sizeof(int32_t) * (size_t)nseg, st);
    return e == hipSuccess ? ORP_OK : (int)e;
  }
  // ---- stage 1: visiting order + per-box records ----------------------------------------------------------------------
  {
    OrpProfScope prof(ORP_PROF_NMS_SORT, st);
    if (one_launch) {
      // threads per segment: 4 lanes per box up to 2048 boxes of capacity, then enough for <= ~512 keys per lane
      long thr_seg = (long)max_seg * (max_seg <= 2048 ? 4 : (max_seg + 511) / 512);
      if (thr_seg < kRankThreads) thr_seg = kRankThreads;
      const unsigned gx = (unsigned)((thr_seg + kRankThreads - 1) / kRankThreads);
      hipLaunchKernelGGL(nms_rankprep_kernel, dim3(gx, nseg), dim3(kRankThreads), 0, st, dets, seg,
                         single_segment ? n_total : -1, presorted, order, boxes, nz_count);
    } else {
      const int tb = 256;
      const int nb = ((n_total > nseg ? n_total : nseg) + tb - 1) / tb;
      if (presorted) {
        hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(tb), 0, st, order, n_total);
      } else {
        u64* keys_in = reinterpret_cast<u64*>(base + L.off_keys_in);
        u64* keys_out = reinterpret_cast<u64*>(base + L.off_keys_out);
        int32_t* vals_in = reinterpret_cast<int32_t*>(base + L.off_vals_in);
        hipLaunchKernelGGL(make_keys_kernel, dim3(nb), dim3(tb), 0, st, dets, n_total, seg, nseg, keys_in, vals_in);
        size_t cub_bytes = L.cub_bytes;
        int end_bit = 32;
        { int s = nseg - 1; while (s > 0) { end_bit++; s >>= 1; } }
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(base + L.off_cub, cub_bytes, keys_in, keys_out, vals_in, order,
                                                          n_total, 0, end_bit, st);
        if (e != hipSuccess) return (int)e;
      }
      hipLaunchKernelGGL(prep_boxes_kernel, dim3(nb), dim3(tb), 0, st, dets, order, n_total, boxes, nz_count, nseg);
    }
  }

  // ---- stage 2: the suppression mask ---------------------------------------------------------------------------------------
  const int max_cb = (max_seg + 63) / 64;
  // exact_n: max_seg IS the box count (orp_rnms); otherwise it is only a capacity (batched / sync-free callers: the
  // count lives in device memory): 16-row tiles and a bounded grid whose workgroups loop over the tiles of the actual
  // count -- an 8 K capacity holding 2 K boxes must not pay for 60 K empty workgroups
  const bool exact_n = single_segment;
  // capacity callers: 16-row tiles (a 4-row tile spends its time staging: 201 -> measured below for 240 segments of <= 168)
  const int R = exact_n ? pick_rows_per_wave(max_seg, nseg) : (max_seg <= 8192 ? 4 : 16);
  const int rpb = R * (kMaskThreads / 64);
  long ntile = ((long)max_cb * ((max_seg + rpb - 1) / rpb)) / 2 + max_cb;       // ~ the upper-triangular tiles at max_seg
  const long cap_wg = 4096 / (nseg < 8 ? nseg : 8);
  if (ntile > cap_wg) ntile = cap_wg;
  // XCD-aware tile lists for one segment (TileMap::decode_xcd; the kernel checks the actual count).  Measured round 3:
  // the launch time does not change (82.9 vs 83.9 us), the column records are fetched into one L2 instead of eight
  static const int forced_map = getenv("ORP_NMS_MAP") ? atoi(getenv("ORP_NMS_MAP")) : -1;   // dev aid
  const int xcd_map = forced_map >= 0 ? forced_map : (nseg == 1 ? 1 : 0);
  if (xcd_map & 1) ntile = (ntile + 7) & ~7L;              // grid.x a multiple of 8
  const dim3 grid((unsigned)ntile, 1, nseg);
  static const int dbg = getenv("ORP_NMS_DBG") ? atoi(getenv("ORP_NMS_DBG")) : 0;   // dev aid (timing): 1 = skip phase B, 2 = skip classifier, 4/8/16 = see tile_drain_terms
  {
    OrpProfScope prof(ORP_PROF_NMS_MASK, st);
    if (flavor == 0) hipLaunchKernelGGL(nms_mask_loop_kernel<false>, grid, dim3(kMaskThreads), 0, st, boxes, seg, R, n_total, thr, mask, dbg, nz_count, nz_rc, xcd_map);
    else hipLaunchKernelGGL(nms_mask_loop_kernel<true>, grid, dim3(kMaskThreads), 0, st, boxes, seg, R, n_total, thr, mask, dbg, nz_count, nz_rc, xcd_map);
  }

  // ---- stage 3: greedy sweep + compaction ------------------------------------------------------------------------------------
  const size_t smem = sweep_smem_bytes(max_cb);
  if (sweep_attr() != hipSuccess) return (int)sweep_attr();
  {
    OrpProfScope prof(ORP_PROF_NMS_SWEEP, st);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(nseg), dim3(kSweepThreads), smem, st, mask, order, seg, n_total, order_out,
                       keep_out, num_keep, nz_count, nz_rc);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // namespace

extern "C" {

#ifdef ORP_NMS_PHASE_PROF
int orp_nms_phase_prof_raw(unsigned long long* out /* [4096 * 16] */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(orp_tile::g_phase_cycles), sizeof(unsigned long long) * 16 * orp_tile::kPhaseRows) == hipSuccess ? 0 : -1;
}
int orp_nms_phase_prof_read(unsigned long long* out16, int reset) {
  const size_t nb = sizeof(unsigned long long) * 16 * orp_tile::kPhaseRows;
  unsigned long long* h = (unsigned long long*)malloc(nb);
  if (!h) return -1;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(orp_tile::g_phase_cycles), nb) != hipSuccess) { free(h); return -1; }
  for (int k = 0; k < 16; k++) out16[k] = 0;
  for (int r = 0; r < orp_tile::kPhaseRows; r++) for (int k = 0; k < 16; k++) out16[k] += h[r * 16 + k];
  if (reset) { memset(h, 0, nb); if (hipMemcpyToSymbol(HIP_SYMBOL(orp_tile::g_phase_cycles), h, nb) != hipSuccess) { free(h); return -1; } }
  free(h);
  return 0;
}
#endif

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
    hipError_t e0 = orp::fill_async(num_keep, 0, sizeof(int32_t), st);
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
                     n, iou_thr, mask);
  const size_t smem = sweep_smem_bytes(max_cb);
  if (sweep_attr() != hipSuccess) return (int)sweep_attr();
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(kSweepThreads), smem, st, mask, order, seg, n, 1, keep_out,
                     num_keep, (const int*)nullptr, (const unsigned*)nullptr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
