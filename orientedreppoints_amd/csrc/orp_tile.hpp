// orp_tile.hpp -- the (rows x 64 columns) pair tile shared by the rotated-NMS mask kernel and the IoU-matrix kernels
// (gfx950, device only): LDS records of the tile's prepared boxes, the unresolved-pair queue, and the phase-B drain
// of that queue as a TERM queue (tile_drain_terms: per-term exact-zero screen, one surviving fan term per lane,
// ordered sum per pair).  See orp_quadfast.hpp for the arithmetic contract.
#pragma once
#include <hip/hip_runtime.h>

#include "orp_quadfast.hpp"

namespace orp_tile {
using orp::Pt;
typedef unsigned long long u64;

// Development aid (-DORP_NMS_PHASE_PROF): shader-clock cycles of thread 0 per phase, summed over all tiles of a launch.
// Slots 0..6 belong to the including kernel file, 8 = B1, 9 = B2, 10 = B3, 11 = chunks, 12 = B2 iterations.
#ifdef ORP_NMS_PHASE_PROF
// One row of 16 counters per workgroup (plain read-modify-write by thread 0: no atomics, the measurement must not queue
// behind itself); the host sums the rows.
constexpr int kPhaseRows = 4096;
static __device__ unsigned long long g_phase_cycles[kPhaseRows * 16];
#define ORP_PHASE_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define ORP_PHASE_ADD(slot, t0, t1) do { if (threadIdx.x == 0) orp_tile::g_phase_cycles[(blockIdx.x % orp_tile::kPhaseRows) * 16 + (slot)] += (unsigned long long)((t1) - (t0)); } while (0)
#else
#define ORP_PHASE_T(var)
#define ORP_PHASE_ADD(slot, t0, t1)
#endif

#ifndef ORP_TILE_ROWS
#define ORP_TILE_ROWS 64
#endif
constexpr int kMaxTileRows = ORP_TILE_ROWS;   // rows a tile may have (dev aid: 16 shrinks the LDS footprint for occupancy experiments)

struct TileLds {
  // oriented fan edges (ax, ay, bx, by) per edge, per tile row / column.  One float4 of padding per edge plane: phase
  // B2 assigns consecutive lanes to the terms of ONE pair, i.e. the same row / column and different edges -- with planes
  // 1024 B apart (a multiple of the 256-B bank period) those four 16-byte reads fell on the same banks
  float4 rowE[4][kMaxTileRows + 1];
  float4 colE[4][64 + 1];
  int rowS[kMaxTileRows];            // 4 signs packed 2 bits each (0 -> 0, +1 -> 1, -1 -> 2) | force_slow << 8
  int colS[64];
  float rowArea[kMaxTileRows];
  float colArea[64];
  u64 words[kMaxTileRows];
  unsigned short queue[kMaxTileRows * 64];
  int qcount;
  // phase A of the NMS mask kernel reads its (wave-uniform) row from here instead of scalar global loads: the quad's
  // vertices in polygon order (x0..x3 | y0..y3), and the row's mask of unresolved columns until the wave files them
  float4 rowV[2][kMaxTileRows];
  u64 pend[kMaxTileRows];
};

__device__ __forceinline__ int pack_signs(const orp::QuadPrep& p) {
  int v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) v |= (p.s[k] == 0 ? 0 : (p.s[k] > 0 ? 1 : 2)) << (2 * k);
  return v | (p.force_slow << 8);
}
__device__ __forceinline__ int unpack_sign(int packed, int k) {
  const int b = (packed >> (2 * k)) & 3;
  return b == 0 ? 0 : (b == 1 ? 1 : -1);
}

// ---- phase B as a TERM queue ------------------------------------------------------------------------------------
// Half of the 16 fan terms of an unresolved pair are exact zeros that orp::pair_term_alive_mask recognises from signs
// (kill-1 / kill-3), and a wave that evaluates whole pairs pays the full decision tree for them anyway.  So:
//   B1  one lane per queued pair: 16-bit mask of the terms that must be evaluated; the pair's terms get a contiguous
//       run of the term queue (wave prefix + one LDS atomic per wave); pairs with no term left are decided at once;
//   B2  one lane per TERM: the register decision tree; a term the tree does not cover (or any term of a box with
//       non-finite coordinates) is evaluated by the generic polygon loop right there -- per term, not per pair: the
//       scratch-resident loop is ~100x slower than the tree and one lane walking all 16 terms of a pair used to be
//       the tail of the whole kernel.  The value (sign applied) replaces the queue entry;
//   B3  one lane per pair: the values are summed in the reference's order (row edge outer, column edge inner -- the
//       run is in ascending term order), then `sink`.
// Chunks of kChunkPairs pairs; kChunkPairs * 16 term slots.
#ifndef ORP_CHUNK_PAIRS
#define ORP_CHUNK_PAIRS 256
#endif
constexpr int kChunkPairs = ORP_CHUNK_PAIRS;   // (a power of two <= kDrainThreads; 128 halves the term queue's 16 KB of LDS)
constexpr int kTermCap = kChunkPairs * 16;
constexpr int kDrainThreads = 256;          // workgroup size of the callers
constexpr int kTermGeneric = 1 << 30;       // queue entry flag: skip the tree

struct TermLds {
  float rowM[kMaxTileRows];          // max |coordinate| per tile row / column (the classifier bound E)
  float colM[64];
  int tq[kTermCap];                  // (pair-in-chunk << 4 | term) before B2, float bits of the term's value after
  int tcount;
};

// resets the per-chunk counter; call before the barrier that precedes tile_drain_terms
__device__ __forceinline__ void term_lds_reset(TermLds& X, int tid) {
  if (tid == 0) X.tcount = 0;
}

// Drains T.queue[0, nq) with a 256-thread workgroup (every thread must call it; contains barriers).  X must have been
// reset (term_lds_reset + barrier).  sink(rl, cl, iou) is called once per queued pair by one lane.
// dbg (development aid, timing only): 4 = skip B2, 8 = no per-term screen, 16 = skip B3.
template <bool GUARD, typename Sink>
__device__ __forceinline__ void tile_drain_terms(const TileLds& T, TermLds& X, int nq, Sink sink, int dbg = 0) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int q0 = 0; q0 < nq; q0 += kChunkPairs) {
    if (q0 > 0) {                                        // the previous chunk's B3 still reads tq
      __syncthreads();
      term_lds_reset(X, tid);
      __syncthreads();
    }
    // ---- B1 ----------------------------------------------------------------------------------------------------
    ORP_PHASE_T(tb0);
    const bool live = (tid < kChunkPairs) && (q0 + tid) < nq;
    const int item = live ? T.queue[q0 + tid] : 0;
    const int rl = item >> 6, cl = item & 63;
    unsigned alive = 0u;
    bool forced = false;
    if (live) {
      const int rs = T.rowS[rl], cs = T.colS[cl];
      float rax[4], ray[4], rbx[4], rby[4], ccx[4], ccy[4], cdx[4], cdy[4];
      int s1[4], s2[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float4 r4 = T.rowE[e][rl], c4 = T.colE[e][cl];
        rax[e] = r4.x; ray[e] = r4.y; rbx[e] = r4.z; rby[e] = r4.w;
        ccx[e] = c4.x; ccy[e] = c4.y; cdx[e] = c4.z; cdy[e] = c4.w;
        s1[e] = unpack_sign(rs, e); s2[e] = unpack_sign(cs, e);
      }
      forced = ((rs | cs) >> 8) != 0;                    // non-finite / huge coordinates: generic loop, every term
      if (forced || (dbg & 8)) {
#pragma unroll
        for (int t = 0; t < 16; t++) alive |= (s1[t >> 2] != 0 && s2[t & 3] != 0) ? (1u << t) : 0u;
      } else {
        alive = orp::pair_term_alive_mask<float>(rax, ray, rbx, rby, s1, X.rowM[rl], ccx, ccy, cdx, cdy, s2, X.colM[cl]);
      }
    }
    const int cnt = __popc(alive);
    int incl = cnt;                                      // inclusive prefix over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      incl += (lane >= off) ? v : 0;
    }
    const int wave_total = __shfl(incl, 63, 64);
    int wbase = 0;
    if (wave_total > 0) {
      if (lane == 0) wbase = atomicAdd(&X.tcount, wave_total);
      wbase = __builtin_amdgcn_readfirstlane(wbase);
    }
    const int base = wbase + incl - cnt;
    if (live) {
      if (alive == 0u) {
        sink(rl, cl, orp::iou_of_zero_inter<GUARD>(T.rowArea[rl], T.colArea[cl]));
      } else {
        const int tag = (tid << 4) | (forced ? kTermGeneric : 0);
        unsigned m = alive;
        int pos = base;
        while (m) {
          const int t = __ffs(m) - 1;
          m &= m - 1u;
          X.tq[pos++] = tag | t;
        }
      }
    }
    __syncthreads();
    ORP_PHASE_T(tb1);
    // ---- B2 ----------------------------------------------------------------------------------------------------
    const int total = (dbg & 4) ? 0 : X.tcount;
    for (int e = tid; e < total; e += kDrainThreads) {
      const int ent = X.tq[e];
      const int p = (ent >> 4) & (kChunkPairs - 1), i = (ent >> 2) & 3, j = ent & 3;
      const int it2 = T.queue[q0 + p];
      const int rl2 = it2 >> 6, cl2 = it2 & 63;
      const float4 r4 = T.rowE[i][rl2];
      const float4 g = T.colE[j][cl2];
      const int sg = unpack_sign(T.rowS[rl2], i) * unpack_sign(T.colS[cl2], j);
      const orp::FanCol f = orp::fan_col(g.x, g.y, g.z, g.w);
      bool slow = (ent & kTermGeneric) != 0;
      float v = 0.f;
      if (!slow) v = orp::tri_term_fast(r4.x, r4.y, r4.z, r4.w, f, slow);
      if (slow) {                                        // rare: this one term through the generic polygon loop
        orp::PolyPriv<float, orp::ORP_CLIP_CAP> P, Q;
        Pt<float> a, b, cc, d;
        a.x = r4.x; a.y = r4.y; b.x = r4.z; b.y = r4.w;
        cc.x = g.x; cc.y = g.y; d.x = g.z; d.y = g.w;
        v = orp::tri_term_oriented<float>(P, Q, a, b, cc, d);
      }
      if (sg == -1) v = -v;
      X.tq[e] = __float_as_int(v);
    }
    __syncthreads();
    ORP_PHASE_T(tb2);
    // ---- B3 ----------------------------------------------------------------------------------------------------
    if (live && alive != 0u && !(dbg & 16)) {
      float inter = 0.f;
      for (int t = 0; t < cnt; t++) inter += __int_as_float(X.tq[base + t]);
      const float uni = T.rowArea[rl] + T.colArea[cl] - inter;
      float iou;
      if (GUARD && uni == 0.f) iou = (inter + 1.f) / (uni + 1.f);
      else iou = inter / uni;
      sink(rl, cl, iou);
    }
#ifdef ORP_NMS_PHASE_PROF
    { ORP_PHASE_T(tb3);
      ORP_PHASE_ADD(8, tb0, tb1); ORP_PHASE_ADD(9, tb1, tb2); ORP_PHASE_ADD(10, tb2, tb3); ORP_PHASE_ADD(11, 0ull, 1ull);
      ORP_PHASE_ADD(12, 0ull, (unsigned long long)((total + kDrainThreads - 1) / kDrainThreads)); }
#endif
  }
}

}  // namespace orp_tile
