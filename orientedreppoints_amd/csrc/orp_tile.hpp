// orp_tile.hpp -- the (rows x 64 columns) pair tile shared by the rotated-NMS mask kernel and the IoU-matrix kernels
// (gfx950, device only): LDS records of the tile's prepared boxes, the unresolved-pair queue, and the phase-B
// evaluation of one queued pair by a quad of lanes.  See orp_quadfast.hpp for the arithmetic contract.
#pragma once
#include <hip/hip_runtime.h>

#include "orp_quadfast.hpp"

namespace orp_tile {
using orp::Pt;
typedef unsigned long long u64;

constexpr int kMaxTileRows = 64;

struct TileLds {
  float4 rowE[4][kMaxTileRows];      // oriented fan edges (ax, ay, bx, by) per edge, per tile row
  float4 colE[4][64];
  int rowS[kMaxTileRows];            // 4 signs packed 2 bits each (0 -> 0, +1 -> 1, -1 -> 2) | force_slow << 8
  int colS[64];
  float rowArea[kMaxTileRows];
  float colArea[64];
  u64 words[kMaxTileRows];
  unsigned short queue[kMaxTileRows * 64];
  int qcount;
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

// value of lane (quad base + k) for every lane of a quad (DPP quad_perm broadcast, one VALU op)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ int quad_bcast_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, true);
}

// Phase B: one queued pair per QUAD of lanes -- lane k of the quad evaluates the four fan terms of row edge k, then
// the 16 values are summed in the reference's order (row edge outer, column edge inner) through quad broadcasts, so
// the fp32 accumulation is unchanged while the critical path per pair is 4 terms instead of 16.
// Returns the pair's IoU (valid on every lane of the quad).
template <bool GUARD>
__device__ __forceinline__ float tile_pair_iou_quad(const TileLds& T, int rl, int cl, int k, bool live) {
  const int rs = T.rowS[rl], cs = T.colS[cl];
  bool slow = live && (((rs | cs) >> 8) != 0);
  const int s1 = unpack_sign(rs, k);
  const float4 e = T.rowE[k][rl];
  float t[4];
#pragma unroll 1
  for (int j = 0; j < 4; j++) {
    const int s2 = unpack_sign(cs, j);
    float v = 0.f;
    if (live && s1 != 0 && s2 != 0) {
      const float4 g = T.colE[j][cl];
      const orp::FanCol f = orp::fan_col(g.x, g.y, g.z, g.w);
      v = orp::tri_term_fast(e.x, e.y, e.z, e.w, f, slow);
      if (s1 * s2 == -1) v = -v;
    }
    // static register slot for a dynamic j without private-memory indexing
    t[0] = (j == 0) ? v : t[0]; t[1] = (j == 1) ? v : t[1]; t[2] = (j == 2) ? v : t[2]; t[3] = (j == 3) ? v : t[3];
  }
  // skipped terms contribute +0: inter never holds -0, so x + (+-0) == x and the sum equals the reference's
  float inter = 0.f;
  inter += quad_bcast<0>(t[0]); inter += quad_bcast<0>(t[1]); inter += quad_bcast<0>(t[2]); inter += quad_bcast<0>(t[3]);
  inter += quad_bcast<1>(t[0]); inter += quad_bcast<1>(t[1]); inter += quad_bcast<1>(t[2]); inter += quad_bcast<1>(t[3]);
  inter += quad_bcast<2>(t[0]); inter += quad_bcast<2>(t[1]); inter += quad_bcast<2>(t[2]); inter += quad_bcast<2>(t[3]);
  inter += quad_bcast<3>(t[0]); inter += quad_bcast<3>(t[1]); inter += quad_bcast<3>(t[2]); inter += quad_bcast<3>(t[3]);
  const int sl = slow ? 1 : 0;
  const int any_slow = quad_bcast_i<0>(sl) | quad_bcast_i<1>(sl) | quad_bcast_i<2>(sl) | quad_bcast_i<3>(sl);
  if (any_slow && k == 0) {                              // generic polygon loop, scratch-resident (rare)
    orp::PolyPriv<float, orp::ORP_CLIP_CAP> P, Q;
    inter = 0.f;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const int si = unpack_sign(rs, i);
      if (si == 0) continue;
      const float4 ei = T.rowE[i][rl];
      Pt<float> a, b;
      a.x = ei.x; a.y = ei.y; b.x = ei.z; b.y = ei.w;
#pragma unroll 1
      for (int j = 0; j < 4; j++) {
        const int s2 = unpack_sign(cs, j);
        if (s2 == 0) continue;
        const float4 g = T.colE[j][cl];
        Pt<float> cc, d;
        cc.x = g.x; cc.y = g.y; d.x = g.z; d.y = g.w;
        float v = orp::tri_term_oriented<float>(P, Q, a, b, cc, d);
        if (si * s2 == -1) v = -v;
        inter += v;
      }
    }
  }
  const float uni = T.rowArea[rl] + T.colArea[cl] - inter;
  if (GUARD) { if (uni == 0.f) return (inter + 1.f) / (uni + 1.f); }
  return inter / uni;
}

}  // namespace orp_tile
