// orp_norm.hip -- fused normalisation + activation passes of the dense head / backbone for gfx950 (inference).
//
// The reference runs conv -> GroupNorm(32) -> ReLU six times per FPN level in the head towers
// (mmdet/models/anchor_heads/orientedreppoints_head.py:91-113, mmdet/ops/conv_module.py:130-140) and
// conv -> BatchNorm(eval) -> (+identity) -> ReLU in every ResNet bottleneck (mmdet/models/backbones/resnet.py:133-170)
// as separate framework kernels.  Profiled on MI355X (round 1, profiles/r01_bench_v2_step.txt) the stock GroupNorm
// costs 1.29 ms / image: 32 groups x B = 32 workgroups for a 16.8 MB tensor, then a second pass, then the ReLU pass.
// These are bandwidth ops: here they are one read-only statistics pass plus ONE read-modify-write pass, all FPN
// levels in one launch.
//
//   orp_groupnorm_act_multi : y = relu?((x - mean_g) * rstd_g * gamma[c] + beta[c]), NCHW, statistics per (image, group)
//       pass 1: one workgroup per 4096-float chunk of a group's contiguous span -> (mean, M2) partials (the chunk is
//               held in registers, so M2 is taken around the chunk mean: no E[x^2] - mean^2 cancellation);
//       pass 2: every workgroup merges its group's partials with the parallel-variance formula (<= a few dozen pairs),
//               then normalises + scales + activates its own chunk with float4 traffic.
//   orp_affine_act          : y = relu?(x * scale[c] + shift[c] (+ residual)) -- eval-mode BatchNorm folded to a
//               per-channel affine, fused with the bottleneck's residual add and ReLU; in place allowed.
//   orp_bias_act_multi      : y = relu?(x + bias[c] (+ residual)), y2 = y - sub[c], all FPN levels in one launch -- the
//               bias / ReLU / `+ pts_out_init` / `- dcn_base_offset` passes around the head's output convolutions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 4096;            // floats per workgroup (16 per thread)
constexpr int kMaxLevels = 8;              // orp_bias_act_multi
constexpr int kGnMaxLevels = 16;           // GroupNorm: both towers' five levels in one launch pair

struct GnLevel {
  const float* x; float* y;
  const float* gamma; const float* beta;     // this tensor's affine parameters
  int hw;                 // H*W
  int cpg;                // chunks per (image, group)
  int chunk0;             // first chunk of this level
};
struct GnParams {
  GnLevel lv[kGnMaxLevels];
  int nlev, B, C, G;
  float eps; int relu;
  float2* partial;        // [total_chunks] (mean, M2)
  float2* stats;          // training: [nlev][B * G] (mean, rstd) per (image, group): level i, image b, group g at (i * B + b) * G + g; or NULL
  // backward (orp_groupnorm_act_multi_backward)
  const float* dy[kGnMaxLevels];
  float* dx[kGnMaxLevels];
  float4* bpart;          // [total_chunks][C / G] (sum dy', sum dy' xhat) per channel of the chunk's group, chunk-local
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                       // red may still be read from a previous call
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// chunk id -> level, (image, group), chunk-in-span; returns the span geometry
struct ChunkGeom { int lvl, bg, k, span, n0, n; };
__device__ __forceinline__ ChunkGeom locate(const GnParams& P, int chunk) {
  ChunkGeom g;
  g.lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) g.lvl = i;
  const GnLevel& L = P.lv[g.lvl];
  const int id = chunk - L.chunk0;
  g.bg = id / L.cpg; g.k = id - g.bg * L.cpg;
  g.span = (P.C / P.G) * L.hw;
  g.n0 = g.k * kChunk;
  g.n = min(kChunk, g.span - g.n0);
  return g;
}

__global__ void __launch_bounds__(kThreads)
gn_stats_kernel(const GnParams P) {
  __shared__ float red[4];
  const ChunkGeom g = locate(P, blockIdx.x);
  const float* src = P.lv[g.lvl].x + (size_t)g.bg * g.span + g.n0;
  float v[16];
  const bool vec = ((g.span & 3) == 0);
  if (vec) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = (threadIdx.x + q * kThreads) * 4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < g.n) t = *reinterpret_cast<const float4*>(src + e);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; q++) { const int e = threadIdx.x + q * kThreads; v[q] = (e < g.n) ? src[e] : 0.f; }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; q++) s += v[q];
  const float mean = block_sum(s, red) / (float)g.n;
  float m2 = 0.f;
  if (vec) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = (threadIdx.x + q * kThreads) * 4;
      if (e < g.n) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const float d = v[4 * q + u] - mean; m2 += d * d; }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; q++) { const int e = threadIdx.x + q * kThreads; if (e < g.n) { const float d = v[q] - mean; m2 += d * d; } }
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) P.partial[blockIdx.x] = make_float2(mean, m2);
}

__global__ void __launch_bounds__(kThreads)
gn_apply_kernel(const GnParams P) {
  __shared__ float red[4];
  const ChunkGeom g = locate(P, blockIdx.x);
  const GnLevel& L = P.lv[g.lvl];
  // merge this group's partials (Chan et al.): mean = sum n_k mean_k / N ; M2 = sum M2_k + n_k (mean_k - mean)^2
  const float2* part = P.partial + L.chunk0 + (size_t)g.bg * L.cpg;
  float sm = 0.f;
  for (int k = threadIdx.x; k < L.cpg; k += kThreads) {
    const int nk = min(kChunk, g.span - k * kChunk);
    sm += (float)nk * part[k].x;
  }
  const float mean = block_sum(sm, red) / (float)g.span;
  float m2 = 0.f;
  for (int k = threadIdx.x; k < L.cpg; k += kThreads) {
    const int nk = min(kChunk, g.span - k * kChunk);
    const float2 p = part[k];
    const float d = p.x - mean;
    m2 += p.y + (float)nk * d * d;
  }
  const float var = block_sum(m2, red) / (float)g.span;
  const float rstd = rsqrtf(var + P.eps);
  if (P.stats && g.k == 0 && threadIdx.x == 0) P.stats[(size_t)g.lvl * P.B * P.G + g.bg] = make_float2(mean, rstd);

  const int cg = P.C / P.G;
  const int grp = g.bg % P.G;
  const float* src = L.x + (size_t)g.bg * g.span + g.n0;
  float* dst = L.y + (size_t)g.bg * g.span + g.n0;
  const bool vec = ((L.hw & 3) == 0);
  if (vec) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int e = (threadIdx.x + q * kThreads) * 4;
      if (e < g.n) {
        const int c = grp * cg + (g.n0 + e) / L.hw;       // 4 consecutive elements share a channel (hw % 4 == 0)
        const float a = rstd * L.gamma[c], b = L.beta[c] - mean * a;
        float4 t = *reinterpret_cast<const float4*>(src + e);
        t.x = t.x * a + b; t.y = t.y * a + b; t.z = t.z * a + b; t.w = t.w * a + b;
        if (P.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
        *reinterpret_cast<float4*>(dst + e) = t;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int e = threadIdx.x + q * kThreads;
      if (e < g.n) {
        const int c = grp * cg + (g.n0 + e) / L.hw;
        const float a = rstd * L.gamma[c], b = L.beta[c] - mean * a;
        float t = src[e] * a + b;
        if (P.relu) t = fmaxf(t, 0.f);
        dst[e] = t;
      }
    }
  }
}

// Pass 2 of orp_groupnorm_act_multi_nhwc: the same normalise + scale + activate as gn_apply_kernel, written TRANSPOSED --
// y_nhwc[b, p, c] -- through a 32 x 33 LDS tile (32 channels = 4 groups x 32 positions per workgroup: 128-byte rows on both
// sides), and optionally also in place / NCHW (the regression tower's last layer feeds a plain convolution AND the
// DeformConv).  The head's DeformConv reads its input channels-last; this replaces the separate nchw_to_nhwc launch (one
// more read + write of every tower output per image).  Same arithmetic as gn_apply_kernel: identical values.
struct GnNhwc {
  float* out[kGnMaxLevels];       // [B, hw, C] per level
  float* nchw[kGnMaxLevels];      // nullptr, or the NCHW output (may alias x)
  int bx0[kGnMaxLevels + 1];      // first blockIdx.x of each level (32-position tiles)
};
__global__ void __launch_bounds__(kThreads)
gn_apply_nhwc_kernel(const GnParams P, const GnNhwc T) {
  __shared__ float tile[32][33];
  __shared__ float2 sStat[32];                    // (mean, rstd) of the tile's groups (32 channels / (C / G) <= 32 groups)
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if ((int)blockIdx.x >= T.bx0[i]) lvl = i;
  const GnLevel& L = P.lv[lvl];
  const int hw = L.hw, cg = P.C / P.G;
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = ((int)blockIdx.x - T.bx0[lvl]) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int ngrp = (32 + cg - 1) / cg;            // groups touched by this channel tile (c0 is a multiple of cg: cg | 32)
  const int span = cg * hw;
  __shared__ float red[4];
  for (int gi = 0; gi < ngrp; gi++) {
    const int grp = c0 / cg + gi;
    if (grp >= P.G) break;                          // (block-uniform)
    // merge this group's partials (Chan et al.) with gn_apply_kernel's own statements and reduction order: the same bits
    const float2* part = P.partial + L.chunk0 + (size_t)(b * P.G + grp) * L.cpg;
    float sm = 0.f;
    for (int k = threadIdx.x; k < L.cpg; k += kThreads) {
      const int nk = min(kChunk, span - k * kChunk);
      sm += (float)nk * part[k].x;
    }
    const float mean = block_sum(sm, red) / (float)span;
    float m2 = 0.f;
    for (int k = threadIdx.x; k < L.cpg; k += kThreads) {
      const int nk = min(kChunk, span - k * kChunk);
      const float2 pk = part[k];
      const float d = pk.x - mean;
      m2 += pk.y + (float)nk * d * d;
    }
    const float var = block_sum(m2, red) / (float)span;
    if (threadIdx.x == 0) sStat[gi] = make_float2(mean, rsqrtf(var + P.eps));
  }
  __syncthreads();
  const float* src = L.x + (size_t)b * P.C * hw;
  float* nchw = T.nchw[lvl] ? T.nchw[lvl] + (size_t)b * P.C * hw : nullptr;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    float t = 0.f;
    if (c < P.C && p < hw) {
      const float2 st = sStat[r / cg];
      const float a = st.y * L.gamma[c], bb = L.beta[c] - st.x * a;
      t = src[(size_t)c * hw + p] * a + bb;
      if (P.relu) t = fmaxf(t, 0.f);
      if (nchw) nchw[(size_t)c * hw + p] = t;
    }
    tile[r][tx] = t;
  }
  __syncthreads();
  float* dst = T.out[lvl] + (size_t)b * hw * P.C;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < hw && c < P.C) dst[(size_t)p * P.C + c] = tile[tx][r];
  }
}

// y = act(x * scale[c] + shift[c] (+ residual)), NCHW.  grid.y = (image, channel) plane, so the per-channel constants are
// block-uniform scalars and no integer division sits in the element loop; float4 traffic when HW % 4 == 0.
__global__ void __launch_bounds__(kThreads)
affine_act_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ scale,
                  const float* __restrict__ shift, float* __restrict__ y, int C, int hw, int relu) {
  const int plane = blockIdx.y;                       // b * C + c
  const int c = plane % C;
  const float a = scale[c], b = shift[c];
  const size_t base = (size_t)plane * hw;
  if ((hw & 3) == 0) {
    const int hw4 = hw >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
    float4* y4 = reinterpret_cast<float4*>(y + base);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
      float4 t = x4[i];
      t.x = t.x * a + b; t.y = t.y * a + b; t.z = t.z * a + b; t.w = t.w * a + b;
      if (r4) { const float4 r = r4[i]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
      if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      y4[i] = t;
    }
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
      float t = x[base + i] * a + b;
      if (res) t += res[base + i];
      if (relu) t = fmaxf(t, 0.f);
      y[base + i] = t;
    }
  }
}

// y = act(x + bias[c] (+ residual)), optionally y2 = y - sub[c]; NCHW, several tensors (FPN levels) per launch.  The
// head's output convolutions run without their bias and this launch adds it for all five levels at once, together with
// what follows in OrientedRepPointsHead.forward_single (head :156-170): ReLU, `+ pts_out_init`, `- dcn_base_offset`.
// Operation order per element is the framework's: fl(x + b), then fl(. + r), then max(., 0), then fl(. - s).
struct BiasLevel {
  const float* x; const float* res; float* y; float* y2;
  int hw;
  int bx0;                // first blockIdx.x of this level
};
struct BiasParams {
  BiasLevel lv[kMaxLevels];
  int nlev, C, relu;
  const float* bias; const float* sub;
};
__global__ void __launch_bounds__(kThreads)
bias_act_multi_kernel(const BiasParams P) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxLevels; i++) l = (i < P.nlev && (int)blockIdx.x >= P.lv[i].bx0) ? i : l;
  const BiasLevel L = P.lv[l];
  const int plane = blockIdx.y;                       // b * C + c
  const int c = plane % P.C;
  const float b = P.bias ? P.bias[c] : 0.f;
  const float sb = P.sub ? P.sub[c] : 0.f;
  const size_t base = (size_t)plane * L.hw;
  const int bx = blockIdx.x - L.bx0;
  const int nbx = ((l + 1 < P.nlev) ? P.lv[l + 1].bx0 : (int)gridDim.x) - L.bx0;
  if ((L.hw & 3) == 0) {
    const int hw4 = L.hw >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(L.x + base);
    const float4* r4 = L.res ? reinterpret_cast<const float4*>(L.res + base) : nullptr;
    float4* y4 = reinterpret_cast<float4*>(L.y + base);
    float4* z4 = L.y2 ? reinterpret_cast<float4*>(L.y2 + base) : nullptr;
    for (int i = bx * kThreads + threadIdx.x; i < hw4; i += nbx * kThreads) {
      float4 t = x4[i];
      t.x += b; t.y += b; t.z += b; t.w += b;
      if (r4) { const float4 r = r4[i]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
      if (P.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      y4[i] = t;
      if (z4) { t.x -= sb; t.y -= sb; t.z -= sb; t.w -= sb; z4[i] = t; }
    }
  } else {
    for (int i = bx * kThreads + threadIdx.x; i < L.hw; i += nbx * kThreads) {
      float t = L.x[base + i] + b;
      if (L.res) t += L.res[base + i];
      if (P.relu) t = fmaxf(t, 0.f);
      L.y[base + i] = t;
      if (L.y2) L.y2[base + i] = t - sb;
    }
  }
}

// ---- GroupNorm (+ ReLU) backward, all levels in one launch pair -------------------------------------------------------------
// y = relu?(xhat * gamma_c + beta_c), xhat = (x - mean_g) * rstd_g.  With dy' = dy * [y > 0]:
//   dgamma_c = sum_{b,hw} dy' xhat,  dbeta_c = sum_{b,hw} dy'
//   dx = rstd_g * (dy' gamma_c - (A + xhat * Bq) / span),  A = sum_g dy' gamma_c,  Bq = sum_g dy' gamma_c xhat
// pass 1 (one workgroup per 4096-float chunk, as the forward): per channel of the chunk's group the chunk-local sums
//         (sum dy', sum dy' xhat) -> bpart[chunk][cg]; pass 2: every workgroup adds its group's partials in chunk order
//         (fixed order: reproducible), forms A and Bq and writes dx for its chunk; pass 3: dgamma / dbeta per parameter
//         set by a fixed-order sum over tensors, images and chunks.
__global__ void __launch_bounds__(kThreads)
gn_bwd_partial_kernel(const GnParams P) {
  __shared__ float red[4];
  const ChunkGeom g = locate(P, blockIdx.x);
  const GnLevel& L = P.lv[g.lvl];
  const int cg = P.C / P.G;
  const float2 st = P.stats[(size_t)g.lvl * P.B * P.G + g.bg];
  const float mean = st.x, rstd = st.y;
  const float* x = L.x + (size_t)g.bg * g.span + g.n0;
  const float* y = L.y + (size_t)g.bg * g.span + g.n0;             // the forward's output: the ReLU mask is read, not recomputed
  const float* dy = P.dy[g.lvl] + (size_t)g.bg * g.span + g.n0;
  // the chunk's elements e in [0, n): channel (g.n0 + e) / hw of the group; accumulate per channel (<= cg of them)
  const int c_first = g.n0 / L.hw, c_last = (g.n0 + g.n - 1) / L.hw;
  for (int cl = c_first; cl <= c_last; cl++) {
    const int e0 = max(cl * L.hw - g.n0, 0), e1 = min((cl + 1) * L.hw - g.n0, g.n);
    float s1 = 0.f, s2 = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += kThreads) {
      const float xv = x[e];
      float d = dy[e];
      if (P.relu && !(y[e] > 0.f)) d = 0.f;
      s1 += d; s2 += d * ((xv - mean) * rstd);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) P.bpart[(size_t)blockIdx.x * cg + cl] = make_float4(s1, s2, 0.f, 0.f);
  }
  // channels of the group this chunk does not touch: zero entries (every slot is read by pass 2 / 3)
  for (int cl = threadIdx.x; cl < cg; cl += kThreads)
    if (cl < c_first || cl > c_last) P.bpart[(size_t)blockIdx.x * cg + cl] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(kThreads)
gn_bwd_apply_kernel(const GnParams P) {
  __shared__ float red[4];
  const ChunkGeom g = locate(P, blockIdx.x);
  const GnLevel& L = P.lv[g.lvl];
  const int cg = P.C / P.G, grp = g.bg % P.G;
  const float2 st = P.stats[(size_t)g.lvl * P.B * P.G + g.bg];
  const float mean = st.x, rstd = st.y;
  // the group's partials: thread t takes entries t, t + 256, ... (chunk-major, channel-minor), then a fixed reduction tree
  float pa = 0.f, pb = 0.f;
  {
    const float4* part = P.bpart + ((size_t)L.chunk0 + (size_t)g.bg * L.cpg) * cg;
    for (int i = threadIdx.x; i < L.cpg * cg; i += kThreads) {
      const float4 p = part[i];
      const float gm = L.gamma[grp * cg + i % cg];
      pa += gm * p.x; pb += gm * p.y;
    }
  }
  const float inv = 1.f / (float)g.span;
  const float A = block_sum(pa, red) * inv;
  const float Bq = block_sum(pb, red) * inv;
  const float* x = L.x + (size_t)g.bg * g.span + g.n0;
  const float* y = L.y + (size_t)g.bg * g.span + g.n0;
  const float* dy = P.dy[g.lvl] + (size_t)g.bg * g.span + g.n0;
  float* dx = P.dx[g.lvl] + (size_t)g.bg * g.span + g.n0;
  for (int e = threadIdx.x; e < g.n; e += kThreads) {
    const int c = grp * cg + (g.n0 + e) / L.hw;
    const float gm = L.gamma[c];
    const float xv = x[e];
    float d = dy[e];
    if (P.relu && !(y[e] > 0.f)) d = 0.f;
    const float xh = (xv - mean) * rstd;
    dx[e] = rstd * (d * gm - A - xh * Bq);
  }
}

// dgamma / dbeta of ONE parameter set: the tensors whose gamma pointer equals `gamma`, fixed order (tensor, image, chunk)
__global__ void gn_bwd_param_kernel(const GnParams P, const float* gamma, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.C) return;
  const int cg = P.C / P.G, grp = c / cg, cl = c - grp * cg;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < P.nlev; i++) {
    const GnLevel& L = P.lv[i];
    if (L.gamma != gamma) continue;
    for (int b = 0; b < P.B; b++) {
      const float4* part = P.bpart + ((size_t)L.chunk0 + (size_t)(b * P.G + grp) * L.cpg) * cg;
      for (int k = 0; k < L.cpg; k++) { const float4 p = part[(size_t)k * cg + cl]; s1 += p.x; s2 += p.y; }
    }
  }
  dbeta[c] = s1; dgamma[c] = s2;
}

// ---- GroupNorm (+ ReLU) of channels-last tensors [B][HW][C] (orp_groupnorm_act_multi_cl) ---------------------------------------
// What sits between two orp_conv_split_multi launches (orp_conv_split.hip): the convolution reads and writes channels-last, so
// the normalisation does too.  A chunk = 4096 / C whole positions (rows of C floats): with 1024 % C == 0 every thread's
// float4s all fall on the SAME four channels, i.e. one group -- the per-channel constants are per-thread constants, and the
// chunk's per-group statistics are a sum over a fixed set of threads.
//   pass 1 (gn_cl_stats): (mean, M2 around that mean) of every group over the chunk's positions, the chunk held in registers;
//   pass 2 (gn_cl_merge): one wave per (tensor, image, group): Chan's merge of the chunk partials -> (mean, rstd);
//   pass 3 (gn_cl_apply): y = relu?(x * a[c] + b[c]), a = rstd * gamma, b = beta - mean * a -- one float4 pass, in place allowed.
struct GnClLevel {
  const float* x; float* y;
  const float* gamma; const float* beta;
  int hw;                 // H*W
  int cpi;                // chunks per image
  int chunk0;             // first chunk of this tensor
};
struct GnClParams {
  GnClLevel lv[kGnMaxLevels];
  int nlev, B, C, G;
  float eps; int relu;
  float2* partial;        // [total_chunks][G] (mean, M2)
  float2* stats;          // [nlev][B][G] (mean, rstd)
  // optional: an upper bound of max |y| over the tensors of a slot, as float bits (what the fp16-pieces convolution that reads
  // y scales its samples by -- orp_conv_split_multi's amax_in -- so that it needs no pass of its own over y)
  float* pmax;            // [total_chunks][G] max |x| of the chunk's group (same layout as partial), or nullptr
  unsigned* amax;         // [nslots], zeroed by the entry
  int slot[kGnMaxLevels];
};

struct ClGeom { int lvl, b, p0, np; };
__device__ __forceinline__ ClGeom cl_locate(const GnClParams& P, int chunk) {
  ClGeom g;
  g.lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) g.lvl = i;
  const GnClLevel& L = P.lv[g.lvl];
  const int id = chunk - L.chunk0, ppc = kChunk / P.C;
  g.b = id / L.cpi;
  g.p0 = (id - g.b * L.cpi) * ppc;
  g.np = min(ppc, L.hw - g.p0);
  return g;
}

// sum over the threads that hold this thread's group, in a fixed order: thread t holds channels 4 * (t % tpc) .. + 3 of rows
// t / tpc, t / tpc + 256 / tpc, ...; its group's members are rep * tpc + grp * tg + j  (rep < 256 / tpc, j < tg)
__device__ __forceinline__ float group_total(float v, float* sh, int tpc, int tg, int grp) {
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
  for (int rep = 0; rep < kThreads / tpc; rep++)
    for (int j = 0; j < tg; j++) s += sh[rep * tpc + grp * tg + j];
  return s;
}

__device__ __forceinline__ float group_max(float v, float* sh, int tpc, int tg, int grp) {
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
  for (int rep = 0; rep < kThreads / tpc; rep++)
    for (int j = 0; j < tg; j++) s = fmaxf(s, sh[rep * tpc + grp * tg + j]);
  return s;
}

__global__ void __launch_bounds__(kThreads)
gn_cl_stats_kernel(const GnClParams P) {
  __shared__ float sh[kThreads];
  const ClGeom g = cl_locate(P, blockIdx.x);
  const GnClLevel& L = P.lv[g.lvl];
  const int n = g.np * P.C;
  const float* src = L.x + ((size_t)g.b * L.hw + g.p0) * P.C;
  const int tpc = P.C >> 2, cg = P.C / P.G, tg = cg >> 2;
  const int grp = (threadIdx.x % tpc) / tg;
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = (threadIdx.x + q * kThreads) * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n) t = *reinterpret_cast<const float4*>(src + e);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; q++) s += v[q];
  const float mean = group_total(s, sh, tpc, tg, grp) / (float)(g.np * cg);
  float m2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = (threadIdx.x + q * kThreads) * 4;
    if (e < n) {
#pragma unroll
      for (int u = 0; u < 4; u++) { const float d = v[4 * q + u] - mean; m2 += d * d; }
    }
  }
  m2 = group_total(m2, sh, tpc, tg, grp);
  // partials of one (tensor, image, group) are contiguous over the image's chunks: [tensor][image][group][chunk]
  const size_t slot_ = (size_t)L.chunk0 * P.G + ((size_t)g.b * P.G + grp) * L.cpi + g.p0 / (kChunk / P.C);
  if (threadIdx.x < tpc && threadIdx.x % tg == 0) P.partial[slot_] = make_float2(mean, m2);
  if (P.pmax) {                                            // (block-uniform)
    float am = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) am = fmaxf(am, fabsf(v[q]));          // (elements past the chunk's end were loaded as 0)
    am = group_max(am, sh, tpc, tg, grp);
    if (threadIdx.x < tpc && threadIdx.x % tg == 0) P.pmax[slot_] = am;
  }
}

// one workgroup per (tensor, image, group)
__global__ void __launch_bounds__(kThreads)
gn_cl_merge_kernel(const GnClParams P) {
  __shared__ float red[4];
  const int grp = blockIdx.x, b = blockIdx.y, lvl = blockIdx.z;
  const GnClLevel& L = P.lv[lvl];
  const int ppc = kChunk / P.C, cg = P.C / P.G;
  const float2* part = P.partial + (size_t)L.chunk0 * P.G + ((size_t)b * P.G + grp) * L.cpi;
  const float total = (float)L.hw * (float)cg;
  float sm = 0.f;
  for (int k = threadIdx.x; k < L.cpi; k += kThreads) {
    const int nk = min(ppc, L.hw - k * ppc) * cg;
    sm += (float)nk * part[k].x;
  }
  const float mean = block_sum(sm, red) / total;
  float m2 = 0.f;
  for (int k = threadIdx.x; k < L.cpi; k += kThreads) {
    const int nk = min(ppc, L.hw - k * ppc) * cg;
    const float2 pk = part[k];
    const float d = pk.x - mean;
    m2 += pk.y + (float)nk * d * d;
  }
  const float var = block_sum(m2, red) / total;
  const float rstd = rsqrtf(var + P.eps);
  if (threadIdx.x == 0) P.stats[((size_t)lvl * P.B + b) * P.G + grp] = make_float2(mean, rstd);
  if (P.pmax) {
    // |y| = |(x - mean) rstd gamma_c + beta_c| <= (max |x| + |mean|) rstd max |gamma| + max |beta| over the group's channels
    const float* pm = P.pmax + (size_t)L.chunk0 * P.G + ((size_t)b * P.G + grp) * L.cpi;
    float xm = 0.f;
    for (int k = threadIdx.x; k < L.cpi; k += kThreads) xm = fmaxf(xm, pm[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xm = fmaxf(xm, __shfl_xor(xm, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = xm;
    __syncthreads();
    if (threadIdx.x == 0) {
      xm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      float gm = 0.f, bm = 0.f;
      for (int c = grp * cg; c < (grp + 1) * cg; c++) { gm = fmaxf(gm, fabsf(L.gamma[c])); bm = fmaxf(bm, fabsf(L.beta[c])); }
      const float bound = (xm + fabsf(mean)) * rstd * gm + bm;
      atomicMax(P.amax + P.slot[lvl], __float_as_uint(bound * 1.0001f));          // (a hair above the rounding of the bound itself)
    }
  }
}

__global__ void __launch_bounds__(kThreads)
gn_cl_apply_kernel(const GnClParams P) {
  const ClGeom g = cl_locate(P, blockIdx.x);
  const GnClLevel& L = P.lv[g.lvl];
  const int n = g.np * P.C;
  const size_t base = ((size_t)g.b * L.hw + g.p0) * P.C;
  const int tpc = P.C >> 2, cg = P.C / P.G;
  const int c0 = (threadIdx.x % tpc) * 4;
  const float2 st = P.stats[((size_t)g.lvl * P.B + g.b) * P.G + c0 / cg];
  const float4 ga = *reinterpret_cast<const float4*>(L.gamma + c0), be = *reinterpret_cast<const float4*>(L.beta + c0);
  const float a0 = st.y * ga.x, a1 = st.y * ga.y, a2 = st.y * ga.z, a3 = st.y * ga.w;
  const float b0 = be.x - st.x * a0, b1 = be.y - st.x * a1, b2 = be.z - st.x * a2, b3 = be.w - st.x * a3;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = (threadIdx.x + q * kThreads) * 4;
    if (e < n) {
      float4 t = *reinterpret_cast<const float4*>(L.x + base + e);
      t.x = t.x * a0 + b0; t.y = t.y * a1 + b1; t.z = t.z * a2 + b2; t.w = t.w * a3 + b3;
      if (P.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      *reinterpret_cast<float4*>(L.y + base + e) = t;
    }
  }
}

// y = relu?(x * a[c] + b[c]) with per-(tensor, image, channel) coefficients (orp_conv_split_gn_finish): the LAST normalisation of a
// tower, materialised for the consumers that cannot apply it on the fly; block 0 also folds the per-group bounds of max |y| into one
// range word per tensor set
__global__ void __launch_bounds__(kThreads)
affine_cl_kernel(const GnClParams P, const float2* __restrict__ coef, const unsigned* __restrict__ bound_in, unsigned* __restrict__ slot_out,
                 int nsets, int per_set) {
  const ClGeom g = cl_locate(P, blockIdx.x);
  const GnClLevel& L = P.lv[g.lvl];
  const int n = g.np * P.C;
  const size_t base = ((size_t)g.b * L.hw + g.p0) * P.C;
  const int tpc = P.C >> 2;
  const int c0 = (threadIdx.x % tpc) * 4;
  const float2* cf = coef + ((size_t)g.lvl * P.B + g.b) * P.C + c0;
  const float2 k0 = cf[0], k1 = cf[1], k2 = cf[2], k3 = cf[3];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = (threadIdx.x + q * kThreads) * 4;
    if (e < n) {
      float4 t = *reinterpret_cast<const float4*>(L.x + base + e);
      t.x = fmaf(t.x, k0.x, k0.y); t.y = fmaf(t.y, k1.x, k1.y); t.z = fmaf(t.z, k2.x, k2.y); t.w = fmaf(t.w, k3.x, k3.y);
      if (P.relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      *reinterpret_cast<float4*>(L.y + base + e) = t;
    }
  }
  if (blockIdx.x == 0 && bound_in && slot_out) {
    __shared__ unsigned red[4];
    for (int s_ = 0; s_ < nsets; s_++) {
      unsigned m = 0u;
      for (int i = threadIdx.x; i < per_set; i += kThreads) m = max(m, bound_in[(size_t)s_ * per_set + i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
      __syncthreads();
      if (threadIdx.x == 0) slot_out[s_] = max(max(red[0], red[1]), max(red[2], red[3]));
    }
  }
}

// fills P's levels; returns the number of chunks, -1 on bad arguments, -2 when a tensor is too large
int fill_cl(const orp_norm_level* levels, int nlevels, int batch, int channels, int groups, GnClParams& P) {
  if (!levels || nlevels <= 0 || nlevels > kGnMaxLevels || batch <= 0 || batch > 65535 || channels <= 0 || groups <= 0 ||
      channels % groups != 0 || 1024 % channels != 0 || (channels / groups) % 4 != 0)
    return -1;
  P.nlev = nlevels; P.B = batch; P.C = channels; P.G = groups;
  const int ppc = kChunk / channels;
  long chunks = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_norm_level& lv = levels[i];
    if (!lv.input || !lv.output || lv.height <= 0 || lv.width <= 0) return -1;
    const long hw = (long)lv.height * lv.width;
    if (hw >= (1L << 31) / channels) return -2;
    GnClLevel& L = P.lv[i];
    L.x = lv.input; L.y = lv.output; L.gamma = nullptr; L.beta = nullptr;
    L.hw = (int)hw; L.cpi = (int)((hw + ppc - 1) / ppc); L.chunk0 = (int)chunks;
    chunks += (long)batch * L.cpi;
    if (chunks >= (1L << 30)) return -2;
  }
  for (int i = nlevels; i < kGnMaxLevels; i++) { P.lv[i] = P.lv[0]; P.lv[i].chunk0 = 0x7fffffff; }
  return (int)chunks;
}


int fill(const orp_norm_level* levels, int nlevels, int batch, int channels, int groups, GnParams& P) {
  if (!levels || nlevels <= 0 || nlevels > kGnMaxLevels || batch <= 0 || channels <= 0 || groups <= 0 ||
      channels % groups)
    return -1;
  int chunks = 0;
  for (int i = 0; i < nlevels; i++) {
    if (!levels[i].input || !levels[i].output || levels[i].height <= 0 || levels[i].width <= 0) return -1;
    GnLevel& L = P.lv[i];
    L.x = levels[i].input; L.y = levels[i].output;
    L.gamma = nullptr; L.beta = nullptr;
    L.hw = levels[i].height * levels[i].width;
    const long span = (long)(channels / groups) * L.hw;
    if (span >= (1L << 30)) return -2;
    L.cpg = (int)((span + kChunk - 1) / kChunk);
    L.chunk0 = chunks;
    chunks += batch * groups * L.cpg;
  }
  for (int i = nlevels; i < kGnMaxLevels; i++) { P.lv[i] = P.lv[0]; P.lv[i].chunk0 = 0x7fffffff; }
  P.nlev = nlevels; P.B = batch; P.C = channels; P.G = groups;
  return chunks;
}

}  // namespace

extern "C" {

size_t orp_groupnorm_workspace_bytes(const orp_norm_level* levels, int nlevels, int batch, int channels, int groups) {
  GnParams P;
  const int chunks = fill(levels, nlevels, batch, channels, groups, P);
  return chunks > 0 ? sizeof(float2) * (size_t)chunks + 256 : 256;
}

int orp_groupnorm_act_multi_ex(const orp_norm_level* levels, const float* const* gammas_host,
                               const float* const* betas_host, int nlevels, int batch, int channels, int groups,
                               float eps, int relu, void* workspace, size_t workspace_bytes, void* stream) {
  GnParams P;
  const int chunks = fill(levels, nlevels, batch, channels, groups, P);
  if (chunks == -2) return ORP_ETOOBIG;
  if (chunks <= 0 || !gammas_host || !betas_host) return ORP_EINVAL;
  if (!workspace || workspace_bytes < sizeof(float2) * (size_t)chunks) return ORP_EWORKSPACE;
  for (int i = 0; i < nlevels; i++) {
    if (!gammas_host[i] || !betas_host[i]) return ORP_EINVAL;
    P.lv[i].gamma = gammas_host[i]; P.lv[i].beta = betas_host[i];
  }
  P.eps = eps; P.relu = relu;
  P.partial = reinterpret_cast<float2*>(workspace);
  P.stats = nullptr;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_groupnorm_act_multi_nhwc(const orp_norm_level* levels, const float* const* gammas_host,
                                 const float* const* betas_host, float* const* nhwc_out_host, int nlevels, int batch,
                                 int channels, int groups, float eps, int relu, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  GnParams P;
  if (!levels || nlevels <= 0 || nlevels > kGnMaxLevels) return ORP_EINVAL;
  orp_norm_level tmp[kGnMaxLevels];
  for (int i = 0; i < nlevels; i++) { tmp[i] = levels[i]; if (!tmp[i].output) tmp[i].output = const_cast<float*>(tmp[i].input); }
  const int chunks = fill(tmp, nlevels, batch, channels, groups, P);
  if (chunks == -2) return ORP_ETOOBIG;
  if (chunks <= 0 || !gammas_host || !betas_host || !nhwc_out_host) return ORP_EINVAL;
  if (channels % 32 != 0 || 32 % (channels / groups) != 0 || batch > 65535) return ORP_EINVAL;
  if (!workspace || workspace_bytes < sizeof(float2) * (size_t)chunks) return ORP_EWORKSPACE;
  GnNhwc T;
  int bx = 0;
  for (int i = 0; i < nlevels; i++) {
    if (!gammas_host[i] || !betas_host[i] || !nhwc_out_host[i]) return ORP_EINVAL;
    P.lv[i].gamma = gammas_host[i]; P.lv[i].beta = betas_host[i];
    T.out[i] = nhwc_out_host[i];
    T.nchw[i] = levels[i].output;                       // NULL: the channels-last tensor is the only output
    T.bx0[i] = bx;
    bx += (P.lv[i].hw + 31) / 32;
  }
  for (int i = nlevels; i <= kGnMaxLevels; i++) T.bx0[i] = 0x7fffffff;
  for (int i = nlevels; i < kGnMaxLevels; i++) { T.out[i] = T.out[0]; T.nchw[i] = nullptr; }
  P.eps = eps; P.relu = relu;
  P.partial = reinterpret_cast<float2*>(workspace);
  P.stats = nullptr;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipLaunchKernelGGL(gn_apply_nhwc_kernel, dim3(bx, channels / 32, batch), dim3(kThreads), 0, st, P, T);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

static size_t gn_cl_stat_offset(size_t chunks, int groups) { return (sizeof(float2) * chunks * groups + 255) & ~(size_t)255; }

size_t orp_groupnorm_cl_workspace_bytes(const orp_norm_level* levels, int nlevels, int batch, int channels, int groups) {
  GnClParams P;
  const int chunks = fill_cl(levels, nlevels, batch, channels, groups, P);
  if (chunks <= 0) return 0;
  // partials | (mean, rstd) per (tensor, image, group) | per-chunk group maxima of the _amax entry
  return ((gn_cl_stat_offset(chunks, groups) + sizeof(float2) * (size_t)nlevels * batch * groups + 255) & ~(size_t)255) +
         sizeof(float) * (size_t)chunks * groups;
}

static int gn_cl_impl(const orp_norm_level* levels, const float* const* gammas_host, const float* const* betas_host,
                      int nlevels, int batch, int channels, int groups, float eps, int relu, const int* slots_host,
                      uint32_t* amax_out, int nslots, void* workspace, size_t workspace_bytes, void* stream) {
  GnClParams P;
  const int chunks = fill_cl(levels, nlevels, batch, channels, groups, P);
  if (chunks == -2) return ORP_ETOOBIG;
  if (chunks <= 0 || !gammas_host || !betas_host) return ORP_EINVAL;
  if (amax_out && (!slots_host || nslots <= 0)) return ORP_EINVAL;
  const size_t stat_off = gn_cl_stat_offset(chunks, groups);
  const size_t pmax_off = (stat_off + sizeof(float2) * (size_t)nlevels * batch * groups + 255) & ~(size_t)255;
  const size_t need = amax_out ? pmax_off + sizeof(float) * (size_t)chunks * groups : pmax_off;
  if (!workspace || workspace_bytes < need) return ORP_EWORKSPACE;
  for (int i = 0; i < nlevels; i++) {
    if (!gammas_host[i] || !betas_host[i]) return ORP_EINVAL;
    if (amax_out && (slots_host[i] < 0 || slots_host[i] >= nslots)) return ORP_EINVAL;
    P.lv[i].gamma = gammas_host[i]; P.lv[i].beta = betas_host[i];
    P.slot[i] = amax_out ? slots_host[i] : 0;
  }
  for (int i = nlevels; i < kGnMaxLevels; i++) P.slot[i] = 0;
  P.eps = eps; P.relu = relu;
  P.partial = reinterpret_cast<float2*>(workspace);
  P.stats = reinterpret_cast<float2*>(reinterpret_cast<char*>(workspace) + stat_off);
  P.pmax = amax_out ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pmax_off) : nullptr;
  P.amax = amax_out;
  hipStream_t st = (hipStream_t)stream;
  if (amax_out) {
    const hipError_t me = orp::fill_async(amax_out, 0, sizeof(unsigned) * (size_t)(nslots), st);
    if (me != hipSuccess) return (int)me;
  }
  hipLaunchKernelGGL(gn_cl_stats_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipLaunchKernelGGL(gn_cl_merge_kernel, dim3(groups, batch, nlevels), dim3(kThreads), 0, st, P);
  hipLaunchKernelGGL(gn_cl_apply_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_affine_act_multi_cl(const orp_norm_level* levels, int nlevels, int batch, int channels, const float* coef, int relu,
                            const uint32_t* bound_in, int nsets, int per_set, uint32_t* slot_out, void* stream) {
  GnClParams P;
  const int chunks = fill_cl(levels, nlevels, batch, channels, 1, P);
  if (chunks == -2) return ORP_ETOOBIG;
  if (chunks <= 0 || !coef || (bound_in && (!slot_out || nsets <= 0 || per_set <= 0))) return ORP_EINVAL;
  P.eps = 0.f; P.relu = relu ? 1 : 0; P.partial = nullptr; P.stats = nullptr; P.pmax = nullptr; P.amax = nullptr;
  hipLaunchKernelGGL(affine_cl_kernel, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, P, reinterpret_cast<const float2*>(coef),
                     bound_in, slot_out, nsets, per_set);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_groupnorm_act_multi_cl(const orp_norm_level* levels, const float* const* gammas_host, const float* const* betas_host,
                               int nlevels, int batch, int channels, int groups, float eps, int relu, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return gn_cl_impl(levels, gammas_host, betas_host, nlevels, batch, channels, groups, eps, relu, nullptr, nullptr, 0, workspace,
                    workspace_bytes, stream);
}

int orp_groupnorm_act_multi_cl_amax(const orp_norm_level* levels, const float* const* gammas_host, const float* const* betas_host,
                                    int nlevels, int batch, int channels, int groups, float eps, int relu, const int* slots_host,
                                    uint32_t* amax_out, int nslots, void* workspace, size_t workspace_bytes, void* stream) {
  if (!amax_out) return ORP_EINVAL;
  return gn_cl_impl(levels, gammas_host, betas_host, nlevels, batch, channels, groups, eps, relu, slots_host, amax_out, nslots,
                    workspace, workspace_bytes, stream);
}

int orp_groupnorm_act_multi_train(const orp_norm_level* levels, const float* const* gammas_host,
                                  const float* const* betas_host, int nlevels, int batch, int channels, int groups,
                                  float eps, int relu, float* stats, void* workspace, size_t workspace_bytes, void* stream) {
  GnParams P;
  const int chunks = fill(levels, nlevels, batch, channels, groups, P);
  if (chunks == -2) return ORP_ETOOBIG;
  if (chunks <= 0 || !gammas_host || !betas_host || !stats) return ORP_EINVAL;
  if (!workspace || workspace_bytes < sizeof(float2) * (size_t)chunks) return ORP_EWORKSPACE;
  for (int i = 0; i < nlevels; i++) {
    if (!gammas_host[i] || !betas_host[i]) return ORP_EINVAL;
    P.lv[i].gamma = gammas_host[i]; P.lv[i].beta = betas_host[i];
  }
  P.eps = eps; P.relu = relu;
  P.partial = reinterpret_cast<float2*>(workspace);
  P.stats = reinterpret_cast<float2*>(stats);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

size_t orp_groupnorm_backward_workspace_bytes(const orp_norm_level* levels, int nlevels, int batch, int channels, int groups) {
  GnParams P;
  const int chunks = fill(levels, nlevels, batch, channels, groups, P);
  return chunks > 0 ? sizeof(float4) * (size_t)chunks * (size_t)(channels / groups) + 256 : 256;
}

int orp_groupnorm_act_multi_backward(const orp_norm_level* levels, const float* const* grad_outputs_host,
                                     float* const* grad_inputs_host, const float* const* gammas_host,
                                     const float* const* betas_host, float* const* dgammas_host, float* const* dbetas_host,
                                     int nlevels, int batch, int channels, int groups, int relu, const float* stats,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  GnParams P;
  const int chunks = fill(levels, nlevels, batch, channels, groups, P);
  if (chunks == -2) return ORP_ETOOBIG;
  if (chunks <= 0 || !grad_outputs_host || !grad_inputs_host || !gammas_host || !betas_host || !dgammas_host ||
      !dbetas_host || !stats)
    return ORP_EINVAL;
  const int cg = channels / groups;
  if (!workspace || workspace_bytes < sizeof(float4) * (size_t)chunks * cg) return ORP_EWORKSPACE;
  for (int i = 0; i < nlevels; i++) {
    if (!gammas_host[i] || !betas_host[i] || !grad_outputs_host[i] || !grad_inputs_host[i]) return ORP_EINVAL;
    P.lv[i].gamma = gammas_host[i]; P.lv[i].beta = betas_host[i];
    P.dy[i] = grad_outputs_host[i]; P.dx[i] = grad_inputs_host[i];
  }
  for (int i = nlevels; i < kGnMaxLevels; i++) { P.dy[i] = nullptr; P.dx[i] = nullptr; }
  P.eps = 0.f; P.relu = relu;
  P.partial = nullptr;
  P.stats = reinterpret_cast<float2*>(const_cast<float*>(stats));
  P.bpart = reinterpret_cast<float4*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(chunks), dim3(kThreads), 0, st, P);
  // one launch per distinct parameter set (the tensors sharing a GroupNorm module)
  for (int i = 0; i < nlevels; i++) {
    bool first = true;
    for (int j = 0; j < i; j++) if (gammas_host[j] == gammas_host[i]) first = false;
    if (!first || !dgammas_host[i] || !dbetas_host[i]) continue;
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((channels + 255) / 256), dim3(256), 0, st, P, gammas_host[i], dgammas_host[i],
                       dbetas_host[i]);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_groupnorm_act_multi(const orp_norm_level* levels, int nlevels, int batch, int channels, int groups,
                            const float* gamma, const float* beta, float eps, int relu, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (nlevels <= 0 || nlevels > kGnMaxLevels) return ORP_EINVAL;
  const float* g[kGnMaxLevels]; const float* b[kGnMaxLevels];
  for (int i = 0; i < nlevels; i++) { g[i] = gamma; b[i] = beta; }
  return orp_groupnorm_act_multi_ex(levels, g, b, nlevels, batch, channels, groups, eps, relu, workspace, workspace_bytes,
                                    stream);
}

int orp_affine_act(const float* x, const float* residual, const float* scale, const float* shift, float* y, int batch,
                   int channels, int hw, int relu, void* stream) {
  if (!x || !scale || !shift || !y || batch <= 0 || channels <= 0 || hw <= 0) return ORP_EINVAL;
  if ((long)batch * channels > 65535L * 1024) return ORP_ETOOBIG;
  const int per = ((hw & 3) == 0) ? (hw >> 2) : hw;                 // work items per plane
  int bx = (per + kThreads * 4 - 1) / (kThreads * 4);               // ~4 items per thread
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(affine_act_kernel, dim3(bx, batch * channels), dim3(kThreads), 0, (hipStream_t)stream, x, residual,
                     scale, shift, y, channels, hw, relu);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_bias_act_multi(const orp_bias_level* levels_host, int nlevels, int batch, int channels, const float* bias,
                       const float* sub, int relu, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > kMaxLevels || batch <= 0 || channels <= 0) return ORP_EINVAL;
  if ((long)batch * channels > 65535L) return ORP_ETOOBIG;
  BiasParams P;
  P.nlev = nlevels; P.C = channels; P.relu = relu ? 1 : 0; P.bias = bias; P.sub = sub;
  int bx = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_bias_level& lv = levels_host[i];
    if (!lv.input || !lv.output || lv.height <= 0 || lv.width <= 0 || (lv.output2 && !sub)) return ORP_EINVAL;
    BiasLevel& L = P.lv[i];
    L.x = lv.input; L.res = lv.residual; L.y = lv.output; L.y2 = lv.output2;
    L.hw = lv.height * lv.width;
    L.bx0 = bx;
    const int per = ((L.hw & 3) == 0) ? (L.hw >> 2) : L.hw;
    int nb = (per + kThreads * 4 - 1) / (kThreads * 4);               // ~4 items per thread
    if (nb < 1) nb = 1;
    if (nb > 64) nb = 64;
    bx += nb;
  }
  for (int i = nlevels; i < kMaxLevels; i++) { P.lv[i] = P.lv[0]; P.lv[i].bx0 = 0x7fffffff; }
  hipLaunchKernelGGL(bias_act_multi_kernel, dim3(bx, batch * channels), dim3(kThreads), 0, (hipStream_t)stream, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
