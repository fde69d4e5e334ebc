// orp_conv_small.hip -- 3x3 pad-1 convolution (stride 1 or 2) of the SMALL FPN levels, all of them in one launch (gfx950).
//
// The dense head runs seven 256->256 3x3 convolutions over every FPN level (orientedreppoints_head.py:91-132: three
// cls tower + three reg tower ConvModules and reppoints_pts_init_conv).  On the 128^2 / 64^2 levels the library's
// Winograd kernels are the right tool; on the 32^2 / 16^2 / 8^2 levels (1344 of the 21824 positions of a 1024^2 image)
// the framework issues an im2col + GEMM pair per level -- six launches and ~60 us per layer for 6 % of the positions,
// a quarter of the layer's time.  Here those levels are ONE launch of an exact-fp32 MFMA implicit GEMM that reads the
// NCHW activations as they are (lane = position: 32 consecutive positions are one coalesced 128 B segment per channel,
// so the A operand of v_mfma_f32_32x32x2_f32 comes straight from L2 -- the three levels are 1.4 MB) and writes NCHW:
//   workgroup = 32 positions x 64 output channels, eight waves (two per SIMD), wave w contracts the w-th eighth of
//   K = 9 taps x Cin (the weights are the [tap][c/4][o][4] packing of orp_dcn_pack_weight: one float4 per lane per
//   4 k-steps, no LDS; three chunk pairs of global loads in flight), then the partial tiles are summed through
//   LDS in a fixed order, so the result is deterministic.
// No bias / activation here: GroupNorm+ReLU (orp_groupnorm_act_multi) or the bias pass (orp_bias_act_multi) follow.
// Stride 2 (orp_conv3x3_small_multi_strided): the FPN's extra output levels (mmdet/models/necks/fpn.py:160-174, P6 / P7 =
// stride-2 convolutions of 32^2 / 16^2 maps).  The library's pick for these two shapes on this part is a split-K
// implicit GEMM that accumulates with atomics: 2e-6 run-to-run differences in P6 / P7, i.e. detections that come and go
// at the score threshold; the fixed-order sum here makes the whole inference step bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kMaxLevels = 16;             // both towers' levels
constexpr int kTaps = 9;
constexpr int kTileM = 32;                    // positions per workgroup
constexpr int kTileN = 64;                    // output channels per workgroup (two 32x32 accumulators per wave)

struct ConvLevel {
  const float* x; float* y;
  const float* w3;                            // this tensor's weights, [tap][Cin/4][Cout][4]
  int H, W;                                   // input height / width
  int Ho, Wo, stride;                         // output height / width; 1 or 2
  int tile0;                                  // first position tile of this level
  float* part;                                // ksplit > 1: this level's slice of the first partial image
};
struct ConvParams {
  ConvLevel lv[kMaxLevels];
  int nlev, B, Cin, Cout;
  int ksplit;                                 // grid-level split of K (blockIdx.z): partial images, summed in fixed order
  size_t part_stride;                         // floats between the partial images of consecutive K slices
};

// one K chunk = 8 input channels of one tap = four v_mfma_f32_32x32x2_f32 steps per accumulator
struct Chunk2 { float a[2][4]; float4 b0[2], b1[2]; };     // two consecutive chunks

// kWaves waves per workgroup (2 or 4 per SIMD); wave w contracts the w-th slice of K = 9 * Cin
template <int kWaves>
__global__ void __launch_bounds__(kWaves * 64)
conv3x3_small_kernel(const ConvParams P) {
  extern __shared__ __attribute__((aligned(16))) float red[];                     // [wave][r2 = 0..31][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, kh = lane >> 5;
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxLevels; i++) l = (i < P.nlev && (int)blockIdx.x >= P.lv[i].tile0) ? i : l;
  const ConvLevel L = P.lv[l];
  const int HW = L.H * L.W, HWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HWo;
  const long p = (long)((int)blockIdx.x - L.tile0) * kTileM + m;
  const bool valid = p < npos;
  const int b = valid ? (int)(p / HWo) : 0;
  const int hw = valid ? (int)(p - (long)b * HWo) : 0;
  const int h = (hw / L.Wo) * L.stride, w = (hw - (hw / L.Wo) * L.Wo) * L.stride;     // input coordinates of the centre tap
  // K = (tap, channel chunk) linearised: chunk g = tap * cpt + t.  This wave owns chunks [g0, g0 + G); with
  // G = 9 * cpt / kWaves <= 1.125 * cpt the range touches at most two taps (boundary gb).
  const int cpt = P.Cin >> 3;                           // chunks per tap
  const int G = (kTaps * cpt) / (kWaves * P.ksplit);
  const int g0 = ((int)blockIdx.z * kWaves + wave) * G;
  const int tap0 = g0 / cpt, gb = (tap0 + 1) * cpt;     // chunks >= gb belong to tap0 + 1
  const int tap1 = tap0 + 1 < kTaps ? tap0 + 1 : tap0;
  const int hh0 = h + tap0 / 3 - 1, ww0 = w + tap0 % 3 - 1, hh1 = h + tap1 / 3 - 1, ww1 = w + tap1 % 3 - 1;
  const bool in0 = valid && hh0 >= 0 && hh0 < L.H && ww0 >= 0 && ww0 < L.W;
  const bool in1 = valid && hh1 >= 0 && hh1 < L.H && ww1 >= 0 && ww1 < L.W;
  // channel 8t + 4kh + i of image b at the tap-shifted position: base + (8t + i) * HW
  const float* xb = L.x + ((size_t)b * P.Cin + 4 * kh) * HW;
  const float* src0 = xb + (in0 ? hh0 * L.W + ww0 : 0) - (size_t)tap0 * cpt * 8 * HW;   // so that index 8*g*HW works
  const float* src1 = xb + (in1 ? hh1 * L.W + ww1 : 0) - (size_t)tap1 * cpt * 8 * HW;
  const int n0 = blockIdx.y * kTileN;
  // weights of chunk g for this lane: float4 #(2g + kh) of output n0 + m  (c4 = (tap*Cin + 8t + 4kh) / 4 = 2g + kh)
  const float* wq = L.w3 + ((size_t)kh * P.Cout + n0 + m) * 4;
  const size_t wstep = (size_t)2 * P.Cout * 4;

  auto load2 = [&](int g, Chunk2& c) {                  // chunks g, g + 1 (same tap: G and cpt are even)
    const bool second = g >= gb;
    const float* src = second ? src1 : src0;
    const bool ok = second ? in1 : in0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int i = 0; i < 4; i++) c.a[u][i] = ok ? src[(size_t)(8 * (g + u) + i) * HW] : 0.f;
      c.b0[u] = *reinterpret_cast<const float4*>(wq + (size_t)(g + u) * wstep);
      c.b1[u] = *reinterpret_cast<const float4*>(wq + (size_t)(g + u) * wstep + 32 * 4);
    }
  };
  floatx16 acc0 = {0}, acc1 = {0};
  auto fma16 = [&](const Chunk2& c) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b0[u].x, c.a[u][0], acc0, 0, 0, 0);      // D[channel][position]
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b1[u].x, c.a[u][0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b0[u].y, c.a[u][1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b1[u].y, c.a[u][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b0[u].z, c.a[u][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b1[u].z, c.a[u][2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b0[u].w, c.a[u][3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.b1[u].w, c.a[u][3], acc1, 0, 0, 0);
    }
  };
  // three chunk pairs in flight: every load is issued two MFMA blocks (32 MFMAs) before its first use
  Chunk2 A, Bq, C;
  const int gend = g0 + G;
  load2(g0, A);
  if (g0 + 2 < gend) load2(g0 + 2, Bq);
  for (int g = g0; g < gend; g += 6) {
    if (g + 4 < gend) load2(g + 4, C);
    fma16(A);
    if (g + 6 < gend) load2(g + 6, A);
    if (g + 2 < gend) fma16(Bq);
    if (g + 8 < gend) load2(g + 8, Bq);
    if (g + 4 < gend) fma16(C);
  }
  float* mine = red + (size_t)wave * kTileM * kTileN;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    mine[r * 64 + lane] = acc0[r];
    mine[(16 + r) * 64 + lane] = acc1[r];
  }
  __syncthreads();
  // fixed-order sum over the K slices; thread e4 owns entries 4*e4 .. 4*e4+3 = (r2, four consecutive lanes)
  if (tid < kTileM * kTileN / 4) {
    float4 s = *reinterpret_cast<const float4*>(red + 4 * tid);
#pragma unroll
    for (int t = 1; t < kWaves; t++) {
      const float4 v = *reinterpret_cast<const float4*>(red + (size_t)t * kTileM * kTileN + 4 * tid);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int r2 = tid >> 4, l0 = (tid & 15) * 4;                   // l0 = first of four consecutive lanes
    const int r = r2 & 15;
    const int ch = n0 + (r2 >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l0 >> 5);
    const float vals[4] = {s.x, s.y, s.z, s.w};
    if (ch < P.Cout) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const long pq = (long)((int)blockIdx.x - L.tile0) * kTileM + ((l0 + q) & 31);
        if (pq < npos) {
          const int bq = (int)(pq / HWo);
          const int hq = (int)(pq - (long)bq * HWo);
          float* dst = P.ksplit > 1 ? L.part + (size_t)blockIdx.z * P.part_stride : L.y;
          dst[((size_t)bq * P.Cout + ch) * HWo + hq] = vals[q];
        }
      }
    }
  }
}

// y[i] = part[0][i] + part[1][i] + ... in this order
__global__ void conv_small_reduce_kernel(const float* __restrict__ part, size_t part_stride, int ksplit, float* __restrict__ y,
                                         long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = part[i];
    for (int z = 1; z < ksplit; z++) v += part[(size_t)z * part_stride + i];
    y[i] = v;
  }
}

// grid-level K split: as many slices as keep the launch near one workgroup per CU, with an even number of chunks per wave
inline int pick_ksplit(int tiles, int nblk, int c_in) {
  int best = 1;
  for (int s = 2; s <= 16; s *= 2) {
    const int per_wave = (kTaps * (c_in >> 3)) / (8 * s);
    if ((kTaps * (c_in >> 3)) % (8 * s) || (per_wave & 1) || per_wave < 4) break;
    if ((long)tiles * nblk * s > 320) break;
    best = s;
  }
  return best;
}

}  // namespace

extern "C" {

// Cin % 128: every wave's K slice (an eighth of 9 * Cin / 8 chunks) is a whole number of chunk PAIRS
int orp_conv3x3_small_ok(int c_in, int c_out) { return (c_in >= 128 && c_in % 128 == 0 && c_out >= 64 && c_out % 64 == 0) ? 1 : 0; }

static int conv3x3_small_launch(const orp_norm_level* levels_host, const float* const* weights_packed_host,
                               const int* strides_host, int nlevels, int batch, int c_in, int c_out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > kMaxLevels || batch <= 0 || !weights_packed_host) return ORP_EINVAL;
  if (!orp_conv3x3_small_ok(c_in, c_out)) return ORP_EINVAL;
  ConvParams P;
  P.nlev = nlevels; P.B = batch; P.Cin = c_in; P.Cout = c_out;
  int tiles = 0;
  size_t out_floats = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_norm_level& lv = levels_host[i];
    const int st = strides_host ? strides_host[i] : 1;
    if (!lv.input || !lv.output || lv.height <= 0 || lv.width <= 0 || lv.input == lv.output || !weights_packed_host[i] ||
        (st != 1 && st != 2))
      return ORP_EINVAL;
    if ((long)batch * lv.height * lv.width >= (1L << 30)) return ORP_ETOOBIG;
    ConvLevel& L = P.lv[i];
    L.x = lv.input; L.y = lv.output; L.H = lv.height; L.W = lv.width; L.tile0 = tiles;
    L.stride = st; L.Ho = (lv.height - 1) / st + 1; L.Wo = (lv.width - 1) / st + 1;      // (H + 2 - 3) / s + 1
    L.w3 = weights_packed_host[i] + (size_t)9 * c_in * c_out;        // second half of orp_dcn_pack_weight's output
    L.part = reinterpret_cast<float*>(workspace) + out_floats;
    out_floats += (size_t)batch * c_out * L.Ho * L.Wo;
    tiles += (int)(((long)batch * L.Ho * L.Wo + kTileM - 1) / kTileM);
  }
  for (int i = nlevels; i < kMaxLevels; i++) { P.lv[i] = P.lv[0]; P.lv[i].tile0 = 0x7fffffff; }
  // few positions (the FPN's extra levels: 256 / 64 of them, K = 9 x 2048 for the first): split K over the grid as well,
  // partial images in the workspace, summed in slice order by a second launch -- parallelism without atomics
  int ks = workspace ? pick_ksplit(tiles, c_out / kTileN, c_in) : 1;
  while (ks > 1 && workspace_bytes < sizeof(float) * out_floats * ks) ks >>= 1;
  P.ksplit = ks; P.part_stride = out_floats;
  // eight waves = two per SIMD (sixteen, with Cin % 256, measured slower: 43 us vs 34 us at the 1024^2 shapes)
  constexpr size_t smem = sizeof(float) * 8 * kTileM * kTileN;                     // 64 KB of partial tiles
  hipLaunchKernelGGL(conv3x3_small_kernel<8>, dim3(tiles, c_out / kTileN, ks), dim3(8 * 64), smem, (hipStream_t)stream, P);
  if (ks > 1) {
    for (int i = 0; i < nlevels; i++) {
      const long n = (long)batch * c_out * P.lv[i].Ho * P.lv[i].Wo;
      long blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
      hipLaunchKernelGGL(conv_small_reduce_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, P.lv[i].part,
                         P.part_stride, ks, P.lv[i].y, n);
    }
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_conv3x3_small_multi_ex(const orp_norm_level* levels_host, const float* const* weights_packed_host, int nlevels,
                               int batch, int c_in, int c_out, void* stream) {
  return conv3x3_small_launch(levels_host, weights_packed_host, nullptr, nlevels, batch, c_in, c_out, nullptr, 0, stream);
}

size_t orp_conv3x3_small_workspace_bytes(const orp_norm_level* levels_host, const int* strides_host, int nlevels, int batch,
                                         int c_out) {
  if (!levels_host || nlevels <= 0) return 0;
  size_t out_floats = 0;
  for (int i = 0; i < nlevels; i++) {
    const int st = strides_host ? strides_host[i] : 1;
    if (st != 1 && st != 2) return 0;
    out_floats += (size_t)batch * c_out * ((levels_host[i].height - 1) / st + 1) * ((levels_host[i].width - 1) / st + 1);
  }
  return sizeof(float) * out_floats * 16;                 // up to 16 K slices
}

int orp_conv3x3_small_multi_strided(const orp_norm_level* levels_host, const float* const* weights_packed_host,
                                    const int* strides_host, int nlevels, int batch, int c_in, int c_out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  if (!strides_host) return ORP_EINVAL;
  return conv3x3_small_launch(levels_host, weights_packed_host, strides_host, nlevels, batch, c_in, c_out, workspace,
                              workspace_bytes, stream);
}

int orp_conv3x3_small_multi(const orp_norm_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                            const float* weight_packed, void* stream) {
  if (nlevels <= 0 || nlevels > kMaxLevels) return ORP_EINVAL;
  const float* w[kMaxLevels];
  for (int i = 0; i < nlevels; i++) w[i] = weight_packed;
  return orp_conv3x3_small_multi_ex(levels_host, w, nlevels, batch, c_in, c_out, stream);
}

}  // extern "C"
