// orp_conv1x1.hip -- the head's 1x1 output convolutions over ALL FPN levels in one launch (gfx950).
//
// reppoints_pts_init_out / reppoints_cls_out / reppoints_pts_refine_out (mmdet/models/anchor_heads/
// orientedreppoints_head.py:105-113, applied at :156-170) map 256 channels to 18 / 15 / 18 per position.  The library runs
// them as five small GEMMs (one per level, 5-12 us each, a few hundred workgroups between them) plus a bias pass; the work
// is one streaming read of the 22 MB input.  Here: one workgroup = 64 positions x 8 channel slices (8 waves: lane =
// position, so every load is a coalesced 256-byte row segment of the NCHW input), the weights of a wave's quarter are
// wave-uniform and live in scalar registers (packed [Cin][32]), fp32 FMA chain in channel order, the eight partial sums are
// added in a fixed order through LDS and the epilogue applies what follows the convolution in the head in the SAME order
// as the separate passes did: + bias, + residual (`pts_out_refine + pts_out_init`), ReLU, and optionally a second output
// `- sub[k]` (`pts_out_init - dcn_base_offset`).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"

namespace {

constexpr int kMaxLevels = 8;
constexpr int KP = 32;           // packed output-channel count (Cout <= 32)
constexpr int NS = 8;            // channel slices = waves per workgroup

struct Lv { const float* x; const float* res; float* y; float* z; int hw; int bx0; };
struct Params {
  Lv lv[kMaxLevels];
  int nlev, B, Cin, Cout, relu;
  const float* wt;     // [Cin][KP], zero padded
  const float* bias; const float* sub;
};

// w [Cout][Cin] -> wt [Cin][KP]
__global__ void pack_1x1_kernel(const float* __restrict__ w, int cout, int cin, float* __restrict__ wt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cin * KP; i += gridDim.x * blockDim.x) {
    const int k = i % KP, c = i / KP;
    wt[i] = k < cout ? w[(size_t)k * cin + c] : 0.f;
  }
}

template <int K>
__global__ void __launch_bounds__(64 * NS)
conv1x1_multi_kernel(const Params P) {
  __shared__ float red[NS][K][64];
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxLevels; i++) l = (i < P.nlev && (int)blockIdx.x >= P.lv[i].bx0) ? i : l;
  const Lv L = P.lv[l];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // channel slice of this wave (scalar)
  const int p0 = ((int)blockIdx.x - L.bx0) * 64, p = p0 + lane;
  const bool live = p < L.hw;
  const int cq = P.Cin / NS;
  const float* xp = L.x + ((size_t)b * P.Cin + (size_t)q * cq) * L.hw + (live ? p : 0);
  const float* wq = P.wt + (size_t)q * cq * KP;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = 0.f;
#pragma unroll 8
  for (int c = 0; c < cq; c++) {
    const float v = live ? xp[(size_t)c * L.hw] : 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = __builtin_fmaf(wq[c * KP + k], v, acc[k]);
  }
#pragma unroll
  for (int k = 0; k < K; k++) red[q][k][lane] = acc[k];
  __syncthreads();
  for (int idx = threadIdx.x; idx < K * 64; idx += 64 * NS) {
    const int k = idx >> 6, pl = idx & 63, pp = p0 + pl;
    if (k >= P.Cout || pp >= L.hw) continue;
    float v = ((red[0][k][pl] + red[1][k][pl]) + (red[2][k][pl] + red[3][k][pl])) +
              ((red[4][k][pl] + red[5][k][pl]) + (red[6][k][pl] + red[7][k][pl]));
    const size_t o = ((size_t)b * P.Cout + k) * L.hw + pp;
    if (P.bias) v += P.bias[k];
    if (L.res) v += L.res[o];
    if (P.relu) v = fmaxf(v, 0.f);
    L.y[o] = v;
    if (L.z) L.z[o] = v - P.sub[k];
  }
}

}  // namespace

extern "C" {

size_t orp_conv1x1_packed_floats(int c_in) { return (size_t)(c_in > 0 ? c_in : 0) * KP; }

int orp_conv1x1_ok(int c_in, int c_out) { return (c_in > 0 && c_in % NS == 0 && c_out > 0 && c_out <= KP) ? 1 : 0; }

int orp_conv1x1_pack_weight(const float* weight, int c_out, int c_in, float* packed, void* stream) {
  if (!weight || !packed || !orp_conv1x1_ok(c_in, c_out)) return ORP_EINVAL;
  hipLaunchKernelGGL(pack_1x1_kernel, dim3((c_in * KP + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, c_out, c_in,
                     packed);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_conv1x1_multi(const orp_bias_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                      const float* weight_packed, const float* bias, const float* sub, int relu, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > kMaxLevels || batch <= 0 || !weight_packed || !orp_conv1x1_ok(c_in, c_out))
    return ORP_EINVAL;
  Params P;
  P.nlev = nlevels; P.B = batch; P.Cin = c_in; P.Cout = c_out; P.relu = relu ? 1 : 0;
  P.wt = weight_packed; P.bias = bias; P.sub = sub;
  int bx = 0;
  for (int i = 0; i < kMaxLevels; i++) {
    const orp_bias_level& s = levels_host[i < nlevels ? i : nlevels - 1];
    if (i < nlevels) {
      if (!s.input || !s.output || s.height <= 0 || s.width <= 0) return ORP_EINVAL;
      if (s.output2 && !sub) return ORP_EINVAL;
    }
    P.lv[i].x = s.input; P.lv[i].res = s.residual; P.lv[i].y = s.output; P.lv[i].z = s.output2;
    P.lv[i].hw = s.height * s.width; P.lv[i].bx0 = i < nlevels ? bx : 0x7fffffff;
    if (i < nlevels) bx += (s.height * s.width + 63) / 64;
  }
  hipStream_t st = (hipStream_t)stream;
  if (c_out <= 16) hipLaunchKernelGGL(conv1x1_multi_kernel<16>, dim3(bx, batch), dim3(64 * NS), 0, st, P);
  else if (c_out <= 18) hipLaunchKernelGGL(conv1x1_multi_kernel<18>, dim3(bx, batch), dim3(64 * NS), 0, st, P);
  else hipLaunchKernelGGL(conv1x1_multi_kernel<32>, dim3(bx, batch), dim3(64 * NS), 0, st, P);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
