// orp_conv_split.hip -- the dense head's 3x3 tower convolutions, ALL FPN levels and one or two layers per launch (gfx950).
//
// The head runs seven 256 -> 256 3x3 convolutions over every FPN level (mmdet/models/anchor_heads/orientedreppoints_head.py:
// 91-113 cls_convs / reg_convs ConvModules, :107 reppoints_pts_init_conv; applied per level by forward_single :148-158).
// The framework issues them level by level on the library: Winograd on the 128^2 / 64^2 maps (168 + 45 us per layer at
// 1024^2, 107-115 TF/s effective) plus this library's small-level launch (33 us) -- 246 us per layer, 1.7 ms per image, the
// largest block of the inference step after the backbone.  fp32 has no fast matrix path on gfx950; the DeformConv forward
// already contracts on the bf16 pipe with every operand split exactly into three bf16 pieces (orp_dcn_split.hip: error
// against the fp64-accumulated oracle below the exact-fp32 MFMA chain's own).  A convolution is that operator without
// offsets: the same kernel, PLAIN instantiation (one row fetch per sample, no bilinear combine), channels-last in, NHWC or
// NCHW out, bias / ReLU in the epilogue; the two towers' layer k are the two grid halves of ONE launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "../../include/orp_hip.h"
#include "orp_dcn_split.hpp"
#include "orp_prof.hpp"
#include "orp_launch.hpp"

namespace {

constexpr int kMaxT = 16;
struct TrLevels {
  const float* in[kMaxT];
  float* out[kMaxT];
  int hw[kMaxT];
  int bx0[kMaxT + 1];
  int nlev;
  int slot[kMaxT];                  // amax != nullptr: the slot tensor i's max |x| goes to
  unsigned* amax;                   // nullptr, or [nslots] float bits, zeroed by the entry (atomicMax per workgroup)
};

// [B][C][HW] -> [B][HW][C] through a 32 x 33 LDS tile, every tensor of the launch back to back along blockIdx.x
__global__ void __launch_bounds__(256)
to_channels_last_kernel(const TrLevels T, int C) {
  __shared__ float tile[32][33];
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxT; i++) l = (i < T.nlev && (int)blockIdx.x >= T.bx0[i]) ? i : l;
  const int HW = T.hw[l];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = ((int)blockIdx.x - T.bx0[l]) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = T.in[l] + (size_t)b * C * HW;
  float* dst = T.out[l] + (size_t)b * C * HW;
  unsigned m = 0u;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    const float v = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
    tile[r][tx] = v;
    m = max(m, __float_as_uint(v) & 0x7fffffffu);
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) dst[(size_t)p * C + c] = tile[tx][r];
  }
  if (T.amax) {                                      // max |x| of the tensors of a slot: what the fp16-pieces convolution scales by
    __shared__ unsigned red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      // thousands of workgroups, one address: the atomic only where it would change the value (a contended atomicMax per
      // workgroup serialised in the L2: 37 us instead of 9 for the FPN's three levels, 190 us at 2 x 1024^2)
      const unsigned mx = max(max(red[0], red[1]), max(red[2], red[3]));
      unsigned* dst_ = T.amax + T.slot[l];
      if (mx > __atomic_load_n(dst_, __ATOMIC_RELAXED)) atomicMax(dst_, mx);
    }
  }
}

inline int out_dim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

}  // namespace

extern "C" {

int orp_conv_split_ok(int c_in, int c_out, int kh, int kw) { return orp_split::shape_ok(c_in, c_out, kh, kw) ? 1 : 0; }

struct GnFuse {                      // orp_conv_split_multi_gn: GroupNorm around the launch (see include/orp_hip.h)
  const float* coef_in; int relu_in; float* partials; size_t partial_floats; int groups; int amax_count;
};

static int fill_args(orp_split::Args& A, const orp_conv_level* levels_host, int nlevels, int batch, int c_in, int c_out, int nconv,
                     int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w) {
  A.nlev = nlevels; A.B = batch; A.Cin = c_in; A.Cout = c_out; A.nconv = nconv;
  A.kh = kh; A.kw = kw; A.sh = stride_h; A.sw = stride_w; A.ph = pad_h; A.pw = pad_w; A.dh = dil_h; A.dw = dil_w;
  for (int i = 0; i < nlevels; i++) {
    const orp_conv_level& lv = levels_host[i];
    if (lv.height <= 0 || lv.width <= 0) return ORP_EINVAL;
    if ((long)batch * lv.height * lv.width >= (1L << 31)) return ORP_ETOOBIG;
    orp_split::Level& S = A.lv[i];
    S.off = nullptr; S.mask = nullptr; S.planes = nullptr; S.bias = nullptr; S.wscale = nullptr;
    S.H = lv.height; S.W = lv.width;
    S.Ho = out_dim(lv.height, pad_h, dil_h, kh, stride_h);
    S.Wo = out_dim(lv.width, pad_w, dil_w, kw, stride_w);
    if (S.Ho <= 0 || S.Wo <= 0) return ORP_EINVAL;
  }
  return ORP_OK;
}

// The tile table of the launch that filled a `partials` buffer, kept on the host (keyed by the buffer; the last kGnPlans launches):
// orp_conv_split_gn_finish merges the tiles by THAT table instead of rebuilding one from its own arguments -- a finish call whose
// layer / level / image counts differ from the launch's would otherwise pick another tile height and merge the wrong slots unnoticed
// (round-5 advisor).  Host calls of a capture happen in program order, so a captured graph is validated when it is captured.
struct GnPlanRecord { const float* partials; orp_split::Plan plan; int nconv, nlev, batch, groups; };
constexpr int kGnPlans = 32;
static GnPlanRecord g_gn_plans[kGnPlans];
static int g_gn_next = 0;
static std::mutex g_gn_mutex;
static void remember_gn_plan(const float* partials, const orp_split::Plan& pl, int nconv, int nlev, int batch, int groups) {
  std::lock_guard<std::mutex> lock(g_gn_mutex);
  int slot = -1;
  for (int i = 0; i < kGnPlans; i++) if (g_gn_plans[i].partials == partials) slot = i;
  if (slot < 0) { slot = g_gn_next; g_gn_next = (g_gn_next + 1) % kGnPlans; }
  g_gn_plans[slot] = GnPlanRecord{partials, pl, nconv, nlev, batch, groups};
}
static bool recall_gn_plan(const float* partials, GnPlanRecord* out) {
  std::lock_guard<std::mutex> lock(g_gn_mutex);
  for (int i = 0; i < kGnPlans; i++)
    if (g_gn_plans[i].partials == partials && partials) { *out = g_gn_plans[i]; return true; }
  return false;
}

static int conv_split_impl(const orp_conv_level* levels_host, const float* const* weights_host, const float* const* biases_host,
                           int nlevels, int batch, int c_in, int c_out, const float* weight_a_packed, const float* weight_b_packed,
                           const float* bias_a, const float* bias_b, int relu, int kh, int kw, int stride_h, int stride_w,
                           int pad_h, int pad_w, int dil_h, int dil_w, int out_layout, int nprod, void* workspace,
                           size_t workspace_bytes, const uint32_t* amax_in, int amax_stride, void* stream, const GnFuse* gn = nullptr) {
  if (!levels_host || nlevels <= 0 || nlevels > orp_split::kMaxLevels || batch <= 0 || !weight_a_packed) return ORP_EINVAL;
  if (!orp_split::shape_ok(c_in, c_out, kh, kw) || (nprod != 3 && nprod != 6 && nprod != 9) || (out_layout != 0 && out_layout != 1))
    return ORP_EINVAL;
  if (nprod == 3 && !amax_in && (!workspace || workspace_bytes < 256)) return ORP_EWORKSPACE;   // max |x| of the inputs lives there
  if (amax_in && amax_stride != 0 && amax_stride != 1 && !(gn && amax_stride >= gn->amax_count && gn->amax_count >= 1)) return ORP_EINVAL;
  if (stride_h <= 0 || stride_w <= 0 || dil_h <= 0 || dil_w <= 0 || pad_h < 0 || pad_w < 0) return ORP_EINVAL;
  const int nconv = weight_b_packed ? 2 : 1;
  orp_split::Args A;
  A.nlev = nlevels; A.B = batch; A.Cin = c_in; A.Cout = c_out;
  A.kh = kh; A.kw = kw; A.sh = stride_h; A.sw = stride_w; A.ph = pad_h; A.pw = pad_w; A.dh = dil_h; A.dw = dil_w;
  const int taps = kh * kw;                                             // orp_dcn_pack_weight: the planes follow the fp32 packings
  A.planes[0] = orp_split::planes_of(weight_a_packed, c_out, c_in, taps, nprod);
  A.planes[1] = nconv == 2 ? orp_split::planes_of(weight_b_packed, c_out, c_in, taps, nprod) : A.planes[0];
  A.wscale[0] = orp_split::wscale_of(weight_a_packed, c_out, c_in, taps);
  A.wscale[1] = nconv == 2 ? orp_split::wscale_of(weight_b_packed, c_out, c_in, taps) : A.wscale[0];
  A.scratch = nprod == 3 ? reinterpret_cast<unsigned*>(workspace) : nullptr;
  A.amax_in = nprod == 3 ? amax_in : nullptr; A.amax_stride = amax_stride;
  A.bias[0] = bias_a; A.bias[1] = nconv == 2 ? bias_b : bias_a;
  A.relu = relu ? 1 : 0; A.nconv = nconv; A.out_nchw = out_layout == 0 ? 1 : 0; A.nprod = nprod;
  if (gn) {
    A.per_image = 1; A.coef_in = gn->coef_in; A.relu_in = gn->relu_in; A.gn_part = gn->partials; A.groups = gn->groups;
    A.amax_count = gn->amax_count;
  }
  for (int i = 0; i < nlevels; i++) {
    const orp_conv_level& lv = levels_host[i];
    if (!lv.input_a || !lv.output_a || lv.height <= 0 || lv.width <= 0) return ORP_EINVAL;
    if (nconv == 2 && (!lv.input_b || !lv.output_b)) return ORP_EINVAL;
    if ((long)batch * lv.height * lv.width >= (1L << 31)) return ORP_ETOOBIG;
    orp_split::Level& S = A.lv[i];
    S.x[0] = lv.input_a; S.x[1] = nconv == 2 ? lv.input_b : lv.input_a;
    S.off = nullptr; S.mask = nullptr;
    S.planes = nullptr; S.bias = nullptr; S.wscale = nullptr;
    if (weights_host) {                                                   // a layer of its own for this level
      if (!weights_host[i]) return ORP_EINVAL;
      S.planes = orp_split::planes_of(weights_host[i], c_out, c_in, taps, nprod);
      S.wscale = orp_split::wscale_of(weights_host[i], c_out, c_in, taps);
      S.bias = biases_host ? biases_host[i] : nullptr;
    }
    S.out[0] = lv.output_a; S.out[1] = nconv == 2 ? lv.output_b : lv.output_a;
    S.H = lv.height; S.W = lv.width;
    S.Ho = out_dim(lv.height, pad_h, dil_h, kh, stride_h);
    S.Wo = out_dim(lv.width, pad_w, dil_w, kw, stride_w);
    if (S.Ho <= 0 || S.Wo <= 0) return ORP_EINVAL;
  }
  orp_split::Plan pl_gn;
  if (gn && gn->partials) {                                // room for [layer][tile][group] (mean, M2, max |y|, count)
    pl_gn = orp_split::plan(A);
    if (gn->partial_floats < (size_t)4 * nconv * pl_gn.tiles * gn->groups) return ORP_EWORKSPACE;
  }
  OrpProfScope prof(ORP_PROF_CONV_SPLIT, (hipStream_t)stream);
  const hipError_t e = orp_split::launch(A, (hipStream_t)stream);
  if (e == hipSuccess && gn && gn->partials) remember_gn_plan(gn->partials, pl_gn, nconv, nlevels, batch, gn->groups);
  return e == hipSuccess ? ORP_OK : (int)e;
}

// ---- GroupNorm fused around the tower convolutions: the merge of the tiles' statistics --------------------------------------------
// one wave per (tensor, image, group): Chan et al.'s merge of the image's tile partials in two fixed-order passes (the weighted
// mean, then M2 around it) -> the (a, b) of  y = x * a[c] + b[c]  for the group's channels, and an upper bound of max |y|
struct GnFinish {
  const float4* part;          // [nconv][tiles][G]
  float2* coef;                // [tensor][B][C]
  unsigned* bound;             // [nconv][nlev * B * G] float bits, or nullptr
  const float* gamma[2 * orp_split::kMaxLevels];
  const float* beta[2 * orp_split::kMaxLevels];
  int tile0[orp_split::kMaxLevels], tpi[orp_split::kMaxLevels];
  int nlev, tiles, B, C, G;
  float eps;
};
__global__ void __launch_bounds__(64)
conv_gn_finish_kernel(const GnFinish F) {
  const int grp = blockIdx.x, b = blockIdx.y, tensor = blockIdx.z;
  const int conv = tensor / F.nlev, lvl = tensor - conv * F.nlev;
  const int cg = F.C / F.G, lane = threadIdx.x;
  const float4* part = F.part + ((size_t)conv * F.tiles + F.tile0[lvl] + (size_t)b * F.tpi[lvl]) * F.G + grp;
  auto wave_sum = [](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  float sn = 0.f, sm = 0.f;
  for (int t = lane; t < F.tpi[lvl]; t += 64) { const float4 p = part[(size_t)t * F.G]; sn += p.w; sm += p.w * p.x; }
  const float total = wave_sum(sn);
  const float mean = wave_sum(sm) / total;
  float m2 = 0.f, mx = 0.f;
  for (int t = lane; t < F.tpi[lvl]; t += 64) {
    const float4 p = part[(size_t)t * F.G];
    const float d = p.x - mean;
    m2 += p.y + p.w * d * d;
    mx = fmaxf(mx, p.z);
  }
  const float var = wave_sum(m2) / total;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float rstd = rsqrtf(var + F.eps);
  const float* gamma = F.gamma[tensor];
  const float* beta = F.beta[tensor];
  float gm = 0.f, bm = 0.f;
  if (lane < cg) {
    const int c = grp * cg + lane;
    const float a = rstd * gamma[c];
    F.coef[((size_t)tensor * F.B + b) * F.C + c] = make_float2(a, beta[c] - mean * a);
    gm = fabsf(gamma[c]); bm = fabsf(beta[c]);
  }
  if (F.bound) {
    // |y| = |(x - mean) rstd gamma_c + beta_c| <= (max |x| + |mean|) rstd max |gamma| + max |beta| over the group's channels
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, o, 64)); bm = fmaxf(bm, __shfl_xor(bm, o, 64)); }
    if (lane == 0)
      F.bound[(size_t)conv * F.nlev * F.B * F.G + ((size_t)lvl * F.B + b) * F.G + grp] =
          __float_as_uint(((mx + fabsf(mean)) * rstd * gm + bm) * 1.0001f);
  }
}

int orp_conv_split_multi(const orp_conv_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                         const float* weight_a_packed, const float* weight_b_packed, const float* bias_a, const float* bias_b,
                         int relu, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                         int out_layout, int nprod, void* workspace, size_t workspace_bytes, const uint32_t* amax_in,
                         int amax_stride, void* stream) {
  return conv_split_impl(levels_host, nullptr, nullptr, nlevels, batch, c_in, c_out, weight_a_packed, weight_b_packed, bias_a,
                         bias_b, relu, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, out_layout, nprod, workspace,
                         workspace_bytes, amax_in, amax_stride, stream);
}

int orp_conv_split_multi_ex(const orp_conv_level* levels_host, const float* const* weights_packed_host,
                            const float* const* biases_host, int nlevels, int batch, int c_in, int c_out, int relu, int kh, int kw,
                            int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int out_layout, int nprod,
                            void* workspace, size_t workspace_bytes, const uint32_t* amax_in, void* stream) {
  if (!weights_packed_host || nlevels <= 0) return ORP_EINVAL;
  return conv_split_impl(levels_host, weights_packed_host, biases_host, nlevels, batch, c_in, c_out, weights_packed_host[0],
                         nullptr, nullptr, nullptr, relu, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, out_layout,
                         nprod, workspace, workspace_bytes, amax_in, 0, stream);
}

static int to_cl_impl(const orp_norm_level* levels_host, int nlevels, int batch, int channels, const int* slots_host,
                      uint32_t* amax_out, int nslots, int reset, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > kMaxT || batch <= 0 || batch > 65535 || channels <= 0) return ORP_EINVAL;
  if (amax_out && (!slots_host || nslots <= 0)) return ORP_EINVAL;
  TrLevels T;
  int bx = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_norm_level& lv = levels_host[i];
    if (!lv.input || !lv.output || lv.input == lv.output || lv.height <= 0 || lv.width <= 0) return ORP_EINVAL;
    if (amax_out && (slots_host[i] < 0 || slots_host[i] >= nslots)) return ORP_EINVAL;
    T.in[i] = lv.input; T.out[i] = lv.output; T.hw[i] = lv.height * lv.width; T.bx0[i] = bx;
    T.slot[i] = amax_out ? slots_host[i] : 0;
    bx += (T.hw[i] + 31) / 32;
  }
  T.nlev = nlevels; T.amax = amax_out;
  for (int i = nlevels; i <= kMaxT; i++) T.bx0[i] = bx;
  for (int i = nlevels; i < kMaxT; i++) { T.in[i] = T.in[0]; T.out[i] = T.out[0]; T.hw[i] = 0; T.slot[i] = 0; }
  if (amax_out && reset) {
    const hipError_t me = orp::fill_async(amax_out, 0, sizeof(unsigned) * (size_t)(nslots), (hipStream_t)stream);
    if (me != hipSuccess) return (int)me;
  }
  hipLaunchKernelGGL(to_channels_last_kernel, dim3(bx, (channels + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream, T, channels);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

size_t orp_conv_split_gn_partial_floats(const orp_conv_level* levels_host, int nlevels, int batch, int groups, int nlayers) {
  if (!levels_host || nlevels <= 0 || nlevels > orp_split::kMaxLevels || batch <= 0 || groups <= 0 || nlayers < 1 || nlayers > 2) return 0;
  // (an upper bound that does not depend on the tile height the launch picks: tiles of 32 positions)
  size_t tiles = 0;
  for (int i = 0; i < nlevels; i++) tiles += (size_t)batch * (((size_t)levels_host[i].height * levels_host[i].width + 31) / 32);
  return (size_t)4 * nlayers * tiles * groups;
}

int orp_conv_split_multi_gn(const orp_conv_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                            const float* weight_a_packed, const float* weight_b_packed, int kh, int kw, int pad_h, int pad_w,
                            int dil_h, int dil_w, int nprod, const float* coef_in, int relu_in, float* partials,
                            size_t partial_floats, int groups, void* workspace, size_t workspace_bytes, const uint32_t* amax_in,
                            int amax_stride, int amax_count, void* stream) {
  if (!partials || groups <= 0 || c_out % groups != 0 || 32 % (c_out / groups) != 0) return ORP_EINVAL;
  if (2 * pad_h != dil_h * (kh - 1) || 2 * pad_w != dil_w * (kw - 1)) return ORP_EINVAL;       // 'same' convolutions: H x W in and out
  if (coef_in && nprod == 3 && !amax_in) nprod = 6;        // (a range pre-pass would see the un-normalised inputs)
  GnFuse gn{coef_in, relu_in ? 1 : 0, partials, partial_floats, groups, amax_in ? amax_count : 0};
  return conv_split_impl(levels_host, nullptr, nullptr, nlevels, batch, c_in, c_out, weight_a_packed, weight_b_packed, nullptr,
                         nullptr, 0, kh, kw, 1, 1, pad_h, pad_w, dil_h, dil_w, 1, nprod, workspace, workspace_bytes, amax_in,
                         amax_stride, stream, &gn);
}

int orp_conv_split_gn_finish(const orp_conv_level* levels_host, int nlevels, int batch, int channels, int groups, int nlayers,
                             float eps, const float* const* gammas_host, const float* const* betas_host, const float* partials,
                             float* coef_out, uint32_t* bound_out, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > orp_split::kMaxLevels || batch <= 0 || batch > 65535 || nlayers < 1 || nlayers > 2 ||
      !gammas_host || !betas_host || !partials || !coef_out || groups <= 0 || channels % groups != 0 || channels / groups > 64)
    return ORP_EINVAL;
  orp_split::Plan pl;                                       // the tile table of the launch that wrote the partials
  GnPlanRecord rec;
  if (recall_gn_plan(partials, &rec)) {
    if (rec.nconv != nlayers || rec.nlev != nlevels || rec.batch != batch || rec.groups != groups) return ORP_EINVAL;
    pl = rec.plan;
  } else {                                                  // (a buffer this process never launched into: the table the arguments imply)
    orp_split::Args A;
    const int rc = fill_args(A, levels_host, nlevels, batch, channels, channels, nlayers, 3, 3, 1, 1, 1, 1, 1, 1);
    if (rc != ORP_OK) return rc;
    A.per_image = 1;
    pl = orp_split::plan(A);
  }
  GnFinish F;
  F.part = reinterpret_cast<const float4*>(partials); F.coef = reinterpret_cast<float2*>(coef_out); F.bound = bound_out;
  F.nlev = nlevels; F.tiles = pl.tiles; F.B = batch; F.C = channels; F.G = groups; F.eps = eps;
  for (int i = 0; i < orp_split::kMaxLevels; i++) { F.tile0[i] = pl.tile0[i]; F.tpi[i] = pl.tpi[i]; }
  for (int i = 0; i < 2 * orp_split::kMaxLevels; i++) { F.gamma[i] = nullptr; F.beta[i] = nullptr; }
  for (int i = 0; i < nlayers * nlevels; i++) {
    if (!gammas_host[i] || !betas_host[i]) return ORP_EINVAL;
    F.gamma[i] = gammas_host[i]; F.beta[i] = betas_host[i];
  }
  hipLaunchKernelGGL(conv_gn_finish_kernel, dim3(groups, batch, nlayers * nlevels), dim3(64), 0, (hipStream_t)stream, F);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_debug_amax_log(uint32_t* log, int capacity_launches) { return orp_split::set_amax_log(log, capacity_launches); }

int orp_nchw_to_nhwc_multi(const orp_norm_level* levels_host, int nlevels, int batch, int channels, void* stream) {
  return to_cl_impl(levels_host, nlevels, batch, channels, nullptr, nullptr, 0, 0, stream);
}

int orp_nchw_to_nhwc_multi_amax(const orp_norm_level* levels_host, int nlevels, int batch, int channels, const int* slots_host,
                                uint32_t* amax_out, int nslots, int reset, void* stream) {
  if (!amax_out) return ORP_EINVAL;
  return to_cl_impl(levels_host, nlevels, batch, channels, slots_host, amax_out, nslots, reset, stream);
}

}  // extern "C"
