// orp_dcn_bwd.hip -- deformable convolution backward (DCNv1 / DCNv2) for gfx950, column formulation.
//
// Replaces deformable_im2col / deformable_col2im / deformable_col2im_coord and their modulated twins
//   (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:190-277, 279-371, 373-465, 570-867) as used by
//   deform_conv_backward_input_cuda / deform_conv_backward_parameters_cuda (deform_conv_cuda.cpp:262-488).
// Round-1 shape: the two GEMMs (grad_col = W^T . grad_out, grad_W = grad_out . col^T) are plain library GEMMs issued
// by the host wrapper; this file provides the sampling kernels around them.  Differences from the reference:
//   * ONE kernel produces grad_input, grad_offset (and grad_mask) from grad_col: the bilinear geometry of a
//     (position, tap) is computed once and reused over all channels of the deformable group (the reference launches
//     col2im with a 5x5 neighbourhood scan per element plus a separate coord kernel that recomputes everything);
//   * lanes run along output positions, so grad_col / offset reads are coalesced.
// (Next step, DESIGN.md: fold the two GEMMs into MFMA kernels like the forward so that no column buffer hits HBM.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_prof.hpp"

namespace {

struct Geo { int B, C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg; };

// T = float | double: the reference instantiates these kernels for both (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// deform_conv_cuda_kernel.cu:259,353,451); for T = float every expression below is the fp32 expression of rounds 1-5
__device__ __forceinline__ float floor_t(float v) { return floorf(v); }
__device__ __forceinline__ double floor_t(double v) { return floor(v); }

// col[(c*taps + t)][b][ho][wo] = (mask *) bilinear(x[b,c], ...)      one thread per (c, b, p), loops taps
template <typename T>
__global__ void dcn_im2col_kernel(const T* __restrict__ x, const T* __restrict__ off,
                                  const T* __restrict__ mask, Geo g, T* __restrict__ col) {
  const int taps = g.kh * g.kw, P = g.Ho * g.Wo, cpdg = g.C / g.dg;
  const long total = (long)g.C * g.B * P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int p = (int)(idx % P);
    const int b = (int)((idx / P) % g.B);
    const int c = (int)(idx / ((long)P * g.B));
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int dgi = c / cpdg;
    const T* xp = x + ((size_t)b * g.C + c) * g.H * g.W;
    const T* op = off + ((size_t)b * g.dg + dgi) * 2 * taps * P + p;
    const T* mp = mask ? mask + ((size_t)b * g.dg + dgi) * taps * P + p : nullptr;
    for (int t = 0; t < taps; t++) {
      const int ki = t / g.kw, kj = t - ki * g.kw;
      const T h_im = (T)(ho * g.sh - g.ph + ki * g.dh) + op[(size_t)(2 * t) * P];
      const T w_im = (T)(wo * g.sw - g.pw + kj * g.dw) + op[(size_t)(2 * t + 1) * P];
      T val = 0;
      if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)g.H && w_im < (T)g.W) {
        const int hl = (int)floor_t(h_im), wl = (int)floor_t(w_im), hh = hl + 1, wh = wl + 1;
        const T lh = h_im - hl, lw = w_im - wl, uh = (T)1 - lh, uw = (T)1 - lw;
        const T v1 = (hl >= 0 && wl >= 0) ? xp[hl * g.W + wl] : (T)0;
        const T v2 = (hl >= 0 && wh <= g.W - 1) ? xp[hl * g.W + wh] : (T)0;
        const T v3 = (hh <= g.H - 1 && wl >= 0) ? xp[hh * g.W + wl] : (T)0;
        const T v4 = (hh <= g.H - 1 && wh <= g.W - 1) ? xp[hh * g.W + wh] : (T)0;
        val = uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
      }
      if (mp) val *= mp[(size_t)t * P];
      col[(((size_t)c * taps + t) * g.B + b) * P + p] = val;
    }
  }
}

// one thread per (b, deformable group, tap, position): loops the group's channels
template <typename T>
__global__ void dcn_col2im_kernel(const T* __restrict__ gcol, const T* __restrict__ x,
                                  const T* __restrict__ off, const T* __restrict__ mask, Geo g,
                                  T* __restrict__ grad_x, T* __restrict__ grad_off,
                                  T* __restrict__ grad_mask) {
  const int taps = g.kh * g.kw, P = g.Ho * g.Wo, cpdg = g.C / g.dg;
  const long total = (long)g.B * g.dg * taps * P;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int p = (int)(idx % P);
    const int t = (int)((idx / P) % taps);
    const int dgi = (int)((idx / ((long)P * taps)) % g.dg);
    const int b = (int)(idx / ((long)P * taps * g.dg));
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int ki = t / g.kw, kj = t - ki * g.kw;
    const size_t obase = (((size_t)b * g.dg + dgi) * 2 * taps) * P + p;
    const T h_im = (T)(ho * g.sh - g.ph + ki * g.dh) + off[obase + (size_t)(2 * t) * P];
    const T w_im = (T)(wo * g.sw - g.pw + kj * g.dw) + off[obase + (size_t)(2 * t + 1) * P];
    const bool inside = h_im > (T)-1 && w_im > (T)-1 && h_im < (T)g.H && w_im < (T)g.W;
    const T m = mask ? mask[(((size_t)b * g.dg + dgi) * taps + t) * P + p] : (T)1;
    T acc_h = 0, acc_w = 0, acc_m = 0;
    if (inside) {
      const int hl = (int)floor_t(h_im), wl = (int)floor_t(w_im), hh = hl + 1, wh = wl + 1;
      const T lh = h_im - hl, lw = w_im - wl, uh = (T)1 - lh, uw = (T)1 - lw;
      const bool t_ok = hl >= 0, b_ok = hh <= g.H - 1, l_ok = wl >= 0, r_ok = wh <= g.W - 1;
      for (int cc = 0; cc < cpdg; cc++) {
        const int c = dgi * cpdg + cc;
        const T top = gcol[(((size_t)c * taps + t) * g.B + b) * P + p];
        const T* xp = x + ((size_t)b * g.C + c) * g.H * g.W;
        T* gp = grad_x + ((size_t)b * g.C + c) * g.H * g.W;
        const T v1 = (t_ok && l_ok) ? xp[hl * g.W + wl] : (T)0;
        const T v2 = (t_ok && r_ok) ? xp[hl * g.W + wh] : (T)0;
        const T v3 = (b_ok && l_ok) ? xp[hh * g.W + wl] : (T)0;
        const T v4 = (b_ok && r_ok) ? xp[hh * g.W + wh] : (T)0;
        const T tm = top * m;
        if (t_ok && l_ok) atomicAdd(gp + hl * g.W + wl, uh * uw * tm);
        if (t_ok && r_ok) atomicAdd(gp + hl * g.W + wh, uh * lw * tm);
        if (b_ok && l_ok) atomicAdd(gp + hh * g.W + wl, lh * uw * tm);
        if (b_ok && r_ok) atomicAdd(gp + hh * g.W + wh, lh * lw * tm);
        // d sample / d h, d sample / d w   (get_coordinate_weight, deform_conv_cuda_kernel.cu:145-188)
        acc_h += tm * (-uw * v1 - lw * v2 + uw * v3 + lw * v4);
        acc_w += tm * (-uh * v1 + uh * v2 - lh * v3 + lh * v4);
        if (grad_mask) acc_m += top * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4);
      }
    }
    grad_off[obase + (size_t)(2 * t) * P] = acc_h;
    grad_off[obase + (size_t)(2 * t + 1) * P] = acc_w;
    if (grad_mask) grad_mask[(((size_t)b * g.dg + dgi) * taps + t) * P + p] = acc_m;
  }
}

// ---- channel-parallel col2im (deformable_groups = 1), everything coalesced -------------------------------------------
// gcolT [B*P, taps, C] = grad_out(NHWC) . W[Cout, taps*C]  (a plain GEMM whose rows are positions), x NHWC.
// ONE WAVE per (image, position, tap): the bilinear geometry is wave-uniform; lane l owns channels l, l+64, ... so the
// grad_col row, the four neighbour rows of x and the four atomicAdd rows into grad_x (NHWC) are each contiguous
// 256-byte segments per instruction, and grad_offset is a wave reduction.  The thread-per-(position, tap) kernel above
// walks its 256 channels serially with 4 scattered loads + 4 scattered atomics per step (11.6 ms per training step at
// 2 x 1024^2, rocprofv3 round 1); this one is bandwidth-shaped.
__global__ void __launch_bounds__(256)
dcn_col2im_nhwc_kernel(const float* __restrict__ gcolT, const float* __restrict__ x, const float* __restrict__ off,
                       Geo g, float* __restrict__ grad_x, float* __restrict__ grad_off) {
  const int taps = g.kh * g.kw, P = g.Ho * g.Wo;
  const long nw = (long)g.B * P * taps;
  const int lane = threadIdx.x & 63;
  for (long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < nw; wv += (long)gridDim.x * 4) {
    const int t = (int)(wv % taps);
    const long bp = wv / taps;
    const int p = (int)(bp % P), b = (int)(bp / P);
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int ki = t / g.kw, kj = t - ki * g.kw;
    const size_t obase = ((size_t)b * 2 * taps) * P + p;
    const float h_im = (float)(ho * g.sh - g.ph + ki * g.dh) + off[obase + (size_t)(2 * t) * P];
    const float w_im = (float)(wo * g.sw - g.pw + kj * g.dw) + off[obase + (size_t)(2 * t + 1) * P];
    float acc_h = 0.f, acc_w = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)g.H && w_im < (float)g.W) {
      const int hl = (int)floorf(h_im), wl = (int)floorf(w_im), hh = hl + 1, wh = wl + 1;
      const float lh = h_im - hl, lw = w_im - wl, uh = 1.f - lh, uw = 1.f - lw;
      const bool t_ok = hl >= 0, b_ok = hh <= g.H - 1, l_ok = wl >= 0, r_ok = wh <= g.W - 1;
      const size_t img = (size_t)b * g.H * g.W;
      const size_t i1 = (img + (size_t)(t_ok ? hl : 0) * g.W + (l_ok ? wl : 0)) * g.C;
      const size_t i2 = (img + (size_t)(t_ok ? hl : 0) * g.W + (r_ok ? wh : 0)) * g.C;
      const size_t i3 = (img + (size_t)(b_ok ? hh : 0) * g.W + (l_ok ? wl : 0)) * g.C;
      const size_t i4 = (img + (size_t)(b_ok ? hh : 0) * g.W + (r_ok ? wh : 0)) * g.C;
      const bool k1 = t_ok && l_ok, k2 = t_ok && r_ok, k3 = b_ok && l_ok, k4 = b_ok && r_ok;
      const float* gc = gcolT + (size_t)wv * g.C;
      for (int c = lane; c < g.C; c += 64) {
        const float top = gc[c];
        const float v1 = k1 ? x[i1 + c] : 0.f, v2 = k2 ? x[i2 + c] : 0.f;
        const float v3 = k3 ? x[i3 + c] : 0.f, v4 = k4 ? x[i4 + c] : 0.f;
        if (k1) atomicAdd(grad_x + i1 + c, uh * uw * top);
        if (k2) atomicAdd(grad_x + i2 + c, uh * lw * top);
        if (k3) atomicAdd(grad_x + i3 + c, lh * uw * top);
        if (k4) atomicAdd(grad_x + i4 + c, lh * lw * top);
        acc_h += top * (-uw * v1 - lw * v2 + uw * v3 + lw * v4);
        acc_w += top * (-uh * v1 + uh * v2 - lh * v3 + lh * v4);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { acc_h += __shfl_xor(acc_h, o, 64); acc_w += __shfl_xor(acc_w, o, 64); }
    }
    if (lane == 0) {
      grad_off[obase + (size_t)(2 * t) * P] = acc_h;
      grad_off[obase + (size_t)(2 * t + 1) * P] = acc_w;
    }
  }
}

inline int out_dim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }
inline int fill(Geo& g, int B, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || dg <= 0 || C % dg) return ORP_EINVAL;
  g.B = B; g.C = C; g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw; g.dh = dh; g.dw = dw; g.dg = dg;
  g.Ho = out_dim(H, ph, dh, kh, sh); g.Wo = out_dim(W, pw, dw, kw, sw);
  return (g.Ho > 0 && g.Wo > 0) ? ORP_OK : ORP_EINVAL;
}
inline int blocks_for(long total) { long b = (total + 255) / 256; if (b > 256L * 64) b = 256L * 64; return (int)(b < 1 ? 1 : b); }

template <typename T>
int im2col_any(const T* input, const T* offset, const T* mask, int batch, int c_in, int height, int width, int kh, int kw,
               int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int deformable_groups, T* columns, void* stream) {
  Geo g;
  if (!input || !offset || !columns) return ORP_EINVAL;
  int rc = fill(g, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, deformable_groups);
  if (rc != ORP_OK) return rc;
  OrpProfScope prof(ORP_PROF_DCN_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL(dcn_im2col_kernel<T>, dim3(blocks_for((long)c_in * batch * g.Ho * g.Wo)), dim3(256), 0,
                     (hipStream_t)stream, input, offset, mask, g, columns);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
template <typename T>
int col2im_any(const T* grad_columns, const T* input, const T* offset, const T* mask, int batch, int c_in, int height, int width,
               int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int deformable_groups,
               T* grad_input, T* grad_offset, T* grad_mask, void* stream) {
  Geo g;
  if (!grad_columns || !input || !offset || !grad_input || !grad_offset) return ORP_EINVAL;
  if ((mask == nullptr) != (grad_mask == nullptr)) return ORP_EINVAL;
  int rc = fill(g, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, deformable_groups);
  if (rc != ORP_OK) return rc;
  OrpProfScope prof(ORP_PROF_DCN_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL(dcn_col2im_kernel<T>, dim3(blocks_for((long)batch * deformable_groups * kh * kw * g.Ho * g.Wo)),
                     dim3(256), 0, (hipStream_t)stream, grad_columns, input, offset, mask, g, grad_input, grad_offset,
                     grad_mask);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
}  // namespace

extern "C" {
int orp_dcn_im2col(const float* input, const float* offset, const float* mask, int batch, int c_in, int height,
                   int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                   int deformable_groups, float* columns, void* stream) {
  return im2col_any<float>(input, offset, mask, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                           deformable_groups, columns, stream);
}
int orp_dcn_im2col_f64(const double* input, const double* offset, const double* mask, int batch, int c_in, int height,
                       int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                       int deformable_groups, double* columns, void* stream) {
  return im2col_any<double>(input, offset, mask, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                            deformable_groups, columns, stream);
}

// grad_input must be ZEROED by the caller (it is accumulated with atomics); grad_offset / grad_mask are overwritten.
int orp_dcn_col2im(const float* grad_columns, const float* input, const float* offset, const float* mask, int batch,
                   int c_in, int height, int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                   int dil_h, int dil_w, int deformable_groups, float* grad_input, float* grad_offset, float* grad_mask,
                   void* stream) {
  return col2im_any<float>(grad_columns, input, offset, mask, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w,
                           dil_h, dil_w, deformable_groups, grad_input, grad_offset, grad_mask, stream);
}
int orp_dcn_col2im_f64(const double* grad_columns, const double* input, const double* offset, const double* mask, int batch,
                       int c_in, int height, int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                       int dil_h, int dil_w, int deformable_groups, double* grad_input, double* grad_offset, double* grad_mask,
                       void* stream) {
  return col2im_any<double>(grad_columns, input, offset, mask, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w,
                            dil_h, dil_w, deformable_groups, grad_input, grad_offset, grad_mask, stream);
}

// Channel-parallel variant (deformable_groups = 1): grad_columns_t [B*Ho*Wo, kh*kw, C] (position-major), input and
// grad_input NHWC [B,H,W,C] (grad_input ZEROED by the caller), offset / grad_offset NCHW [B, 2*kh*kw, Ho, Wo].
int orp_dcn_col2im_nhwc(const float* grad_columns_t, const float* input_nhwc, const float* offset, int batch, int c_in,
                        int height, int width, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                        int dil_h, int dil_w, float* grad_input_nhwc, float* grad_offset, void* stream) {
  Geo g;
  if (!grad_columns_t || !input_nhwc || !offset || !grad_input_nhwc || !grad_offset) return ORP_EINVAL;
  int rc = fill(g, batch, c_in, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, 1);
  if (rc != ORP_OK) return rc;
  const long nw = (long)batch * g.Ho * g.Wo * kh * kw;
  long blocks = (nw + 3) / 4; if (blocks > 256L * 256) blocks = 256L * 256; if (blocks < 1) blocks = 1;
  OrpProfScope prof(ORP_PROF_DCN_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL(dcn_col2im_nhwc_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grad_columns_t,
                     input_nhwc, offset, g, grad_input_nhwc, grad_offset);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
}
