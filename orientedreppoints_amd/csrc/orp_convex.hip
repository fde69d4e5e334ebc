// orp_convex.hip -- convex_iou: IoU(hull(9 points), gt quad) for all (point set, gt) pairs on gfx950.
//
// Replaces convex_iou_kernel + convex_iou_cuda (mmdet/ops/iou/src/convex_iou_kernel.cu:268-360), the op behind
// MaxIoUAssigner.assign (mmdet/core/bbox/assigners/max_iou_assigner.py:66).  Reference shape: one thread per point
// set looping over all K gts, the Jarvis hull recomputed K times, Point[100] private arrays, and a D2H -> host
// loop -> H2D round trip of N*K floats.  Here:
//   * lane = point set; the hull is built ONCE per lane and parked in an LDS column as float (hull vertices are
//     copies of the fp32 inputs, so the float -> double widening on load is exact);
//   * the gt quad is wave-uniform (scalar loads), grid.y splits the gt range so small N still fills the chip;
//   * fp64 clipping scratch is a per-lane LDS column of 8 + 8 vertices instead of ~3 KB of private stack;
//   * results go through a [64][gts per workgroup] LDS tile and are written to the [N, K] matrix as row pieces (a lane storing
//     one float at stride K touched a 128-byte line per float: 40 MB of writes for a 2.8 MB result) -- no host round trip.
// Arithmetic (fp64 internals, eps 1e-8, signed triangle fan WITHOUT fabs on each term, convex_iou_kernel.cu:137)
// follows the reference operation for operation: assignment decisions downstream compare these floats.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_hull.hpp"
#include "orp_quadfast.hpp"
#include "orp_prof.hpp"

namespace {
using orp::Pt;

constexpr int kThreads = 64;                  // one wave per workgroup: LDS per lane is what limits occupancy
constexpr int kHullKeep = 12;                 // stored hull vertices (a 9-point hull has <= 9)
constexpr int kMaxGtsPerBlock = 16;           // gts per workgroup: queue capacity = 64 lanes x this many, result tile 64 x this many

// float-backed store that widens to double on access (exact)
struct HullStoreF {
  Pt<float>* base; int stride;
  __device__ __forceinline__ Pt<double> get(int i) const { Pt<float> v = base[i * stride]; Pt<double> r; r.x = (double)v.x; r.y = (double)v.y; return r; }
  __device__ __forceinline__ void set(int i, Pt<double> v) const { Pt<float> f; f.x = (float)v.x; f.y = (float)v.y; base[i * stride] = f; }
};
// same, but silently drops writes past its capacity (degenerate / NaN inputs only)
template <int CAPACITY> struct HullStoreFCap {
  Pt<float>* base; int stride;
  __device__ __forceinline__ Pt<double> get(int i) const { Pt<float> v = base[(i < CAPACITY ? i : CAPACITY - 1) * stride]; Pt<double> r; r.x = (double)v.x; r.y = (double)v.y; return r; }
  __device__ __forceinline__ void set(int i, Pt<double> v) const { if (i < CAPACITY) { Pt<float> f; f.x = (float)v.x; f.y = (float)v.y; base[i * stride] = f; } }
};

__global__ void __launch_bounds__(kThreads)
convex_iou_kernel(const float* __restrict__ pts, int n, const float* __restrict__ gts, int k, int gts_per_block,
                  float* __restrict__ out) {
  // region A: first the hull builder's input (9) + left chain (10) as float points, later the fp64 clip scratch
  __shared__ __attribute__((aligned(16))) unsigned char s_a[2 * orp::ORP_CLIP_CAP * sizeof(Pt<double>) * kThreads];
  __shared__ Pt<float> s_hull[kHullKeep][kThreads];
  const int lane = threadIdx.x;
  const int idx = blockIdx.x * kThreads + lane;
  const bool active = idx < n;
  const int j0 = blockIdx.y * gts_per_block;
  const int j1 = min(k, j0 + gts_per_block);

  int n1 = 0;
  double s_pred = 0.0, hull_mabs = 0.0;
  bool finite_ok = true;
  HullStoreFCap<kHullKeep> H{&s_hull[0][lane], kThreads};
  if (active) {
    Pt<float>* fa = reinterpret_cast<Pt<float>*>(s_a);
    HullStoreF IN{fa + lane, kThreads};
    HullStoreF L{fa + 9 * kThreads + lane, kThreads};
    const float2* src = reinterpret_cast<const float2*>(pts + (size_t)idx * 18);
#pragma unroll
    for (int i = 0; i < 9; i++) { float2 v = src[i]; Pt<double> p; p.x = (double)v.x; p.y = (double)v.y; IN.set(i, p); }
    n1 = orp::jarvis_hull<double>(IN, 9, H, L);
    if (n1 > kHullKeep) n1 = kHullKeep;
    // orient CCW once (intersectAreaO: area(ps1) < 0 -> reverse1), then S_pred = area of the oriented ring
    s_pred = orp::poly_area<double>(H, n1);
    if (s_pred < 0) {
      for (int a = 0, b = n1 - 1; a < b; a++, b--) { Pt<double> t = H.get(a); H.set(a, H.get(b)); H.set(b, t); }
      s_pred = orp::poly_area<double>(H, n1);
    }
    for (int v = 0; v < n1; v++) {
      const Pt<double> p = H.get(v);
      hull_mabs = fmax(hull_mabs, fmax(fabs(p.x), fabs(p.y)));
      finite_ok = finite_ok && (fabs(p.x) < 1e100) && (fabs(p.y) < 1e100);
    }
  }
  __syncthreads();   // region A changes role (single wave, but keep the LDS ordering explicit)
  Pt<double>* da = reinterpret_cast<Pt<double>*>(s_a);
  orp::PolyLds<double> P{da + lane, kThreads};
  orp::PolyLds<double> Q{da + orp::ORP_CLIP_CAP * kThreads + lane, kThreads};

  // ---- phase A: lane = point set, gt wave-uniform.  Exact-zero classifier (fp64 twin of orp::pair_is_far,
  // orp_quadfast.hpp): proves without a division that every fan term of the (hull, gt) pair is exactly 0.
  //   cw_far : every hull vertex v is not strictly left of every ray O->w: X[v][w] = w.x*v.y - v.x*w.y <= eps is the
  //            reference's own stage-1 test, so each term dies there;
  //   ccw_far: every X > E and the stage-2 crossings stay away from the origin (beta * X[v][d] > E): all vertices that
  //            reach stage 3 are strictly right of d->O and only origin points remain.  E = 48*u*D*(M+D) + 1e-7 with
  //            u = 2^-53 (derivation in orp_quadfast.hpp; the absolute 1e-8 of sig() dominates in fp64).
  // Resolved pairs are written at once; the others are queued and evaluated densely in phase B (one pair per lane,
  // hull and gt fetched by index), instead of one straggler lane holding 63 finished ones.
  __shared__ unsigned short s_queue[kThreads * kMaxGtsPerBlock];
  __shared__ float s_out[kThreads][kMaxGtsPerBlock + 1];
  __shared__ int s_n1[kThreads];
  __shared__ double s_spred[kThreads];
  __shared__ int s_qcount;
  s_n1[lane] = n1; s_spred[lane] = s_pred;
  if (lane == 0) s_qcount = 0;
  __syncthreads();
  for (int j = j0; j < j1; j++) {
    const float* g = gts + (size_t)j * 8;          // wave-uniform
    Pt<double> q[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { q[t].x = (double)g[2 * t]; q[t].y = (double)g[2 * t + 1]; }
    auto area4 = [](const Pt<double>* v) {
      double res = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) res += v[i].x * v[(i + 1) & 3].y - v[i].y * v[(i + 1) & 3].x;
      return res / 2.0;
    };
    if (area4(q) < 0) { Pt<double> t = q[0]; q[0] = q[3]; q[3] = t; t = q[1]; q[1] = q[2]; q[2] = t; }
    const double s_gt = area4(q);
    const bool far = active && finite_ok && orp::hull_quad_is_far<double>(H, n1, hull_mabs, q);
    if (active && far) {
      const double inter0 = 0;
      const double uni0 = fabs(s_pred) + fabs(s_gt) - inter0;
      s_out[lane][j - j0] = (float)(inter0 / uni0);
    }
    const bool pend = active && !far;
    const unsigned long long pmask = __ballot(pend);
    if (pmask) {
      const int base = s_qcount;                       // single wave: uniform read, then lane 0 bumps it
      if (pend) s_queue[base + __popcll(pmask & ((1ull << lane) - 1ull))] = (unsigned short)(((j - j0) << 6) | lane);
      __syncthreads();
      if (lane == 0) s_qcount = base + __popcll(pmask);
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- phase B: one queued pair per lane ---------------------------------------------------------------------------
  const int nq = s_qcount;
  for (int q0 = 0; q0 < nq; q0 += kThreads) {
    const int qi = q0 + lane;
    if (qi >= nq) continue;
    const int item = s_queue[qi];
    const int sl = item & 63, j = j0 + (item >> 6);
    const int hn = s_n1[sl];
    const double hs = s_spred[sl];
    HullStoreFCap<kHullKeep> HS{&s_hull[0][sl], kThreads};
    const float* g = gts + (size_t)j * 8;
    Pt<double> q[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { q[t].x = (double)g[2 * t]; q[t].y = (double)g[2 * t + 1]; }
    auto area4 = [](const Pt<double>* v) {
      double res = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) res += v[i].x * v[(i + 1) & 3].y - v[i].y * v[(i + 1) & 3].x;
      return res / 2.0;
    };
    if (area4(q) < 0) { Pt<double> t = q[0]; q[0] = q[3]; q[3] = t; t = q[1]; q[1] = q[2]; q[2] = t; }
    const double s_gt = area4(q);
    // oriented gt fan triangles (tri_term swaps c,d when cross(O,c,d) < 0) and their stage-2/3 line constants
    orp::FanColT<double> gf[4];
    int gs[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      Pt<double> c = q[t], d = q[(t + 1) & 3];
      gs[t] = orp::sig(orp::cross3(Pt<double>{0.0, 0.0}, c, d));
      if (gs[t] == -1) { const Pt<double> tmp = c; c = d; d = tmp; }
      gf[t] = orp::fan_col<double>(c.x, c.y, d.x, d.y);
    }
    // register decision tree (orp_quadfast.hpp, fp64 instantiation, signed terms); a term that falls outside the tree
    // goes through the generic polygon loop on the per-lane LDS columns -- that one term, not the whole pair
    double inter = 0;
    Pt<double> a = HS.get(0);
    const Pt<double> h0 = a;
    for (int i = 0; i < hn; i++) {
      const Pt<double> b = (i + 1 < hn) ? HS.get(i + 1) : h0;
      const int s1 = orp::sig(orp::cross3(Pt<double>{0.0, 0.0}, a, b));
      if (s1 != 0) {
        const Pt<double> ea = (s1 == -1) ? b : a, eb = (s1 == -1) ? a : b;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (gs[t] == 0) continue;
          bool slow = false;
          double v = orp::tri_term_fast_t<double, false>(ea.x, ea.y, eb.x, eb.y, gf[t], slow);
          if (slow) {
            const Pt<double> c{gf[t].cx, gf[t].cy}, d{gf[t].dx, gf[t].dy};
            v = orp::tri_term_oriented_signed<double>(P, Q, ea, eb, c, d);
          }
          if (s1 * gs[t] == -1) v = -v;
          inter += v;
        }
      }
      a = b;
    }
    const double uni = fabs(hs) + fabs(s_gt) - inter;
    s_out[sl][j - j0] = (float)(inter / uni);
  }
  __syncthreads();
  // the tile's rows, piece by piece: consecutive lanes write consecutive floats of a row
  const int ng = j1 - j0;
  const int rows = min(kThreads, n - blockIdx.x * kThreads);
  for (int e = lane; e < rows * ng; e += kThreads) {
    const int r = e / ng, c = e - r * ng;
    out[(size_t)(blockIdx.x * kThreads + r) * k + j0 + c] = s_out[r][c];
  }
}
}  // namespace

extern "C" {
int orp_convex_iou(const float* pts, int n, const float* gts, int k, float* out, void* stream) {
  if (n < 0 || k < 0 || ((n > 0 && k > 0) && (!pts || !gts || !out))) return ORP_EINVAL;
  if (n == 0 || k == 0) return ORP_OK;
  const int nb = (n + kThreads - 1) / kThreads;
  // gts per workgroup: every workgroup rebuilds the hulls of its 64 point sets, so as many gts as possible behind one build --
  // while the launch still has ~2 000 workgroups (one wave each, LDS holds ~7 per CU) when the problem is large enough
  int gpb = kMaxGtsPerBlock;
  while (gpb > 1 && (long)nb * ((k + gpb - 1) / gpb) < 2048) gpb >>= 1;      // (1 280 / 4 096 measured no better: 21 824 x 64 in 387 / 341 us against 338)
  if (gpb > k) gpb = k;
  const int ysplit = (k + gpb - 1) / gpb;
  OrpProfScope prof(ORP_PROF_CONVEX_IOU, (hipStream_t)stream);
  hipLaunchKernelGGL(convex_iou_kernel, dim3(nb, ysplit), dim3(kThreads), 0, (hipStream_t)stream, pts, n, gts, k, gpb,
                     out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
}
