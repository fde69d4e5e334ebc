// orp_geom.hpp -- device geometry core shared by the rotated-IoU / NMS / convex-IoU kernels (gfx950).
//
// Convex-polygon intersection by a signed triangle fan anchored at the coordinate origin -- the algorithm the
// reference uses in rnms_kernel.cu:16-147, poly_nms_kernel.cu:36-212, poly_overlaps_kernel.cu:36-328,
// polyiou.cpp:9-128 and convex_iou_kernel.cu:21-154 (reference = /root/reference, LiWentomng/OrientedRepPoints).
//
// Contract: every floating-point operation of the reference is performed with the same operands, in the same
// order and precision (compile with -ffp-contract=off, IEEE division), because `iou > thr` decisions of the NMS
// must be bit-exact.  What is NOT taken from the reference is the storage: the reference keeps Point[510]
// per-thread arrays (8 KB of scratch per thread); here a clipped triangle never exceeds ORP_CLIP_CAP = 8
// vertices and lives in a per-lane LDS column (stride = workgroup size, so a wave's accesses to one slot are
// bank-conflict free whatever slot each lane is at) or, for the PolyRegs variant, in registers.
#pragma once
// The header is plain C++: under hipcc every function is a __device__ inline; under a host compiler (g++) the very
// same source builds as ordinary inline functions, which is how tests/ check the fp32 operation order on the CPU
// against the oracle before a GPU is involved (tests/host_harness).
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORP_HD __host__ __device__ __forceinline__
#else
#define ORP_HD inline
#endif

namespace orp {

template <typename T> struct Pt { T x, y; };

ORP_HD float orp_abs(float x) { return __builtin_fabsf(x); }
ORP_HD double orp_abs(double x) { return __builtin_fabs(x); }

template <typename T> ORP_HD int sig(T d) {
  return (int)(d > (T)1E-8) - (int)(d < -(T)1E-8);
}
template <typename T> ORP_HD bool same_pt(Pt<T> a, Pt<T> b) {
  return sig(a.x - b.x) == 0 && sig(a.y - b.y) == 0;
}
template <typename T> ORP_HD T cross3(Pt<T> o, Pt<T> a, Pt<T> b) {
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

// ---- polygon scratch stores ---------------------------------------------------------------------------------
// (a) per-lane column in LDS: element i of this lane's polygon lives at base[i * stride]
template <typename T> struct PolyLds {
  Pt<T>* base; int stride;
  ORP_HD Pt<T> get(int i) const { return base[i * stride]; }
  ORP_HD void set(int i, Pt<T> v) const { base[i * stride] = v; }
};
// (b) private array (the compiler decides between registers and scratch); used by low-volume kernels
template <typename T, int CAP> struct PolyPriv {
  Pt<T> v[CAP];
  ORP_HD Pt<T> get(int i) const { return v[i]; }
  ORP_HD void set(int i, Pt<T> p) { v[i] = p; }
};

constexpr int ORP_CLIP_CAP = 8;   // clipped triangle: 3 -> <=4 -> <=5 -> <=6 vertices (+ eps-sign duplicates)

// Keep the part of polygon P (n vertices) left of a->b; result back in P (reference polygon_cut).
// S1 = working polygon store, S2 = scratch store.  Returns the new vertex count.
template <typename T, typename S1, typename S2>
ORP_HD int polygon_cut(S1& P, S2& Q, int n, Pt<T> a, Pt<T> b) {
  if (n == 0) return 0;
  int m = 0;
  const T bax = b.x - a.x, bay = b.y - a.y;     // loop-invariant sub-expressions of cross3(a, b, .)
  Pt<T> p0 = P.get(0);
  Pt<T> cur = p0;
  T ccur = bax * (cur.y - a.y) - (cur.x - a.x) * bay;
  int scur = sig(ccur);
  for (int i = 0; i < n; i++) {
    Pt<T> nxt = (i + 1 < n) ? P.get(i + 1) : p0;
    T cnxt = bax * (nxt.y - a.y) - (nxt.x - a.x) * bay;
    int snxt = sig(cnxt);
    if (scur > 0) { if (m < ORP_CLIP_CAP) Q.set(m, cur); m++; }
    if (scur != snxt) {
      // lineCross(a, b, cur, nxt): s1 = cross(a,b,cur) = ccur, s2 = cross(a,b,nxt) = cnxt
      Pt<T> x; x.x = (T)0; x.y = (T)0;
      bool both0 = (scur == 0 && snxt == 0);
      T den = cnxt - ccur;
      if (!both0 && sig(den) != 0) {
        x.x = (cur.x * cnxt - nxt.x * ccur) / den;
        x.y = (cur.y * cnxt - nxt.y * ccur) / den;
      }
      if (m < ORP_CLIP_CAP) Q.set(m, x);
      m++;
    }
    cur = nxt; ccur = cnxt; scur = snxt;
  }
  if (m > ORP_CLIP_CAP) m = ORP_CLIP_CAP;
  // drop consecutive eps-duplicates, then trailing duplicates of the first vertex
  int k = 0;
  Pt<T> prev, first;
  for (int i = 0; i < m; i++) {
    Pt<T> v = Q.get(i);
    if (i == 0) first = v;
    if (i == 0 || !same_pt(v, prev)) { P.set(k, v); k++; }
    prev = v;
  }
  while (k > 1 && same_pt(P.get(k - 1), first)) k--;
  return k;
}

// signed shoelace / 2 of the polygon in P (n vertices)
template <typename T, typename S1>
ORP_HD T poly_area(const S1& P, int n) {
  T res = 0;
  if (n == 0) return res / (T)2;
  Pt<T> p0 = P.get(0), cur = p0;
  for (int i = 0; i < n; i++) {
    Pt<T> nxt = (i + 1 < n) ? P.get(i + 1) : p0;
    res += cur.x * nxt.y - cur.y * nxt.x;
    cur = nxt;
  }
  return res / (T)2;
}

// signed area of triangle(O,a,b) ∩ triangle(O,c,d)   (reference intersectArea(a,b,c,d))
template <typename T, bool ABS_TERM, typename S1, typename S2>
ORP_HD T tri_term(S1& P, S2& Q, Pt<T> a, Pt<T> b, Pt<T> c, Pt<T> d) {
  Pt<T> o; o.x = (T)0; o.y = (T)0;
  int s1 = sig(cross3(o, a, b));
  int s2 = sig(cross3(o, c, d));
  if (s1 == 0 || s2 == 0) return (T)0;
  if (s1 == -1) { Pt<T> t = a; a = b; b = t; }
  if (s2 == -1) { Pt<T> t = c; c = d; d = t; }
  P.set(0, o); P.set(1, a); P.set(2, b);
  int n = 3;
  n = polygon_cut<T>(P, Q, n, o, c);
  n = polygon_cut<T>(P, Q, n, c, d);
  n = polygon_cut<T>(P, Q, n, d, o);
  T res = poly_area<T>(P, n);
  if (ABS_TERM) res = orp_abs(res);
  if (s1 * s2 == -1) res = -res;
  return res;
}

// Small fixed polygons (quads, hulls <= 9) held in registers / private memory.
template <typename T, int N> struct SmallPoly {
  Pt<T> v[N + 1];
  int n;
};

template <typename T, int N>
ORP_HD T small_area(const SmallPoly<T, N>& s) {
  T res = 0;
  for (int i = 0; i < N; i++) {
    if (i < s.n) {
      Pt<T> cur = s.v[i];
      Pt<T> nxt = (i + 1 < s.n) ? s.v[i + 1] : s.v[0];
      res += cur.x * nxt.y - cur.y * nxt.x;
    }
  }
  return res / (T)2;
}

template <typename T, int N>
ORP_HD void small_reverse(SmallPoly<T, N>& s) {
  for (int i = 0; i < N / 2; i++) {
    int j = s.n - 1 - i;
    if (i < j) { Pt<T> t = s.v[i]; s.v[i] = s.v[j]; s.v[j] = t; }
  }
}

// quad-quad specialisation: everything unrolled, vertices in registers (reference devrIoU / devPolyIoU /
// iou_poly).  GUARD adds poly_nms' `union == 0 -> (i+1)/(u+1)` rule (poly_nms_kernel.cu:205-207).
template <typename T, bool GUARD, typename S1, typename S2>
ORP_HD T quad_iou(S1& P, S2& Q, const T* p8, const T* q8) {
  Pt<T> a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i].x = p8[2 * i]; a[i].y = p8[2 * i + 1]; b[i].x = q8[2 * i]; b[i].y = q8[2 * i + 1]; }
  // orientation: area(ps) < 0 -> reverse; the unions' |area| are taken AFTER the reversal, same summation order
  auto area4 = [](const Pt<T>* v) {
    T res = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) res += v[i].x * v[(i + 1) & 3].y - v[i].y * v[(i + 1) & 3].x;
    return res / (T)2;
  };
  if (area4(a) < 0) { Pt<T> t = a[0]; a[0] = a[3]; a[3] = t; t = a[1]; a[1] = a[2]; a[2] = t; }
  if (area4(b) < 0) { Pt<T> t = b[0]; b[0] = b[3]; b[3] = t; t = b[1]; b[1] = b[2]; b[2] = t; }
  T inter = 0;
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      // dynamic vertex pick without dynamic register indexing
      Pt<T> ai = a[0], aj = a[1], bi = b[0], bj = b[1];
      if (i == 1) { ai = a[1]; aj = a[2]; } else if (i == 2) { ai = a[2]; aj = a[3]; } else if (i == 3) { ai = a[3]; aj = a[0]; }
      if (j == 1) { bi = b[1]; bj = b[2]; } else if (j == 2) { bi = b[2]; bj = b[3]; } else if (j == 3) { bi = b[3]; bj = b[0]; }
      inter += tri_term<T, true>(P, Q, ai, aj, bi, bj);
    }
  }
  T uni = orp_abs(area4(a)) + orp_abs(area4(b)) - inter;
  if (GUARD) { if (uni == (T)0) return (inter + (T)1) / (uni + (T)1); }
  return inter / uni;
}

}  // namespace orp
