// orp_quadfast.hpp -- register-resident fast path of the fp32 quad-quad triangle-fan IoU (gfx950).
//
// Same arithmetic as orp::quad_iou (orp_geom.hpp), i.e. as the reference's devrIoU / devPolyIoU
// (mmdet/ops/nms/src/rnms_kernel.cu:16-147, DOTA_devkit/poly_nms_gpu/poly_nms_kernel.cu:36-212): every value that
// can reach the result is produced by the same fp32 operations on the same operands in the same order.  What
// changes is how the work is organised for a 64-wide wavefront:
//
//   * per-box work is hoisted into a QuadPrep record written once by a pre-pass (polygon orientation, the four
//     origin-fan triangles already oriented CCW, their orientation signs, |area|): the row box of a wave is then
//     scalar-loaded and the column box lives in registers for all rows the wave visits;
//   * a fan term  area(tri(O,a,b) ∩ tri(O,c,d))  is evaluated by a straight-line decision tree on the sign
//     patterns that occur in practice (measured on 32 M terms of dense DOTA-like scenes: 46 % die at the first
//     half-plane, 46 % at the third, the rest are triangles/quads of <= 4 vertices) with the polygon held in
//     registers -- no per-lane LDS polygon, no loops;
//   * exact zeros are recognised from signs alone: a clipped polygon with < 3 distinct vertices has a shoelace
//     sum of exactly 0, so  "both fan vertices not strictly left of O->c"  (stage 1) and  "every vertex strictly
//     right of d->O"  (stage 3) return 0 without computing a single intersection point;
//   * anything the tree does not cover exactly (a sign of 0 in an unexpected place, an eps-duplicate vertex, a
//     near-zero denominator, non-monotone sign patterns) is detected and handed to the generic loop
//     (orp::tri_term_oriented on per-lane LDS columns), so the result is bit-identical by construction.
//
// Why the shortcuts are exact (T = float, eps = 1e-8, sig() as in orp_geom.hpp):
//   - the first vertex of the working polygon is the origin; the reference holds a COMPUTED zero there (+-0), this
//     code holds +0.  A zero's sign can only change the sign of another exact zero (x - (+-0) = x, x * (+-0) = +-0),
//     and every consumer (sig(), same_pt(), |area|, inter += t with inter starting at +0) is blind to it;
//   - crossing points on an edge that starts or ends at the origin with a boundary value of exactly 0 are exactly
//     the origin; the de-duplication then collapses them, which is what the `near0` / `same` checks below mirror.
#pragma once
#include "orp_geom.hpp"

namespace orp {

// per-precision constants: unit roundoff factor of the classifier bound (48 * u), "infinity", finite-input bound
template <typename T> struct PrecT;
template <> struct PrecT<float>  { static ORP_HD float  e48u() { return 2.861e-6f; } static ORP_HD float  big() { return 3.0e38f; } static ORP_HD float  fin() { return 1e9f; } };
template <> struct PrecT<double> { static ORP_HD double e48u() { return 5.33e-15; }  static ORP_HD double big() { return 1e300; }   static ORP_HD double fin() { return 1e100; } };

template <typename T> struct QuadPrepT {   // float: 32 words = 128 B per box
  T ax[4], ay[4];            // fan triangle k = (O, a_k, b_k), oriented CCW (endpoints swapped when s[k] == -1)
  T bx[4], by[4];
  int s[4];                  // sig(cross(O, v_k, v_{k+1})) of the polygon edge BEFORE the swap; 0 = degenerate
  T vx[4], vy[4];            // the quad's vertices after the polygon-level re-orientation
  T area_abs;                // |shoelace/2| of the (re-oriented) quad
  int force_slow;            // non-finite / huge coordinates: use the generic path for every term of this box
  T mabs;                    // max |coordinate|
  T pad0;
};
typedef QuadPrepT<float> QuadPrep;

// quad8 = x1,y1,..,x4,y4 as stored in dets rows.  Mirrors the head of quad_iou(): polygon-level reversal when the
// signed area is negative, areas recomputed after the reversal, then the per-term orientation of tri_term().
template <typename T>
ORP_HD void quad_prepare(const T* q8, QuadPrepT<T>& o) {
  Pt<T> v[4];
  bool fin = true;
  T m = (T)0;
  for (int i = 0; i < 4; i++) {
    v[i].x = q8[2 * i]; v[i].y = q8[2 * i + 1];
    fin = fin && (orp_abs(v[i].x) < PrecT<T>::fin()) && (orp_abs(v[i].y) < PrecT<T>::fin());   // false for NaN / inf
    const T ax_ = orp_abs(v[i].x), ay_ = orp_abs(v[i].y);
    m = (ax_ > m) ? ax_ : m; m = (ay_ > m) ? ay_ : m;
  }
  T res = 0;
  for (int i = 0; i < 4; i++) res += v[i].x * v[(i + 1) & 3].y - v[i].y * v[(i + 1) & 3].x;
  if (res / (T)2 < 0) { Pt<T> t = v[0]; v[0] = v[3]; v[3] = t; t = v[1]; v[1] = v[2]; v[2] = t; }
  res = 0;
  for (int i = 0; i < 4; i++) res += v[i].x * v[(i + 1) & 3].y - v[i].y * v[(i + 1) & 3].x;
  o.area_abs = orp_abs(res / (T)2);
  o.force_slow = fin ? 0 : 1;
  o.mabs = m; o.pad0 = (T)0;
  for (int k = 0; k < 4; k++) {
    Pt<T> a = v[k], b = v[(k + 1) & 3];
    o.vx[k] = a.x; o.vy[k] = a.y;
    int s = sig<T>(a.x * b.y - b.x * a.y);                // cross3(o, a, b) with o = (0,0): (a.x-0)*(b.y-0) - ...
    if (s == -1) { Pt<T> t = a; a = b; b = t; }
    o.ax[k] = a.x; o.ay[k] = a.y; o.bx[k] = b.x; o.by[k] = b.y; o.s[k] = s;
  }
}

// |area| of tri(O,a,b) ∩ tri(O,c,d) for triangles that are ALREADY oriented CCW (generic loop; the slow path).
template <typename T, typename S1, typename S2>
ORP_HD T tri_term_oriented_signed(S1& P, S2& Q, Pt<T> a, Pt<T> b, Pt<T> c, Pt<T> d) {
  Pt<T> o; o.x = (T)0; o.y = (T)0;
  P.set(0, o); P.set(1, a); P.set(2, b);
  int n = 3;
  n = polygon_cut<T>(P, Q, n, o, c);
  n = polygon_cut<T>(P, Q, n, c, d);
  n = polygon_cut<T>(P, Q, n, d, o);
  return poly_area<T>(P, n);
}
template <typename T, typename S1, typename S2>
ORP_HD T tri_term_oriented(S1& P, S2& Q, Pt<T> a, Pt<T> b, Pt<T> c, Pt<T> d) {
  return orp_abs(tri_term_oriented_signed<T>(P, Q, a, b, c, d));
}

// sign predicates of sig() as single compares (NaN behaves as sig(NaN) = 0); T = float (quad IoU) or double (convex IoU)
constexpr float kEps = 1e-8f;
template <typename T> ORP_HD bool pos_(T d) { return d > (T)1e-8; }                       // sig(d) > 0
template <typename T> ORP_HD bool neg_(T d) { return d < -(T)1e-8; }                      // sig(d) < 0
template <typename T> ORP_HD bool zer_(T d) { return !(orp_abs(d) > (T)1e-8); }           // sig(d) == 0
template <typename T> ORP_HD bool near0(T x, T y) { return zer_(x) & zer_(y); }
template <typename T> ORP_HD bool same2(T ax, T ay, T bx, T by) { return zer_(ax - bx) & zer_(ay - by); }

// Column-side constants of one fan triangle (O, c, d), CCW: everything stage 2 / 3 need that does not depend on the
// other box.
template <typename T> struct FanColT {
  T cx, cy, dx, dy;
  T bax2, bay2, c0;          // line c->d: direction and its value at the origin
};
typedef FanColT<float> FanCol;
template <typename T> ORP_HD FanColT<T> fan_col(T cx, T cy, T dx, T dy) {
  FanColT<T> f; f.cx = cx; f.cy = cy; f.dx = dx; f.dy = dy;
  f.bax2 = dx - cx; f.bay2 = dy - cy;
  f.c0 = f.bax2 * ((T)0 - cy) - ((T)0 - cx) * f.bay2;
  return f;
}

// crossing of segment cur->nxt with the cutting line, from the line values at both ends (reference lineCross):
// X = (cur*cnxt - nxt*ccur) / (cnxt - ccur).  With cur or nxt = O = (+0,+0) this is the reference's expression up to
// the sign of an exact zero.
template <typename T>
ORP_HD void cross_pt(T curx, T cury, T ccur, T nxtx, T nxty, T cnxt, T& x, T& y, bool& bad) {
  const T den = cnxt - ccur;
  bad = bad | zer_(den);
  x = (curx * cnxt - nxtx * ccur) / den;
  y = (cury * cnxt - nxty * ccur) / den;
}

// Returns area(tri(O,a,b) ∩ tri(O,c,d)) (its absolute value when ABS_TERM, the fp32 quad kernels; signed, the fp64
// convex kernels) exactly as the generic polygon loop would, or sets `slow` when the decision tree does not cover the
// configuration (the caller then evaluates the generic loop; the value returned with slow set is meaningless).
// (ax,ay)->(bx,by) and f are oriented CCW.  Written for a 64-wide wave: predicates are single compares combined with
// non-short-circuit logic (scalar mask ops, no branches), one crossing routine per stage with selected operands (a
// wave whose lanes sit in different sign cases executes each division block once), early returns only where lanes
// commonly leave together.
template <typename T, bool ABS_TERM>
ORP_HD T tri_term_fast_t(T ax, T ay, T bx, T by, const FanColT<T>& f, bool& slow) {
  const T Z = (T)0;
  // ---- stage 1: keep left of O->c.  value(p) = c.x*p.y - p.x*c.y, value(O) = 0 exactly ------------------------
  const T ca = f.cx * ay - ax * f.cy;
  const T cb = f.cx * by - bx * f.cy;
  const bool a_pos = pos_(ca), b_pos = pos_(cb);
  if (!(a_pos | b_pos)) return Z;                        // <= 2 distinct vertices survive: area exactly 0
  bool bad = !b_pos | zer_(ca);                          // (+,-), (+,0), (0,+): not a strict CCW fan pattern
  T p1x = ax, p1y = ay;
  const T p2x = bx, p2y = by;
  if (neg_(ca)) cross_pt<T>(ax, ay, ca, bx, by, cb, p1x, p1y, bad);    // [O, X(a->b), b]
  bad = bad | near0(p1x, p1y) | near0(p2x, p2y) | same2(p1x, p1y, p2x, p2y);

  // ---- stage 2: keep left of c->d on [O, p1, p2] ------------------------------------------------------------------
  const T c1 = f.bax2 * (p1y - f.cy) - (p1x - f.cx) * f.bay2;
  const T c2 = f.bax2 * (p2y - f.cy) - (p2x - f.cx) * f.bay2;
  const bool n1 = neg_(c1), n2 = neg_(c2);
  bad = bad | !pos_(f.c0) | zer_(c1) | zer_(c2);
  // working polygon [O, w1, w2, (w3)]
  T w1x = p1x, w1y = p1y, w2x = p2x, w2y = p2y, w3x = p2x, w3y = p2y;
  bool four = false;
  if (n1 | n2) {
    // crossing A: the first sign change walking O -> p1 -> p2 -> O; crossing B: the second
    T Ax, Ay, Bx, By;
    cross_pt<T>(n1 ? Z : p1x, n1 ? Z : p1y, n1 ? f.c0 : c1, n1 ? p1x : p2x, n1 ? p1y : p2y, n1 ? c1 : c2, Ax, Ay, bad);
    cross_pt<T>(n2 ? p2x : p1x, n2 ? p2y : p1y, n2 ? c2 : c1, n2 ? Z : p2x, n2 ? Z : p2y, n2 ? f.c0 : c2, Bx, By, bad);
    w1x = n1 ? Ax : p1x; w1y = n1 ? Ay : p1y;           // (-,-): [O,A,B]   (-,+): [O,A,B,p2]   (+,-): [O,p1,A,B]
    w2x = n1 ? Bx : Ax;  w2y = n1 ? By : Ay;
    w3x = n1 ? p2x : Bx; w3y = n1 ? p2y : By;
    four = !(n1 & n2);
    bad = bad | near0(w1x, w1y) | same2(w1x, w1y, w2x, w2y) | (four & same2(w2x, w2y, w3x, w3y)) |
          near0(four ? w3x : w2x, four ? w3y : w2y);
  }

  // ---- stage 3: keep left of d->O.  value(O) = 0 exactly ---------------------------------------------------------
  const T bax3 = Z - f.dx, bay3 = Z - f.dy;
  const T t1 = bax3 * (w1y - f.dy) - (w1x - f.dx) * bay3;
  const T t2 = bax3 * (w2y - f.dy) - (w2x - f.dx) * bay3;
  const T t3 = bax3 * (w3y - f.dy) - (w3x - f.dx) * bay3;
  const bool u1p = pos_(t1), u2p = pos_(t2), u3p = pos_(t3);
  // exact zero: triangle with no strictly-left fan vertex, or quad with every vertex strictly right
  const bool dead = four ? (neg_(t1) & neg_(t2) & neg_(t3)) : !(u1p | u2p);
  if (dead) { slow = slow | bad; return Z; }
  bad = bad | !u1p | zer_(t2) | (four & (zer_(t3) | (!u2p & u3p)));
  // u1 > 0 from here on (or bad).  `two`: w2 kept; `cut`: some vertex is cut off -> one crossing X after the last kept
  const bool two = u2p;
  const bool cut = four ? !(two & u3p) : !two;
  T Xx = Z, Xy = Z;
  if (cut) {
    cross_pt<T>(two ? w2x : w1x, two ? w2y : w1y, two ? t2 : t1, two ? w3x : w2x, two ? w3y : w2y, two ? t3 : t2, Xx, Xy, bad);
    bad = bad | near0(Xx, Xy) | same2(two ? w2x : w1x, two ? w2y : w1y, Xx, Xy);
  }
  // shoelace over [O, w1, (w2), (w3 | X)]: the two terms touching O are exact zeros
  const T v2x = two ? w2x : Xx, v2y = two ? w2y : Xy;
  T res = w1x * v2y - w1y * v2x;
  if (two & four) {
    const T v3x = cut ? Xx : w3x, v3y = cut ? Xy : w3y;
    res += w2x * v3y - w2y * v3x;
  }
  slow = slow | bad;
  res = res / (T)2;
  return ABS_TERM ? orp_abs(res) : res;
}
ORP_HD float tri_term_fast(float ax, float ay, float bx, float by, const FanCol& f, bool& slow) {
  return tri_term_fast_t<float, true>(ax, ay, bx, by, f, slow);
}

// ---- pair-level exact-zero classifier ("phase A") ---------------------------------------------------------------
// Decides, without a single division, that EVERY fan term of a (row, col) pair is exactly 0, i.e. inter = +0:
//   cw_far : every row vertex v is not strictly left of every ray O->w (w a column vertex):  X[v][w] <= eps for all
//            16 vertex pairs, X[v][w] = w.x*v.y - v.x*w.y.  X is the reference's own stage-1 expression, so each term
//            dies at stage 1 -- exact, no error analysis involved;
//   ccw_far: every X[v][w] > E and, for every column edge j, the stage-2 crossings stay away from the origin.  Then every
//            term passes stage 1 unchanged ([O,a,b]), whatever stage 2 does its output vertices are O, a, b, points of
//            segment a-b, or beta*a / beta*b with beta = c0/(c0 - c(v)) in (0,1), each computed to within 4.1*u*M of
//            that real point, and the reference's stage-3 value t(w) = fl(-dx*(w.y-dy) + (w.x-dx)*dy) of each of them is
//            <= F(w*) + 14.3*u*D*(M+D) with F(w*) = beta * (-(d x v)) <= -beta*(X[v][d] - 4.01*u*D*M): below -1e-8 as
//            soon as beta * X[v][d] > 18.4*u*D*(M+D) + 1e-8.  All non-origin vertices strictly right of d->O leaves only
//            origin points after stage 3 and a shoelace sum of exactly 0.  (u = 2^-24, M / D = max |coordinate| of the
//            row / column box.)  E below is 48*u*D*(M+D) + 1e-7: a 2.6x margin over that bound, which also absorbs the
//            rounding of the test itself.
// Anything else is "unresolved" and gets the full evaluation.  Degenerate column edges (s == 0) are skipped by the
// reference, so they impose no condition.
template <typename T> struct FarColT {      // per-lane column constants of the classifier
  T wx[4], wy[4];               // column vertices
  T cx[4], cy[4], bax2[4], bay2[4], c0[4];   // oriented column edges: start point, direction, value at O
  int s[4];
  T mabs;
};
typedef FarColT<float> FarCol;
template <typename T>
ORP_HD FarColT<T> far_col(const QuadPrepT<T>& p) {
  FarColT<T> f;
  for (int j = 0; j < 4; j++) {
    f.wx[j] = p.vx[j]; f.wy[j] = p.vy[j];
    const FanColT<T> t = fan_col<T>(p.ax[j], p.ay[j], p.bx[j], p.by[j]);
    f.cx[j] = t.cx; f.cy[j] = t.cy; f.bax2[j] = t.bax2; f.bay2[j] = t.bay2; f.c0[j] = t.c0; f.s[j] = p.s[j];
  }
  f.mabs = p.mabs;
  return f;
}
// rvx/rvy: the row box's vertices, rm its mabs.  Returns true when inter == +0 exactly.
template <typename T>
ORP_HD bool pair_is_far(const T* rvx, const T* rvy, T rm, const FarColT<T>& c) {
  T mx = -PrecT<T>::big(), mn = PrecT<T>::big(), mnv[4];
#pragma unroll
  for (int v = 0; v < 4; v++) {
    T m = PrecT<T>::big();
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const T x = c.wx[w] * rvy[v] - rvx[v] * c.wy[w];
      mx = (x > mx) ? x : mx;
      m = (x < m) ? x : m;
    }
    mnv[v] = m;
    mn = (m < mn) ? m : mn;
  }
  if (!(mx > (T)1e-8)) return true;                            // cw_far (NaN-free: callers exclude force_slow boxes)
  const T E = PrecT<T>::e48u() * c.mabs * (rm + c.mabs) + (T)1e-7;   // 48 * u * D * (M + D) + 1e-7
  if (!(mn > E)) return false;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    bool okj = pos_(c.c0[j]);
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const T cv = c.bax2[j] * (rvy[v] - c.cy[j]) - (rvx[v] - c.cx[j]) * c.bay2[j];
      okj = okj & (pos_(cv) | (neg_(cv) & (mnv[v] * c.c0[j] > E * (c.c0[j] - cv))));
    }
    ok = ok & (okj | (c.s[j] == 0));
  }
  return ok;
}

// ---- per-TERM exact-zero screen ("phase B1") ----------------------------------------------------------------------
// For a pair the classifier leaves unresolved, decides term by term which of the 16 fan terms are exactly 0 -- again
// without a division -- so that only the others (about 1 in 8 on dense DOTA-like scenes) go through the decision tree:
//   kill-1: neither fan vertex of the row edge is strictly left of O->c:  ca <= eps and cb <= eps, the test
//           tri_term_fast_t starts with (<= 2 distinct vertices survive stage 1: area exactly 0);
//   kill-3: the per-term form of ccw_far above.  With ca, cb > E stage 1 leaves [O, a, b] (plus exact-origin crossings
//           the de-duplication folds into O); stage 2 can only add points of the segments O-a, a-b, b-O; under
//           X[v][d] > E and, for v in {a, b},  c(v) > eps  or  (c(v) < -eps and beta * X[v][d] > E, beta = c0/(c0-c(v)))
//           and c0 > eps, the reference's stage-3 value of every such non-origin vertex is < -1e-8 (same bound as
//           ccw_far: that analysis is per term, the pair test only ANDs it over the 16 terms with min X in place of
//           X[v][d]), every denominator of a crossing actually computed exceeds eps, the origin's stage-3 value is an
//           exact 0, so stage 3 leaves origin points only and the shoelace sum is exactly 0.
// A term with a degenerate edge (s == 0) is skipped by the reference.  Bit 4*i + j of the result = term (row edge i,
// column edge j) must be evaluated.  Dropping exact-zero terms does not change the sum: inter starts at +0 and
// x + (+-0) == x.  Inputs: oriented fan edges (a -> b per row edge, c -> d per column edge), signs, max |coordinate|.
template <typename T>
ORP_HD unsigned pair_term_alive_mask(const T* rax, const T* ray, const T* rbx, const T* rby, const int* rs, T rm,
                                     const T* ccx, const T* ccy, const T* cdx, const T* cdy, const int* cs, T cm) {
  const T E = PrecT<T>::e48u() * cm * (rm + cm) + (T)1e-7;          // 48 * u * D * (M + D) + 1e-7
  // row vertex v = start of polygon edge v (the oriented edge is swapped when s == -1)
  T vx[4], vy[4];
#pragma unroll
  for (int v = 0; v < 4; v++) { const bool sw = rs[v] < 0; vx[v] = sw ? rbx[v] : rax[v]; vy[v] = sw ? rby[v] : ray[v]; }
  unsigned alive = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const T bax2 = cdx[j] - ccx[j], bay2 = cdy[j] - ccy[j];
    const T c0 = bax2 * ((T)0 - ccy[j]) - ((T)0 - ccx[j]) * bay2;
    const bool c0_pos = pos_(c0);
    bool notleft[4], beyond[4];                                      // per row vertex, against column edge j
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const T xc = ccx[j] * vy[v] - vx[v] * ccy[j];                  // the reference's stage-1 value of v
      const T xd = cdx[j] * vy[v] - vx[v] * cdy[j];
      const T cv = bax2 * (vy[v] - ccy[j]) - (vx[v] - ccx[j]) * bay2;
      notleft[v] = !pos_(xc);
      beyond[v] = (xc > E) & (xd > E) & (pos_(cv) | (neg_(cv) & (xd * c0 > E * (c0 - cv))));
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int i1 = (i + 1) & 3;
      const bool kill1 = notleft[i] & notleft[i1];
      const bool kill3 = c0_pos & beyond[i] & beyond[i1];
      const bool live = (rs[i] != 0) & (cs[j] != 0) & !(kill1 | kill3);
      alive |= live ? (1u << (4 * i + j)) : 0u;
    }
  }
  return alive;
}

// The composition of the term-queue kernels as one function (what tests/host_harness checks against the oracle):
// classifier, per-term screen, decision tree for the surviving terms (generic polygon loop for a term the tree does not
// cover) summed in the reference's order.  stats[0] += classifier-resolved pairs, stats[1] += generic TERMS,
// stats[2] += terms evaluated, stats[3] += pairs the per-term screen emptied.
template <typename T, bool GUARD>
ORP_HD T quad_iou_term_queue_t(const QuadPrepT<T>* r, const QuadPrepT<T>* c, long long* stats = nullptr);

// ---- the same classifier for (convex hull of <= 12 vertices, quad) pairs: convex_iou (fp64) -------------------------
// H.get(v) = hull vertex v (CCW), q[0..3] = the gt quad re-oriented CCW as convex_iou's intersectAreaO does; hull_mabs =
// max |coordinate| of the hull.  Returns true when every fan term (hull edge i, quad edge j) is exactly 0.  The hull is
// the "row" (a, b) side and the quad the "column" (c, d) side of the analysis above, with M = hull_mabs, D = max |q|.
template <typename T, typename HS>
ORP_HD bool hull_quad_is_far(const HS& H, int n1, T hull_mabs, const Pt<T>* q) {
  T gD = (T)0;
  for (int t = 0; t < 4; t++) { const T ax_ = orp_abs(q[t].x), ay_ = orp_abs(q[t].y); gD = (ax_ > gD) ? ax_ : gD; gD = (ay_ > gD) ? ay_ : gD; }
  if (!(gD < PrecT<T>::fin())) return false;
  T mx = -PrecT<T>::big(), mn = PrecT<T>::big();
  for (int v = 0; v < n1; v++) {
    const Pt<T> p = H.get(v);
    for (int w = 0; w < 4; w++) {
      const T x = q[w].x * p.y - p.x * q[w].y;
      mx = (x > mx) ? x : mx; mn = (x < mn) ? x : mn;
    }
  }
  if (!(mx > (T)1e-8)) return true;                                   // cw_far: every term dies at stage 1
  const T E = PrecT<T>::e48u() * gD * (hull_mabs + gD) + (T)1e-7;
  if (!(mn > E)) return false;
  bool ok = true;
  for (int t = 0; t < 4; t++) {
    // oriented gt edge (tri_term swaps c,d when cross(O,c,d) < 0), its direction and value at the origin
    Pt<T> c = q[t], d = q[(t + 1) & 3];
    Pt<T> o; o.x = (T)0; o.y = (T)0;
    const int s2 = sig<T>(cross3<T>(o, c, d));
    if (s2 == 0) continue;                                            // the reference skips degenerate gt edges
    if (s2 == -1) { const Pt<T> tmp = c; c = d; d = tmp; }
    const T bax = d.x - c.x, bay = d.y - c.y;
    const T c0 = bax * ((T)0 - c.y) - ((T)0 - c.x) * bay;
    bool okt = pos_(c0);
    for (int v = 0; v < n1; v++) {
      const Pt<T> p = H.get(v);
      T mnv = PrecT<T>::big();
      for (int w = 0; w < 4; w++) { const T x = q[w].x * p.y - p.x * q[w].y; mnv = (x < mnv) ? x : mnv; }
      const T cv = bax * (p.y - c.y) - (p.x - c.x) * bay;
      okt = okt & (pos_(cv) | (neg_(cv) & (mnv * c0 > E * (c0 - cv))));
    }
    ok = ok & okt;
  }
  return ok;
}

// Column box held in registers by a lane: the four fan triangles + per-box scalars.
template <typename T> struct QuadColT {
  FanColT<T> f[4];
  int s[4];
  T area_abs;
  int force_slow;
};
typedef QuadColT<float> QuadCol;
template <typename T>
ORP_HD QuadColT<T> quad_col(const QuadPrepT<T>& p) {
  QuadColT<T> c;
  for (int j = 0; j < 4; j++) { c.f[j] = fan_col<T>(p.ax[j], p.ay[j], p.bx[j], p.by[j]); c.s[j] = p.s[j]; }
  c.area_abs = p.area_abs; c.force_slow = p.force_slow;
  return c;
}

// Generic evaluation of one prepared pair (every term through the polygon loop).  Rare path: scratch-resident
// private polygons, one code instance.
template <typename T, bool GUARD>
ORP_HD T quad_iou_prepared_generic_t(const QuadPrepT<T>* r, const QuadColT<T>& c) {
  PolyPriv<T, ORP_CLIP_CAP> P, Q;
  T inter = (T)0;
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const int s1 = r->s[i];
    if (s1 == 0) continue;
    Pt<T> a, b;
    a.x = r->ax[i]; a.y = r->ay[i]; b.x = r->bx[i]; b.y = r->by[i];
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      FanColT<T> f = c.f[0]; int s2 = c.s[0];
      if (j == 1) { f = c.f[1]; s2 = c.s[1]; } else if (j == 2) { f = c.f[2]; s2 = c.s[2]; } else if (j == 3) { f = c.f[3]; s2 = c.s[3]; }
      if (s2 == 0) continue;
      Pt<T> cc, d;
      cc.x = f.cx; cc.y = f.cy; d.x = f.dx; d.y = f.dy;
      T t = tri_term_oriented<T>(P, Q, a, b, cc, d);
      if (s1 * s2 == -1) t = -t;
      inter += t;
    }
  }
  const T uni = r->area_abs + c.area_abs - inter;
  if (GUARD) { if (uni == (T)0) return (inter + (T)1) / (uni + (T)1); }
  return inter / uni;
}

// IoU of the prepared row box `r` (wave-uniform: scalar loads on the GPU) with the register-resident column box.
// Term order (row edge outer, column edge inner) and the accumulation order are those of quad_iou().  A term the
// decision tree does not cover is evaluated by the generic polygon loop -- that term only.
// `nslow` (optional, host statistics) counts pairs that needed the generic loop for at least one term.
template <typename T, bool GUARD>
ORP_HD T quad_iou_prepared_t(const QuadPrepT<T>* r, const QuadColT<T>& c, int* nslow = nullptr) {
  if ((r->force_slow | c.force_slow) != 0) {
    if (nslow) (*nslow)++;
    return quad_iou_prepared_generic_t<T, GUARD>(r, c);
  }
  PolyPriv<T, ORP_CLIP_CAP> P, Q;
  T inter = (T)0;
  bool any_slow = false;
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const int s1 = r->s[i];
    if (s1 == 0) continue;
    const T ax = r->ax[i], ay = r->ay[i], bx = r->bx[i], by = r->by[i];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (c.s[j] == 0) continue;
      bool slow = false;
      T t = tri_term_fast_t<T, true>(ax, ay, bx, by, c.f[j], slow);
      if (slow) {
        Pt<T> a, b, cc, d;
        a.x = ax; a.y = ay; b.x = bx; b.y = by;
        cc.x = c.f[j].cx; cc.y = c.f[j].cy; d.x = c.f[j].dx; d.y = c.f[j].dy;
        t = tri_term_oriented<T>(P, Q, a, b, cc, d);
        any_slow = true;
      }
      if (s1 * c.s[j] == -1) t = -t;
      inter += t;
    }
  }
  if (any_slow && nslow) (*nslow)++;
  const T uni = r->area_abs + c.area_abs - inter;
  if (GUARD) { if (uni == (T)0) return (inter + (T)1) / (uni + (T)1); }
  return inter / uni;
}
template <bool GUARD>
ORP_HD float quad_iou_prepared(const QuadPrep* r, const QuadCol& c, int* nslow = nullptr) {
  return quad_iou_prepared_t<float, GUARD>(r, c, nslow);
}

// IoU value of a pair whose intersection is exactly +0 (what quad_iou() returns when every term is 0).
template <typename T, bool GUARD>
ORP_HD T iou_of_zero_inter_t(T row_area_abs, T col_area_abs) {
  const T inter = (T)0;
  const T uni = row_area_abs + col_area_abs - inter;
  if (GUARD) { if (uni == (T)0) return (inter + (T)1) / (uni + (T)1); }
  return inter / uni;
}
template <bool GUARD>
ORP_HD float iou_of_zero_inter(float row_area_abs, float col_area_abs) {
  return iou_of_zero_inter_t<float, GUARD>(row_area_abs, col_area_abs);
}

// The composition the kernels implement (classifier, then full evaluation of what it leaves), as one function: this
// is what tests/host_harness runs on the CPU against the oracle, and what the fp64 merge-NMS kernel calls per pair.
// stats[0] += pairs resolved by the classifier, stats[1] += pairs on the generic path.
template <typename T, bool GUARD>
ORP_HD T quad_iou_two_phase_t(const QuadPrepT<T>* r, const QuadPrepT<T>* c, long long* stats = nullptr) {
  if ((r->force_slow | c->force_slow) == 0) {
    const FarColT<T> fc = far_col<T>(*c);
    if (pair_is_far<T>(r->vx, r->vy, r->mabs, fc)) {
      if (stats) stats[0]++;
      return iou_of_zero_inter_t<T, GUARD>(r->area_abs, c->area_abs);
    }
  }
  const QuadColT<T> qc = quad_col<T>(*c);
  int nslow = 0;
  const T v = quad_iou_prepared_t<T, GUARD>(r, qc, &nslow);
  if (stats) stats[1] += nslow;
  return v;
}
template <bool GUARD>
ORP_HD float quad_iou_two_phase(const QuadPrep* r, const QuadPrep* c, long long* stats = nullptr) {
  return quad_iou_two_phase_t<float, GUARD>(r, c, stats);
}

template <typename T, bool GUARD>
ORP_HD T quad_iou_term_queue_t(const QuadPrepT<T>* r, const QuadPrepT<T>* c, long long* stats) {
  const bool forced = (r->force_slow | c->force_slow) != 0;
  unsigned alive = 0u;
  if (forced) {
    for (int t = 0; t < 16; t++) alive |= (r->s[t >> 2] != 0 && c->s[t & 3] != 0) ? (1u << t) : 0u;
  } else {
    const FarColT<T> fc = far_col<T>(*c);
    if (pair_is_far<T>(r->vx, r->vy, r->mabs, fc)) {
      if (stats) stats[0]++;
      return iou_of_zero_inter_t<T, GUARD>(r->area_abs, c->area_abs);
    }
    alive = pair_term_alive_mask<T>(r->ax, r->ay, r->bx, r->by, r->s, r->mabs, c->ax, c->ay, c->bx, c->by, c->s, c->mabs);
  }
  if (alive == 0u) {
    if (stats) stats[3]++;
    return iou_of_zero_inter_t<T, GUARD>(r->area_abs, c->area_abs);
  }
  PolyPriv<T, ORP_CLIP_CAP> P, Q;
  T inter = (T)0;
  for (int t = 0; t < 16; t++) {
    if (!((alive >> t) & 1u)) continue;
    const int i = t >> 2, j = t & 3;
    bool slow = forced;
    T v = (T)0;
    if (!slow) v = tri_term_fast_t<T, true>(r->ax[i], r->ay[i], r->bx[i], r->by[i], fan_col<T>(c->ax[j], c->ay[j], c->bx[j], c->by[j]), slow);
    if (slow) {                                          // this one term through the generic polygon loop
      Pt<T> a, b, cc, d;
      a.x = r->ax[i]; a.y = r->ay[i]; b.x = r->bx[i]; b.y = r->by[i];
      cc.x = c->ax[j]; cc.y = c->ay[j]; d.x = c->bx[j]; d.y = c->by[j];
      v = tri_term_oriented<T>(P, Q, a, b, cc, d);
      if (stats) stats[1]++;
    }
    if (r->s[i] * c->s[j] == -1) v = -v;
    inter += v;
    if (stats) stats[2]++;
  }
  const T uni = r->area_abs + c->area_abs - inter;
  if (GUARD) { if (uni == (T)0) return (inter + (T)1) / (uni + (T)1); }
  return inter / uni;
}

}  // namespace orp
