// orp_hull.hpp -- Jarvis-march convex hull of a 9-point set on gfx950, reference-faithful.
//
// Mirrors the point ORDER and tie rules of the reference's Jarvis_and_index
//   (mmdet/ops/minarearect/src/minarearect_kernel.cu:215-341, float points + double cross product;
//    mmdet/ops/iou/src/convex_iou_kernel.cu:157-266 and convex_giou_kernel.cu:618-728, double points)
// including its quirks (pivot swapped to slot 0 during the scan while p_max / max_index are tracked in the same
// scan; `max_index == 0 -> 1` patch).  Storage is re-designed: instead of Point[20]/Stack[20] private arrays
// (dynamic indexing -> scratch) each lane owns LDS columns (orp_geom.hpp PolyLds), and the index stack is gone --
// the chain only ever needs its last point.  Each chain is capped at ORP_HULL_CAP steps (a 9-point march cannot
// legitimately exceed it; the reference would overrun its 20-entry stack instead).
#pragma once
#include "orp_geom.hpp"

namespace orp {

constexpr int ORP_HULL_CAP = 9;
constexpr int ORP_HULL_MAX = 2 * ORP_HULL_CAP;       // worst-case vertex count written to the hull store

// (a-o) x (b-o) with every operand promoted to double first
template <typename T>
ORP_HD double cross_d(Pt<T> o, Pt<T> a, Pt<T> b) {
  return ((double)a.x - (double)o.x) * ((double)b.y - (double)o.y) -
         ((double)b.x - (double)o.x) * ((double)a.y - (double)o.y);
}
template <typename T> ORP_HD T dis2(Pt<T> a, Pt<T> b) {
  return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y);
}

// IN : store holding the n input points (modified: pivot swaps, as the reference does)
// H  : store receiving the hull (>= ORP_HULL_MAX slots), L: scratch store (>= ORP_HULL_CAP + 1 slots)
// returns hull size
template <typename T, typename SI, typename SH, typename SL>
ORP_HD int jarvis_hull(SI& IN, int n, SH& H, SL& L) {
  Pt<T> p0 = IN.get(0);
  Pt<T> p_max = p0;
  int max_index = 0;
  for (int i = 0; i < n; i++) {
    Pt<T> pi = (i == 0) ? p0 : IN.get(i);
    if (pi.y < p0.y || (pi.y == p0.y && pi.x < p0.x)) {
      IN.set(0, pi); IN.set(i, p0);
      Pt<T> t = p0; p0 = pi; pi = t;
    }
    if (i == 0) { p_max = p0; max_index = 0; }
    if (pi.y > p_max.y || (pi.y == p_max.y && pi.x > p_max.x)) { p_max = pi; max_index = i; }
  }
  if (max_index == 0) { max_index = 1; p_max = IN.get(1); }

  // right chain: H[0..top1]
  int top1 = 0, k_index = 0;
  Pt<T> last = p0;
  H.set(0, p0);
  while (k_index != max_index && top1 < ORP_HULL_CAP) {
    Pt<T> p_k = p_max; k_index = max_index;
    for (int i = 1; i < n; i++) {
      Pt<T> pi = IN.get(i);
      double s = cross_d(last, pi, p_k);
      if (s > 0 || (s == 0 && dis2(last, pi) > dis2(last, p_k))) { p_k = pi; k_index = i; }
    }
    top1++;
    last = IN.get(k_index);          // == in_poly[Stack[top1]] (NOT p_k: p_max may be a stale copy)
    H.set(top1, last);
  }
  // left chain: L[0..top2]
  int top2 = 0; k_index = 0; last = p0;
  L.set(0, p0);
  while (k_index != max_index && top2 < ORP_HULL_CAP) {
    Pt<T> p_k = p_max; k_index = max_index;
    for (int i = 1; i < n; i++) {
      Pt<T> pi = IN.get(i);
      double s = cross_d(last, pi, p_k);
      if (s < 0 || (s == 0 && dis2(last, pi) > dis2(last, p_k))) { p_k = pi; k_index = i; }
    }
    top2++;
    last = IN.get(k_index);
    if (top2 <= ORP_HULL_CAP) L.set(top2, last);
  }
  // merged ring: right chain, then left chain top-down without its two end points
  for (int i = top1 + 1; i < top1 + top2; i++) H.set(i, L.get(top2 - (i - top1)));
  return top1 + top2;
}

}  // namespace orp
