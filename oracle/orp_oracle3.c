/* oracle/orp_oracle3.c -- CPU ORACLE part 3 (TEST INFRASTRUCTURE ONLY): convex GIoU and its gradient.
 *
 * Restates devrIoU of mmdet/ops/iou/src/convex_giou_kernel.cu:730-804 and what it calls:
 *   Jarvis_and_index :618-728, intersectAreaO :440-452, intersectArea(+grad) :213-438, polygon_cut(+grad) :176-211,
 *   lineCross(+grad) :122-175, polygen_area_grad :73-120, intersectAreaPoly :544-615, Jarvis :454-542.
 *
 * Same fp64 arithmetic for the VALUES (giou bit-comparable with the reference).  The GRADIENT is restated in
 * reverse mode: the reference builds the dense Jacobians of the three polygon cuts ([2n x 2k] each) and multiplies
 * them (p3_p2 * p2_p1 * p1_p), here the area gradient is pulled back through the cuts one vertex at a time --
 * mathematically the same product, different summation order (parity bar on gradients: 1e-4, not bits).
 *
 * The reference's scratch arrays are too small for a clipped polygon of 6 vertices (ccur_grad/cut_grad[100] hold
 * 4*k*n doubles, p*_grad[10][10] hold 2n <= 10): there it reads/writes out of bounds (undefined).  This restatement
 * has no such limit and REPORTS those rows (orc_convex_giou flag output) so that parity tests can skip them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y; } dpt;
#define GCAP 16

static int gsig(double d) { return (d > 1E-8) - (d < -1E-8); }
static int gsame(dpt a, dpt b) { return gsig(a.x - b.x) == 0 && gsig(a.y - b.y) == 0; }
static double gcross(dpt o, dpt a, dpt b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }
static double gdis(dpt a, dpt b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y); }
static double garea(dpt* ps, int n) {
  ps[n] = ps[0];
  double res = 0;
  for (int i = 0; i < n; i++) res += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
  return res / 2.0;
}

/* one output vertex of a cut: kind 0 = kept input vertex `src`; 1 = crossing of edge (src, src+1 mod k) with
 * Jacobian blocks wrt c = p[src] and d = p[src+1]; 2 = crossing whose lineCross bailed out (no gradient) */
typedef struct { int kind, src; double dxp_dxc, dyp_dxc, dxp_dyc, dyp_dyc, dxp_dxd, dyp_dxd, dxp_dyd, dyp_dyd; } cutrec;

static int gcut(dpt* p, int n, dpt a, dpt b, cutrec* rec_out, int* ref_overflow) {
  dpt pp[GCAP]; cutrec rr[GCAP];
  int m = 0, k = n;
  memset(pp, 0, sizeof(pp));
  p[n] = p[0];
  for (int i = 0; i < n; i++) {
    int si = gsig(gcross(a, b, p[i])), sj = gsig(gcross(a, b, p[i + 1]));
    if (si > 0) { pp[m] = p[i]; rr[m].kind = 0; rr[m].src = i; m++; }
    if (si != sj) {
      dpt c = p[i], d = p[i + 1];
      double s1 = gcross(a, b, c), s2 = gcross(a, b, d);
      double ds1_dxc = -(b.y - a.y), ds1_dyc = b.x - a.x, ds2_dxd = ds1_dxc, ds2_dyd = ds1_dyc;
      double s2_s1_2 = (s2 - s1) * (s2 - s1);
      rr[m].kind = 2; rr[m].src = i;
      if (!(gsig(s1) == 0 && gsig(s2) == 0) && gsig(s2 - s1) != 0) {
        rr[m].kind = 1;
        rr[m].dxp_dxc = ((s2 - d.x * ds1_dxc) * (s2 - s1) - (c.x * s2 - d.x * s1) * (-ds1_dxc)) / s2_s1_2;
        rr[m].dxp_dyc = ((0 - d.x * ds1_dyc) * (s2 - s1) - (c.x * s2 - d.x * s1) * (-ds1_dyc)) / s2_s1_2;
        rr[m].dxp_dxd = ((c.x * ds2_dxd - s1) * (s2 - s1) - (c.x * s2 - d.x * s1) * (ds2_dxd)) / s2_s1_2;
        rr[m].dxp_dyd = ((c.x * ds2_dyd - 0) * (s2 - s1) - (c.x * s2 - d.x * s1) * (ds2_dyd)) / s2_s1_2;
        rr[m].dyp_dxc = ((0 - d.y * ds1_dxc) * (s2 - s1) - (c.y * s2 - d.y * s1) * (-ds1_dxc)) / s2_s1_2;
        rr[m].dyp_dyc = ((s2 - d.y * ds1_dyc) * (s2 - s1) - (c.y * s2 - d.y * s1) * (-ds1_dyc)) / s2_s1_2;
        rr[m].dyp_dxd = ((c.y * ds2_dxd - 0) * (s2 - s1) - (c.y * s2 - d.y * s1) * (ds2_dxd)) / s2_s1_2;
        rr[m].dyp_dyd = ((c.y * ds2_dyd - s1) * (s2 - s1) - (c.y * s2 - d.y * s1) * (ds2_dyd)) / s2_s1_2;
        pp[m].x = (c.x * s2 - d.x * s1) / (s2 - s1);
        pp[m].y = (c.y * s2 - d.y * s1) / (s2 - s1);
      }
      m++;
    }
    if (m > GCAP - 3) break;
  }
  /* the reference indexes ccur_grad[100] / cut_grad[100] with 4*k*m + 4*i + 3 */
  if (m > 0 && 4 * k * (m - 1) + 4 * (k - 1) + 3 >= 100) *ref_overflow = 1;
  int nn = 0;
  for (int i = 0; i < m; i++)
    if (!i || !gsame(pp[i], pp[i - 1])) { p[nn] = pp[i]; rec_out[nn] = rr[i]; nn++; }
  while (nn > 1 && gsame(p[nn - 1], p[0])) nn--;
  if (nn > 5) *ref_overflow = 1;     /* p*_grad[10][10] */
  return nn;
}

/* pull a gradient on the cut's output vertices back to its k input vertices */
static void gcut_backward(const cutrec* rec, int n_out, int k, const double* g_out, double* g_in) {
  for (int i = 0; i < 2 * k; i++) g_in[i] = 0;
  for (int r = 0; r < n_out; r++) {
    double gx = g_out[2 * r], gy = g_out[2 * r + 1];
    if (rec[r].kind == 0) { g_in[2 * rec[r].src] += gx; g_in[2 * rec[r].src + 1] += gy; }
    else if (rec[r].kind == 1) {
      int c = rec[r].src, d = (rec[r].src + 1 == k) ? 0 : rec[r].src + 1;
      g_in[2 * c] += gx * rec[r].dxp_dxc + gy * rec[r].dyp_dxc;
      g_in[2 * c + 1] += gx * rec[r].dxp_dyc + gy * rec[r].dyp_dyc;
      g_in[2 * d] += gx * rec[r].dxp_dxd + gy * rec[r].dyp_dxd;
      g_in[2 * d + 1] += gx * rec[r].dxp_dyd + gy * rec[r].dyp_dyd;
    }
  }
}

/* d(signed area)/d(vertex v) of polygon ps[0..n) */
static void garea_grad(const dpt* ps, int n, double* g) {
  for (int v = 0; v < n; v++) {
    dpt prev = ps[(v + n - 1) % n], next = ps[(v + 1) % n];
    g[2 * v] = (-prev.y + next.y) / 2;
    g[2 * v + 1] = (prev.x + -next.x) / 2;
  }
}

static double gtri_term(dpt a, dpt b, dpt c, dpt d, double* grad_AB, int order, int convex_n, int* ref_overflow) {
  dpt o = {0, 0};
  int res_flag = 0;
  int s1 = gsig(gcross(o, a, b)), s2 = gsig(gcross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) { dpt t = a; a = b; b = t; res_flag = 1; }
  if (s2 == -1) { dpt t = c; c = d; d = t; }
  dpt p[GCAP + 1];
  cutrec r1[GCAP], r2[GCAP], r3[GCAP];
  p[0] = o; p[1] = a; p[2] = b;
  int n0 = 3;
  int n1 = gcut(p, n0, o, c, r1, ref_overflow);
  int n2 = gcut(p, n1, c, d, r2, ref_overflow);
  int n3 = gcut(p, n2, d, o, r3, ref_overflow);
  double res = garea(p, n3);
  double g3[2 * GCAP], g2[2 * GCAP], g1[2 * GCAP], g0[2 * GCAP];
  garea_grad(p, n3, g3);
  gcut_backward(r3, n3, n2, g3, g2);
  gcut_backward(r2, n2, n1, g2, g1);
  gcut_backward(r1, n1, n0, g1, g0);
  double sg = 1.0;
  if (s1 * s2 == -1) { sg = -1.0; res = -res; }
  /* g0[2..3] = wrt (possibly swapped) a, g0[4..5] = wrt b */
  double ga_x = sg * g0[2], ga_y = sg * g0[3], gb_x = sg * g0[4], gb_y = sg * g0[5];
  if (res_flag) { double t = ga_x; ga_x = gb_x; gb_x = t; t = ga_y; ga_y = gb_y; gb_y = t; }
  int nxt = (order != convex_n - 1) ? order + 1 : 0;
  grad_AB[2 * order] += ga_x; grad_AB[2 * order + 1] += ga_y;
  grad_AB[2 * nxt] += gb_x; grad_AB[2 * nxt + 1] += gb_y;
  return res;
}

/* Jarvis march over double points (convex_giou_kernel.cu:454-542 / 618-728), same rules as oracle/orp_hull.inc */
static int gjarvis(dpt* in_poly, int n_poly, int* to_input, int cap) {
  int n_input = n_poly;
  dpt input_poly[32];
  for (int i = 0; i < n_input; i++) input_poly[i] = in_poly[i];
  dpt p_max = in_poly[0], p_k;
  int max_index = 0, k_index, stack[64], top1, top2;
  dpt right_point[64], left_point[64];
  for (int i = 0; i < n_poly; i++) {
    if (in_poly[i].y < in_poly[0].y || (in_poly[i].y == in_poly[0].y && in_poly[i].x < in_poly[0].x)) {
      dpt t = in_poly[0]; in_poly[0] = in_poly[i]; in_poly[i] = t;
    }
    if (i == 0) { p_max = in_poly[0]; max_index = 0; }
    if (in_poly[i].y > p_max.y || (in_poly[i].y == p_max.y && in_poly[i].x > p_max.x)) { p_max = in_poly[i]; max_index = i; }
  }
  if (max_index == 0) { max_index = 1; p_max = in_poly[max_index]; }
  k_index = 0; stack[0] = 0; top1 = 0;
  while (k_index != max_index && top1 < cap) {
    p_k = p_max; k_index = max_index;
    for (int i = 1; i < n_poly; i++) {
      double sign = gcross(in_poly[stack[top1]], in_poly[i], p_k);
      if (sign > 0 || (sign == 0 && gdis(in_poly[stack[top1]], in_poly[i]) > gdis(in_poly[stack[top1]], p_k))) { p_k = in_poly[i]; k_index = i; }
    }
    top1++; stack[top1] = k_index;
  }
  for (int i = 0; i <= top1; i++) right_point[i] = in_poly[stack[i]];
  k_index = 0; stack[0] = 0; top2 = 0;
  while (k_index != max_index && top2 < cap) {
    p_k = p_max; k_index = max_index;
    for (int i = 1; i < n_poly; i++) {
      double sign = gcross(in_poly[stack[top2]], in_poly[i], p_k);
      if (sign < 0 || (sign == 0 && gdis(in_poly[stack[top2]], in_poly[i]) > gdis(in_poly[stack[top2]], p_k))) { p_k = in_poly[i]; k_index = i; }
    }
    top2++; stack[top2] = k_index;
  }
  for (int i = top2 - 1; i >= 0; i--) left_point[i] = in_poly[stack[i]];
  for (int i = 0; i < top1 + top2; i++) in_poly[i] = (i <= top1) ? right_point[i] : left_point[top2 - (i - top1)];
  n_poly = top1 + top2;
  if (to_input)
    for (int i = 0; i < n_poly; i++)
      for (int j = 0; j < n_input; j++)
        if (gsame(in_poly[i], input_poly[j])) { to_input[i] = j; break; }
  return n_poly;
}

/* one aligned pair: out19 = 18 grads (input point order) + giou.  returns 1 when the reference itself is undefined
 * for this row (scratch overflow) */
static int convex_giou_one(const float* p, const float* q, float* out19) {
  dpt convex[32], ps1[32], ps2[8];
  int to_input[32];
  int ref_overflow = 0;
  for (int i = 0; i < 32; i++) to_input[i] = -1;
  for (int i = 0; i < 9; i++) { convex[i].x = (double)p[2 * i]; convex[i].y = (double)p[2 * i + 1]; }
  int n1 = gjarvis(convex, 9, to_input, 9);
  int n2 = 4;
  for (int i = 0; i < n1; i++) ps1[i] = convex[i];
  for (int i = 0; i < 4; i++) { ps2[i].x = (double)q[2 * i]; ps2[i].y = (double)q[2 * i + 1]; }
  double grad_A[36], grad_AB[36], grad_C[36];
  memset(grad_A, 0, sizeof(grad_A)); memset(grad_AB, 0, sizeof(grad_AB)); memset(grad_C, 0, sizeof(grad_C));

  /* intersectAreaO: orient both CCW in place, then the 4*n1 signed triangle terms */
  if (garea(ps1, n1) < 0) for (int a = 0, b = n1 - 1; a < b; a++, b--) { dpt t = ps1[a]; ps1[a] = ps1[b]; ps1[b] = t; }
  if (garea(ps2, n2) < 0) for (int a = 0, b = n2 - 1; a < b; a++, b--) { dpt t = ps2[a]; ps2[a] = ps2[b]; ps2[b] = t; }
  ps1[n1] = ps1[0]; ps2[n2] = ps2[0];
  double inter = 0;
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++) inter += gtri_term(ps1[i], ps1[i + 1], ps2[j], ps2[j + 1], grad_AB, i, n1, &ref_overflow);

  /* S_pred and its gradient */
  double s_pred = garea(ps1, n1);
  garea_grad(ps1, n1, grad_A);
  if (s_pred < 0) for (int i = 0; i < 2 * n1; i++) grad_A[i] = -grad_A[i];
  double uni = fabs(s_pred) + fabs(garea(ps2, n2)) - inter;
  double iou = inter / uni;

  /* intersectAreaPoly: area of hull(ps1 U ps2) and its gradient wrt ps1 */
  {
    int n = n1 + n2, m2 = n2;
    for (int i = 0; i < n1; i++)
      for (int j = 0; j < n - n1; j++)
        if (gsame(ps1[i], ps2[j])) { for (int k = j; k < n - n1 - 1; k++) ps2[k] = ps2[k + 1]; m2--; break; }
    dpt poly[64];
    int n_poly = n1 + m2;
    for (int i = 0; i < n_poly; i++) poly[i] = (i < n1) ? ps1[i] : ps2[i - n1];
    n_poly = gjarvis(poly, n_poly, 0, 18);
    int map_hull[18], map_ps1[18], n_pred = 0;
    for (int i = 0; i < n_poly; i++)
      for (int j = 0; j < n1; j++)
        if (poly[i].x == ps1[j].x && poly[i].y == ps1[j].y) { if (n_pred < 18) { map_hull[n_pred] = i; map_ps1[n_pred] = j; n_pred++; } break; }
    double c_area = garea(poly, n_poly);
    if (n_pred > 0) {
      double gh[128];
      garea_grad(poly, n_poly, gh);
      /* polygen_area_grad's index loop: for each hull vertex the FIRST map entry that names it */
      for (int v = 0; v < n_poly; v++)
        for (int j = 0; j < n_pred; j++)
          if (map_hull[j] == v) { grad_C[2 * map_ps1[j]] = gh[2 * v]; grad_C[2 * map_ps1[j] + 1] = gh[2 * v + 1]; break; }
      if (c_area < 0) for (int i = 0; i < 18; i++) grad_C[i] = -grad_C[i];
    }
    c_area = fabs(c_area);
    double giou = iou - (c_area - uni) / c_area;
    float tmp[18];
    for (int i = 0; i < 18; i++) tmp[i] = 0.f;
    for (int i = 0; i < n1 && i < 9; i++) {
      int gp = to_input[i];
      if (gp < 0 || gp > 8) continue;
      for (int t = 0; t < 2; t++)
        tmp[2 * gp + t] = (float)((uni + inter) / (uni * uni) * grad_AB[2 * i + t] - iou / uni * grad_A[2 * i + t] -
                                  1 / c_area * (grad_AB[2 * i + t] - grad_A[2 * i + t]) - (uni) / c_area / c_area * grad_C[2 * i + t]);
    }
    for (int i = 0; i < 18; i++) out19[i] = tmp[i];
    out19[18] = (float)giou;
  }
  return ref_overflow;
}

void orc_convex_giou(const float* pts, const float* gts, int n, float* out19, int32_t* ref_undefined) {
  for (int i = 0; i < n; i++) {
    int f = convex_giou_one(pts + (size_t)i * 18, gts + (size_t)i * 8, out19 + (size_t)i * 19);
    if (ref_undefined) ref_undefined[i] = f;
  }
}
