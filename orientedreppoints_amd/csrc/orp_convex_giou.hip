// orp_convex_giou.hip -- GIoU(hull(9 points), gt quad) and its analytic gradient w.r.t. the 18 point coordinates,
// aligned pairs, fp64 internals, for gfx950.
//
// Replaces convex_giou_kernel + convex_giou_cuda (mmdet/ops/iou/src/convex_giou_kernel.cu:730-868), the kernel
// behind GIoULossFuction.forward (mmdet/models/losses/iou_loss.py:74).  The reference (a) keeps >10 KB of private
// arrays per thread (three dense [2n x 2k] Jacobians per polygon cut, multiplied p3_p2 * p2_p1 * p1_p for each of the
// 36 triangle terms), (b) raw cudaMalloc / malloc / blocking copies per call.  Here the gradient is REVERSE-MODE: the
// area gradient of the clipped polygon is pulled back through the three cuts vertex by vertex (each output vertex
// depends on at most two input vertices), which needs 8 doubles per clipped vertex instead of dense matrices, and the
// result is written straight into the caller's [P,19] buffer on the caller's stream.
// Values (giou) follow the reference's fp64 operation order; gradients agree to rounding (bar: 1e-4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orp_hip.h"
#include "orp_prof.hpp"

namespace {

struct D2 { double x, y; };
constexpr int CAP = 8;            // clipped-triangle vertices (<= 6 in exact arithmetic)
constexpr int HCAP = 20;          // hull of 9 + 4 points
constexpr int kThreads = 64;

__device__ __forceinline__ int sg(double d) { return (int)(d > 1E-8) - (int)(d < -1E-8); }
__device__ __forceinline__ bool same(D2 a, D2 b) { return sg(a.x - b.x) == 0 && sg(a.y - b.y) == 0; }
__device__ __forceinline__ double crs(D2 o, D2 a, D2 b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }
__device__ __forceinline__ double dis2(D2 a, D2 b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y); }

__device__ double area_of(const D2* ps, int n) {
  double res = 0;
  for (int i = 0; i < n; i++) { const D2 a = ps[i], b = ps[(i + 1 < n) ? i + 1 : 0]; res += a.x * b.y - a.y * b.x; }
  return res / 2.0;
}
__device__ void area_grad(const D2* ps, int n, double* g) {
  for (int v = 0; v < n; v++) {
    const D2 prev = ps[(v + n - 1) % n], next = ps[(v + 1) % n];
    g[2 * v] = (-prev.y + next.y) / 2;
    g[2 * v + 1] = (prev.x + -next.x) / 2;
  }
}

struct CutRec { int kind, src; double j[8]; };   // j = dxp_dxc, dyp_dxc, dxp_dyc, dyp_dyc, dxp_dxd, dyp_dxd, dxp_dyd, dyp_dyd

__device__ int cut(D2* p, int n, D2 a, D2 b, CutRec* rec) {
  D2 pp[CAP]; CutRec rr[CAP];
  int m = 0;
  for (int i = 0; i < n; i++) {
    const D2 c = p[i], d = p[(i + 1 < n) ? i + 1 : 0];
    const double s1 = crs(a, b, c), s2 = crs(a, b, d);
    const int si = sg(s1), sj = sg(s2);
    if (si > 0 && m < CAP) { pp[m] = c; rr[m].kind = 0; rr[m].src = i; m++; }
    if (si != sj && m < CAP) {
      rr[m].kind = 2; rr[m].src = i; pp[m].x = 0; pp[m].y = 0;
      if (!(si == 0 && sj == 0) && sg(s2 - s1) != 0) {
        const double k1x = -(b.y - a.y), k1y = b.x - a.x;        // ds1/dxc = ds2/dxd, ds1/dyc = ds2/dyd
        const double den = s2 - s1, den2 = den * den;
        const double nx = c.x * s2 - d.x * s1, ny = c.y * s2 - d.y * s1;
        rr[m].kind = 1;
        rr[m].j[0] = ((s2 - d.x * k1x) * den - nx * (-k1x)) / den2;   // dxp_dxc
        rr[m].j[2] = ((0 - d.x * k1y) * den - nx * (-k1y)) / den2;    // dxp_dyc
        rr[m].j[4] = ((c.x * k1x - s1) * den - nx * (k1x)) / den2;    // dxp_dxd
        rr[m].j[6] = ((c.x * k1y - 0) * den - nx * (k1y)) / den2;     // dxp_dyd
        rr[m].j[1] = ((0 - d.y * k1x) * den - ny * (-k1x)) / den2;    // dyp_dxc
        rr[m].j[3] = ((s2 - d.y * k1y) * den - ny * (-k1y)) / den2;   // dyp_dyc
        rr[m].j[5] = ((c.y * k1x - 0) * den - ny * (k1x)) / den2;     // dyp_dxd
        rr[m].j[7] = ((c.y * k1y - s1) * den - ny * (k1y)) / den2;    // dyp_dyd
        pp[m].x = nx / den; pp[m].y = ny / den;
      }
      m++;
    }
  }
  int nn = 0;
  for (int i = 0; i < m; i++)
    if (!i || !same(pp[i], pp[i - 1])) { p[nn] = pp[i]; rec[nn] = rr[i]; nn++; }
  while (nn > 1 && same(p[nn - 1], p[0])) nn--;
  return nn;
}

__device__ void cut_backward(const CutRec* rec, int n_out, int k, const double* g_out, double* g_in) {
  for (int i = 0; i < 2 * k; i++) g_in[i] = 0;
  for (int r = 0; r < n_out; r++) {
    const double gx = g_out[2 * r], gy = g_out[2 * r + 1];
    if (rec[r].kind == 0) { g_in[2 * rec[r].src] += gx; g_in[2 * rec[r].src + 1] += gy; }
    else if (rec[r].kind == 1) {
      const int c = rec[r].src, d = (rec[r].src + 1 == k) ? 0 : rec[r].src + 1;
      g_in[2 * c] += gx * rec[r].j[0] + gy * rec[r].j[1];
      g_in[2 * c + 1] += gx * rec[r].j[2] + gy * rec[r].j[3];
      g_in[2 * d] += gx * rec[r].j[4] + gy * rec[r].j[5];
      g_in[2 * d + 1] += gx * rec[r].j[6] + gy * rec[r].j[7];
    }
  }
}

// value of the fan term and its gradient w.r.t. the ORIGINAL (a, b): g4 = d/da.x, d/da.y, d/db.x, d/db.y (zeros on the
// early exits)
__device__ double tri_term_grad4(D2 a, D2 b, D2 c, D2 d, double* g4) {
  g4[0] = g4[1] = g4[2] = g4[3] = 0.0;
  D2 o; o.x = 0; o.y = 0;
  bool swapped = false;
  const int s1 = sg(crs(o, a, b)), s2 = sg(crs(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) { D2 t = a; a = b; b = t; swapped = true; }
  if (s2 == -1) { D2 t = c; c = d; d = t; }
  // exact zero (argument in orp_quadfast.hpp): if neither fan vertex is strictly left of O->c, at most 2 distinct
  // vertices survive the first cut -- the shoelace sum AND its gradient (area_grad of a <= 2-gon) are exactly 0, which
  // is what the full evaluation below would return.  About half of the 36 terms of an overlapping pair end here.
  {
    const double ca = c.x * a.y - a.x * c.y, cb = c.x * b.y - b.x * c.y;     // crs(o, c, a), crs(o, c, b)
    if (!(ca > 1E-8) && !(cb > 1E-8)) return 0.0;
  }
  D2 p[CAP];
  CutRec r1[CAP], r2[CAP], r3[CAP];
  p[0] = o; p[1] = a; p[2] = b;
  const int n1 = cut(p, 3, o, c, r1);
  const int n2 = cut(p, n1, c, d, r2);
  const int n3 = cut(p, n2, d, o, r3);
  double res = area_of(p, n3);
  double g3[2 * CAP], g2[2 * CAP], g1[2 * CAP], g0[6];
  area_grad(p, n3, g3);
  cut_backward(r3, n3, n2, g3, g2);
  cut_backward(r2, n2, n1, g2, g1);
  cut_backward(r1, n1, 3, g1, g0);
  double sgn = 1.0;
  if (s1 * s2 == -1) { sgn = -1.0; res = -res; }
  double gax = sgn * g0[2], gay = sgn * g0[3], gbx = sgn * g0[4], gby = sgn * g0[5];
  if (swapped) { double t = gax; gax = gbx; gbx = t; t = gay; gay = gby; gby = t; }
  g4[0] = gax; g4[1] = gay; g4[2] = gbx; g4[3] = gby;
  return res;
}

// Jarvis march, reference order and tie rules (convex_giou_kernel.cu:454-542, 618-728); chains capped
__device__ int jarvis(D2* in_poly, int n_poly, int* to_input, int cap) {
  const int n_input = n_poly;
  D2 input_poly[HCAP];
  for (int i = 0; i < n_input; i++) input_poly[i] = in_poly[i];
  D2 p_max = in_poly[0], p_k;
  int max_index = 0, k_index;
  D2 right_point[HCAP], left_point[HCAP];
  for (int i = 0; i < n_poly; i++) {
    if (in_poly[i].y < in_poly[0].y || (in_poly[i].y == in_poly[0].y && in_poly[i].x < in_poly[0].x)) {
      D2 t = in_poly[0]; in_poly[0] = in_poly[i]; in_poly[i] = t;
    }
    if (i == 0) { p_max = in_poly[0]; max_index = 0; }
    if (in_poly[i].y > p_max.y || (in_poly[i].y == p_max.y && in_poly[i].x > p_max.x)) { p_max = in_poly[i]; max_index = i; }
  }
  if (max_index == 0) { max_index = 1; p_max = in_poly[1]; }
  int top1 = 0, top2 = 0;
  D2 last = in_poly[0];
  right_point[0] = last; k_index = 0;
  while (k_index != max_index && top1 < cap) {
    p_k = p_max; k_index = max_index;
    for (int i = 1; i < n_poly; i++) {
      const double s = crs(last, in_poly[i], p_k);
      if (s > 0 || (s == 0 && dis2(last, in_poly[i]) > dis2(last, p_k))) { p_k = in_poly[i]; k_index = i; }
    }
    top1++; last = in_poly[k_index]; right_point[top1] = last;
  }
  last = in_poly[0]; left_point[0] = last; k_index = 0;
  while (k_index != max_index && top2 < cap) {
    p_k = p_max; k_index = max_index;
    for (int i = 1; i < n_poly; i++) {
      const double s = crs(last, in_poly[i], p_k);
      if (s < 0 || (s == 0 && dis2(last, in_poly[i]) > dis2(last, p_k))) { p_k = in_poly[i]; k_index = i; }
    }
    top2++; last = in_poly[k_index]; left_point[top2] = last;
  }
  for (int i = 0; i < top1 + top2 && i < HCAP; i++) in_poly[i] = (i <= top1) ? right_point[i] : left_point[top2 - (i - top1)];
  n_poly = top1 + top2; if (n_poly > HCAP) n_poly = HCAP;
  if (to_input)
    for (int i = 0; i < n_poly; i++)
      for (int j = 0; j < n_input; j++)
        if (same(in_poly[i], input_poly[j])) { to_input[i] = j; break; }
  return n_poly;
}

// One WAVE per (point set, gt) pair.  Lane 0 builds the hull and orients both polygons (LDS), then the <= 36 fan terms
// (hull edge i, gt edge j) run one per lane -- each with its clip records and reverse-mode pull-back -- and park their
// value + 4 gradient components in LDS; lane 0 accumulates them in the reference's (i outer, j inner) order, so values
// and gradients are what the serial loop produced, and finishes union / enclosing hull / GIoU.  The previous shape
// (one THREAD per pair, 36 serial terms on private arrays) left a call with a few thousand positives running on a few
// dozen waves for 0.6 ms.
__global__ void __launch_bounds__(kThreads)
convex_giou_kernel(const float* __restrict__ pts, const float* __restrict__ gts, int n, float* __restrict__ out19) {
  const int idx = blockIdx.x;
  const int lane = threadIdx.x;
  __shared__ D2 s_ps1[HCAP];
  __shared__ D2 s_ps2[5];
  __shared__ int s_to_input[HCAP];
  __shared__ int s_n1;
  __shared__ double s_val[36];
  __shared__ double s_g4[36][4];
  const float* p = pts + (size_t)idx * 18;
  const float* q = gts + (size_t)idx * 8;
  const int n2 = 4;
  if (lane == 0) {
    D2 ps1[HCAP], ps2[5];
    int to_input[HCAP];
    for (int i = 0; i < HCAP; i++) to_input[i] = -1;
    for (int i = 0; i < 9; i++) { ps1[i].x = (double)p[2 * i]; ps1[i].y = (double)p[2 * i + 1]; }
    int n1 = jarvis(ps1, 9, to_input, 9);
    if (n1 > 9) n1 = 9;
    for (int i = 0; i < 4; i++) { ps2[i].x = (double)q[2 * i]; ps2[i].y = (double)q[2 * i + 1]; }
    if (area_of(ps1, n1) < 0) for (int a = 0, b = n1 - 1; a < b; a++, b--) { D2 t = ps1[a]; ps1[a] = ps1[b]; ps1[b] = t; }
    if (area_of(ps2, n2) < 0) for (int a = 0, b = n2 - 1; a < b; a++, b--) { D2 t = ps2[a]; ps2[a] = ps2[b]; ps2[b] = t; }
    for (int i = 0; i < HCAP; i++) { s_ps1[i] = ps1[i < n1 ? i : 0]; s_to_input[i] = to_input[i]; }
    for (int i = 0; i < 4; i++) s_ps2[i] = ps2[i];
    s_n1 = n1;
  }
  __syncthreads();
  const int n1 = s_n1;
  if (lane < 4 * n1 && lane < 36) {
    const int i = lane >> 2, j = lane & 3;
    double g4[4];
    const double v = tri_term_grad4(s_ps1[i], s_ps1[(i + 1 < n1) ? i + 1 : 0], s_ps2[j], s_ps2[(j + 1 < n2) ? j + 1 : 0], g4);
    s_val[lane] = v;
    s_g4[lane][0] = g4[0]; s_g4[lane][1] = g4[1]; s_g4[lane][2] = g4[2]; s_g4[lane][3] = g4[3];
  }
  __syncthreads();
  if (lane != 0) return;

  D2 ps1[HCAP], ps2[5];
  for (int i = 0; i < n1; i++) ps1[i] = s_ps1[i];
  for (int i = 0; i < 4; i++) ps2[i] = s_ps2[i];
  double grad_A[18], grad_AB[20], grad_C[18];
  for (int i = 0; i < 18; i++) { grad_A[i] = 0; grad_AB[i] = 0; grad_C[i] = 0; }
  grad_AB[18] = grad_AB[19] = 0;
  double inter = 0;
  for (int i = 0; i < n1; i++) {
    const int nxt = (i != n1 - 1) ? i + 1 : 0;
    for (int j = 0; j < n2; j++) {
      const int t = i * 4 + j;
      inter += s_val[t];
      grad_AB[2 * i] += s_g4[t][0]; grad_AB[2 * i + 1] += s_g4[t][1];
      grad_AB[2 * nxt] += s_g4[t][2]; grad_AB[2 * nxt + 1] += s_g4[t][3];
    }
  }

  const double s_pred = area_of(ps1, n1);
  area_grad(ps1, n1, grad_A);
  if (s_pred < 0) for (int i = 0; i < 2 * n1; i++) grad_A[i] = -grad_A[i];
  const double uni = fabs(s_pred) + fabs(area_of(ps2, n2)) - inter;
  const double iou = inter / uni;

  // enclosing hull of ps1 U ps2 (gt vertices eps-equal to a hull vertex are dropped first), area + gradient wrt ps1
  int m2 = n2;
  ps2[4] = ps2[0];
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < 4; j++)
      if (same(ps1[i], ps2[j])) { for (int k = j; k < 3; k++) ps2[k] = ps2[k + 1]; m2--; break; }
  if (m2 < 0) m2 = 0;
  D2 poly[HCAP];
  int n_poly = n1 + m2;
  for (int i = 0; i < n_poly; i++) poly[i] = (i < n1) ? ps1[i] : ps2[i - n1];
  n_poly = jarvis(poly, n_poly, nullptr, 18);
  double c_area = area_of(poly, n_poly);
  {
    double gh[2 * HCAP];
    area_grad(poly, n_poly, gh);
    bool any = false;
    for (int v = 0; v < n_poly; v++)            // ascending: a later hull vertex naming the same ps1 point overwrites
      for (int j = 0; j < n1; j++)
        if (poly[v].x == ps1[j].x && poly[v].y == ps1[j].y) { grad_C[2 * j] = gh[2 * v]; grad_C[2 * j + 1] = gh[2 * v + 1]; any = true; break; }
    if (any && c_area < 0) for (int i = 0; i < 18; i++) grad_C[i] = -grad_C[i];
  }
  c_area = fabs(c_area);
  const double giou = iou - (c_area - uni) / c_area;

  float g[18];
  for (int i = 0; i < 18; i++) g[i] = 0.f;
  for (int i = 0; i < n1; i++) {
    const int gp = s_to_input[i];
    if (gp < 0 || gp > 8) continue;
    for (int t = 0; t < 2; t++)
      g[2 * gp + t] = (float)((uni + inter) / (uni * uni) * grad_AB[2 * i + t] - iou / uni * grad_A[2 * i + t] -
                              1 / c_area * (grad_AB[2 * i + t] - grad_A[2 * i + t]) - (uni) / c_area / c_area * grad_C[2 * i + t]);
  }
  float* o = out19 + (size_t)idx * 19;
  for (int i = 0; i < 18; i++) o[i] = g[i];
  o[18] = (float)giou;
}
}  // namespace

extern "C" int orp_convex_giou(const float* pts, const float* gts, int n, float* out19, void* stream) {
  if (n < 0 || (n > 0 && (!pts || !gts || !out19))) return ORP_EINVAL;
  if (n == 0) return ORP_OK;
  OrpProfScope prof(ORP_PROF_CONVEX_GIOU, (hipStream_t)stream);
  hipLaunchKernelGGL(convex_giou_kernel, dim3(n), dim3(kThreads), 0, (hipStream_t)stream, pts, gts, n, out19);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
