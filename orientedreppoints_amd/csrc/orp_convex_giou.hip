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
constexpr int kPW = 1;            // pairs per wave (measured round 4: 4 pairs per wave = 112 us for 5 000 pairs against 85 -- the serial parts are branchy, lanes on different pairs diverge and their paths serialise)
constexpr int kPL = kThreads / kPW;   // lanes per pair

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

// One half-plane cut of polygon in[0..n) by the line a->b (keep the left side), written to out[]; rec[r] = (kind << 8) | src
// of output vertex r: kind 0 = input vertex src kept, 1 = crossing of edge (src, src + 1) with the line, 2 = degenerate
// crossing (no gradient).  The 2 x 2 Jacobians of a crossing w.r.t. its edge's end points are NOT stored: the backward pass
// recomputes them from the stage's input polygon with the very same expressions (the per-lane state drops from 2.9 KB of
// scratch -- which throttled the kernel to ~3 waves per CU -- to 0.7 KB).
__device__ int cut(const D2* in, int n, D2 a, D2 b, D2* out, int* rec) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const D2 c = in[i], d = in[(i + 1 < n) ? i + 1 : 0];
    const double s1 = crs(a, b, c), s2 = crs(a, b, d);
    const int si = sg(s1), sj = sg(s2);
    if (si > 0 && m < CAP) { out[m] = c; rec[m] = i; m++; }
    if (si != sj && m < CAP) {
      D2 x; x.x = 0; x.y = 0;
      int kind = 2;
      if (!(si == 0 && sj == 0) && sg(s2 - s1) != 0) {
        const double den = s2 - s1;
        const double nx = c.x * s2 - d.x * s1, ny = c.y * s2 - d.y * s1;
        kind = 1;
        x.x = nx / den; x.y = ny / den;
      }
      out[m] = x; rec[m] = (kind << 8) | i; m++;
    }
  }
  int nn = 0;
  for (int i = 0; i < m; i++)
    if (!i || !same(out[i], out[i - 1])) { out[nn] = out[i]; rec[nn] = rec[i]; nn++; }
  while (nn > 1 && same(out[nn - 1], out[0])) nn--;
  return nn;
}

// pull the gradient w.r.t. the cut's output vertices back to its k input vertices
__device__ void cut_backward(const D2* in, int k, D2 a, D2 b, const int* rec, int n_out, const double* g_out, double* g_in) {
  for (int i = 0; i < 2 * k; i++) g_in[i] = 0;
  for (int r = 0; r < n_out; r++) {
    const double gx = g_out[2 * r], gy = g_out[2 * r + 1];
    const int kind = rec[r] >> 8, src = rec[r] & 255;
    if (kind == 0) { g_in[2 * src] += gx; g_in[2 * src + 1] += gy; }
    else if (kind == 1) {
      const int ci = src, di = (src + 1 == k) ? 0 : src + 1;
      const D2 c = in[ci], d = in[di];
      const double s1 = crs(a, b, c), s2 = crs(a, b, d);
      const double k1x = -(b.y - a.y), k1y = b.x - a.x;        // ds1/dxc = ds2/dxd, ds1/dyc = ds2/dyd
      const double den = s2 - s1, den2 = den * den;
      const double nx = c.x * s2 - d.x * s1, ny = c.y * s2 - d.y * s1;
      const double j0 = ((s2 - d.x * k1x) * den - nx * (-k1x)) / den2;   // dxp_dxc
      const double j2 = ((0 - d.x * k1y) * den - nx * (-k1y)) / den2;    // dxp_dyc
      const double j4 = ((c.x * k1x - s1) * den - nx * (k1x)) / den2;    // dxp_dxd
      const double j6 = ((c.x * k1y - 0) * den - nx * (k1y)) / den2;     // dxp_dyd
      const double j1 = ((0 - d.y * k1x) * den - ny * (-k1x)) / den2;    // dyp_dxc
      const double j3 = ((s2 - d.y * k1y) * den - ny * (-k1y)) / den2;   // dyp_dyc
      const double j5 = ((c.y * k1x - 0) * den - ny * (k1x)) / den2;     // dyp_dxd
      const double j7 = ((c.y * k1y - s1) * den - ny * (k1y)) / den2;    // dyp_dyd
      g_in[2 * ci] += gx * j0 + gy * j1;
      g_in[2 * ci + 1] += gx * j2 + gy * j3;
      g_in[2 * di] += gx * j4 + gy * j5;
      g_in[2 * di + 1] += gx * j6 + gy * j7;
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
  D2 q0[3], q1[CAP], q2[CAP], q3[CAP];
  int r1[CAP], r2[CAP], r3[CAP];
  q0[0] = o; q0[1] = a; q0[2] = b;
  const int n1 = cut(q0, 3, o, c, q1, r1);
  const int n2 = cut(q1, n1, c, d, q2, r2);
  const int n3 = cut(q2, n2, d, o, q3, r3);
  double res = area_of(q3, n3);
  double ga[2 * CAP], gb[2 * CAP];
  area_grad(q3, n3, ga);
  cut_backward(q2, n2, d, o, r3, n3, ga, gb);
  cut_backward(q1, n1, c, d, r2, n2, gb, ga);
  cut_backward(q0, 3, o, c, r1, n1, ga, gb);
  double sgn = 1.0;
  if (s1 * s2 == -1) { sgn = -1.0; res = -res; }
  double gax = sgn * gb[2], gay = sgn * gb[3], gbx = sgn * gb[4], gby = sgn * gb[5];
  if (swapped) { double t = gax; gax = gbx; gbx = t; t = gay; gay = gby; gby = t; }
  g4[0] = gax; g4[1] = gay; g4[2] = gbx; g4[3] = gby;
  return res;
}

// Jarvis march, reference order and tie rules (convex_giou_kernel.cu:454-542, 618-728); chains capped
// (input_poly / right_point / left_point: HCAP-vertex work arrays in LDS -- one lane runs this, private arrays would be
//  scratch memory with a DRAM-like latency on every dynamically indexed access)
__device__ int jarvis(D2* in_poly, int n_poly, int* to_input, int cap, D2* input_poly, D2* right_point, D2* left_point) {
  const int n_input = n_poly;
  for (int i = 0; i < n_input; i++) input_poly[i] = in_poly[i];
  D2 p_max = in_poly[0], p_k;
  int max_index = 0, k_index;
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

// kPW (point set, gt) pairs per wave.  A pair's first lane builds the hull and orients both polygons (LDS), then the <= 36
// fan terms (hull edge i, gt edge j) run on the pair's lanes -- each with its clip records and reverse-mode pull-back --
// and park their value + 4 gradient components in LDS; the first lane accumulates them in the reference's (i outer, j
// inner) order, so values and gradients are what the serial loop produced, and finishes union / enclosing hull / GIoU.
// History: one THREAD per pair (36 serial terms on private arrays) ran a few thousand positives on a few dozen waves for
// 0.6 ms; one WAVE per pair with 2.9 KB of scratch per lane (clip records with their 2 x 2 Jacobians, the first lane's
// hull work arrays): 237 us for 5 000 pairs at ~3 resident waves per CU; round 4: Jacobians recomputed in the pull-back,
// the first lane's arrays in LDS, 0.75 KB of scratch: 85 us.  A single wave takes ~50 us (two serial hull marches around
// the parallel term phase); several pairs per wave (kPW > 1) were measured and are SLOWER (see kPW).
__global__ void __launch_bounds__(kThreads)
convex_giou_kernel(const float* __restrict__ pts, const float* __restrict__ gts, int n, float* __restrict__ out19) {
  // kPW pairs per wave, kPL = 64 / kPW lanes each: the serial parts (hull marches, finish) of kPW pairs run side by side on
  // the groups' first lanes, the <= 36 fan terms of a pair take ceil(36 / kPL) rounds on its kPL lanes.
  const int grp = threadIdx.x / kPL, lane = threadIdx.x % kPL;
  const int idx = blockIdx.x * kPW + grp;
  const bool valid = idx < n;
  __shared__ D2 sh_ps1[kPW][HCAP];
  __shared__ D2 sh_ps2[kPW][5];
  __shared__ int sh_to_input[kPW][HCAP];
  __shared__ int sh_n1[kPW];
  __shared__ double sh_val[kPW][36];
  __shared__ double sh_g4[kPW][36][4];
  __shared__ D2 sh_w0[kPW][HCAP], sh_w1[kPW][HCAP], sh_w2[kPW][HCAP], sh_poly[kPW][HCAP];   // the first lane's work arrays
  __shared__ double sh_gA[kPW][18], sh_gAB[kPW][20], sh_gC[kPW][18], sh_gh[kPW][2 * HCAP];
  D2* s_ps1 = sh_ps1[grp]; D2* s_ps2 = sh_ps2[grp]; int* s_to_input = sh_to_input[grp];
  double* s_val = sh_val[grp]; double (*s_g4)[4] = sh_g4[grp];
  D2* s_w0 = sh_w0[grp]; D2* s_w1 = sh_w1[grp]; D2* s_w2 = sh_w2[grp]; D2* s_poly = sh_poly[grp];
  double* s_gA = sh_gA[grp]; double* s_gAB = sh_gAB[grp]; double* s_gC = sh_gC[grp]; double* s_gh = sh_gh[grp];
  const float* p = pts + (size_t)(valid ? idx : 0) * 18;
  const float* q = gts + (size_t)(valid ? idx : 0) * 8;
  const int n2 = 4;
  if (lane == 0) {
    D2* ps1 = s_ps1; D2* ps2 = s_ps2;
    for (int i = 0; i < HCAP; i++) s_to_input[i] = -1;
    for (int i = 0; i < 9; i++) { ps1[i].x = (double)p[2 * i]; ps1[i].y = (double)p[2 * i + 1]; }
    int n1 = jarvis(ps1, 9, s_to_input, 9, s_w0, s_w1, s_w2);
    if (n1 > 9) n1 = 9;
    for (int i = 0; i < 4; i++) { ps2[i].x = (double)q[2 * i]; ps2[i].y = (double)q[2 * i + 1]; }
    if (area_of(ps1, n1) < 0) for (int a = 0, b = n1 - 1; a < b; a++, b--) { D2 t = ps1[a]; ps1[a] = ps1[b]; ps1[b] = t; }
    if (area_of(ps2, n2) < 0) for (int a = 0, b = n2 - 1; a < b; a++, b--) { D2 t = ps2[a]; ps2[a] = ps2[b]; ps2[b] = t; }
    for (int i = n1; i < HCAP; i++) s_ps1[i] = ps1[0];
    sh_n1[grp] = n1;
  }
  __syncthreads();
  const int n1 = sh_n1[grp];
  for (int t = lane; t < 4 * n1 && t < 36; t += kPL) {
    const int i = t >> 2, j = t & 3;
    double g4[4];
    const double v = tri_term_grad4(s_ps1[i], s_ps1[(i + 1 < n1) ? i + 1 : 0], s_ps2[j], s_ps2[(j + 1 < n2) ? j + 1 : 0], g4);
    s_val[t] = v;
    s_g4[t][0] = g4[0]; s_g4[t][1] = g4[1]; s_g4[t][2] = g4[2]; s_g4[t][3] = g4[3];
  }
  __syncthreads();
  if (lane != 0 || !valid) return;

  D2* ps1 = s_ps1; D2* ps2 = s_ps2;
  double* grad_A = s_gA; double* grad_AB = s_gAB; double* grad_C = s_gC;
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
  D2* poly = s_poly;
  int n_poly = n1 + m2;
  for (int i = 0; i < n_poly; i++) poly[i] = (i < n1) ? ps1[i] : ps2[i - n1];
  n_poly = jarvis(poly, n_poly, nullptr, 18, s_w0, s_w1, s_w2);
  double c_area = area_of(poly, n_poly);
  {
    double* gh = s_gh;
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
  hipLaunchKernelGGL(convex_giou_kernel, dim3((n + kPW - 1) / kPW), dim3(kThreads), 0, (hipStream_t)stream, pts, gts, n, out19);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
