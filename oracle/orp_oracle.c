/* oracle/orp_oracle.c -- CPU ORACLE: TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's algorithms for the Oriented RepPoints dense-head hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (orientedreppoints_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Parity pin: the reference ships NO tests or golden vectors for this path, so this
 * restatement is pinned against the reference's OWN device functions host-compiled by oracle/build_ref.py
 * (oracle/_ref/libref_orp.so) in tests/test_oracle_vs_ref.py, and against tests/golden/ fixtures generated from
 * that library (tests/golden/make_golden.py), plus the one implied known answer polyiou.cpp:131-132 -> 1/7.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared oracle/orp_oracle.c -o oracle/liborp_oracle.so -lm
 * (-ffp-contract=off: fp32 operation order is part of the contract for bit-exact NMS decisions.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------- */
/* polygon clipping core in three flavours                                                                  */
/* ------------------------------------------------------------------------------------------------------- */
#define REAL float
#define NAME(x) f32_##x
#define ABS_TERM 1
#include "orp_polyclip.inc"
#undef REAL
#undef NAME
#undef ABS_TERM

#define REAL double
#define NAME(x) f64_##x
#define ABS_TERM 1
#include "orp_polyclip.inc"
#undef REAL
#undef NAME
#undef ABS_TERM

#define REAL double
#define NAME(x) cvx_##x
#define ABS_TERM 0
#include "orp_polyclip.inc"
#undef REAL
#undef NAME
#undef ABS_TERM

/* hulls */
#define REAL float
#define HPT f32_pt
#define HNAME(x) hf32_##x
#include "orp_hull.inc"
#undef REAL
#undef HPT
#undef HNAME

#define REAL double
#define HPT cvx_pt
#define HNAME(x) hf64_##x
#include "orp_hull.inc"
#undef REAL
#undef HPT
#undef HNAME

/* debug statistics: largest intermediate polygon seen by the clipper, and whether the scratch cap was hit */
static int g_max_clip_n = 0, g_clip_overflow = 0;
int orc_stat_max_clip_n(void) { return g_max_clip_n; }
int orc_stat_clip_overflow(void) { return g_clip_overflow; }
void orc_stat_reset(void) { g_max_clip_n = 0; g_clip_overflow = 0; }

/* ------------------------------------------------------------------------------------------------------- */
/* a6/a8: fp32 quad-quad IoU  (rnms_kernel.cu:131-147 devrIoU; poly_nms_kernel.cu:192-212 devPolyIoU)        */
/* ------------------------------------------------------------------------------------------------------- */
static float quad_iou_f32(const float* p, const float* q, int zero_union_guard) {
  f32_pt ps1[8], ps2[8];
  for (int i = 0; i < 4; i++) {
    ps1[i].x = p[2 * i]; ps1[i].y = p[2 * i + 1];
    ps2[i].x = q[2 * i]; ps2[i].y = q[2 * i + 1];
  }
  float inter = f32_inter(ps1, 4, ps2, 4, &g_max_clip_n, &g_clip_overflow);
  float uni = (float)fabs((double)f32_area(ps1, 4)) + (float)fabs((double)f32_area(ps2, 4)) - inter;
  if (zero_union_guard && uni == 0) return (inter + 1) / (uni + 1);   /* poly_nms_kernel.cu:205-207 */
  return inter / uni;
}
float orc_quad_iou_f32(const float* p, const float* q) { return quad_iou_f32(p, q, 0); }
float orc_poly_nms_iou_f32(const float* p, const float* q) { return quad_iou_f32(p, q, 1); }

void orc_quad_iou_matrix_f32(const float* a, int n, const float* b, int k, int stride, int guard, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = quad_iou_f32(a + (size_t)i * stride, b + (size_t)j * stride, guard);
}

/* a7: fp64 quad-quad IoU (DOTA_devkit/polyiou.cpp:108-128 iou_poly) */
double orc_polyiou_f64(const double* p, const double* q) {
  f64_pt ps1[8], ps2[8];
  for (int i = 0; i < 4; i++) {
    ps1[i].x = p[2 * i]; ps1[i].y = p[2 * i + 1];
    ps2[i].x = q[2 * i]; ps2[i].y = q[2 * i + 1];
  }
  double inter = f64_inter(ps1, 4, ps2, 4, &g_max_clip_n, &g_clip_overflow);
  double uni = fabs(f64_area(ps1, 4)) + fabs(f64_area(ps2, 4)) - inter;
  return inter / uni;
}

/* ------------------------------------------------------------------------------------------------------- */
/* greedy NMS                                                                                               */
/* ------------------------------------------------------------------------------------------------------- */
/* dets already sorted by score; box i suppresses a LATER box j iff IoU(i, j) > thr with (row=i, col=j) argument
 * order (rnms_kernel.cu:183-197 tile rule + host sweep :245-257; poly_nms_kernel.cu:246-262,308-321).
 * keep_pos receives positions in the sorted order; returns their count. */
int orc_nms_sorted_f32(const float* dets, int n, int stride, float thr, int guard, int32_t* keep_pos) {
  unsigned char* removed = (unsigned char*)calloc(n > 0 ? n : 1, 1);
  int nk = 0;
  for (int i = 0; i < n; i++) {
    if (removed[i]) continue;
    keep_pos[nk++] = i;
    for (int j = i + 1; j < n; j++) {
      if (removed[j]) continue;
      if (quad_iou_f32(dets + (size_t)i * stride, dets + (size_t)j * stride, guard) > thr) removed[j] = 1;
    }
  }
  free(removed);
  return nk;
}

typedef struct { float score; int32_t idx; } orc_sk;
static int cmp_score_desc_idx_asc(const void* a, const void* b) {
  const orc_sk *x = (const orc_sk*)a, *y = (const orc_sk*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* stable order = (score desc, index asc): SURVEY.md A3 (the reference's Tensor.sort order on ties is unspecified) */
void orc_sort_order(const float* scores, int n, int stride, int32_t* order) {
  orc_sk* k = (orc_sk*)malloc(sizeof(orc_sk) * (n > 0 ? n : 1));
  for (int i = 0; i < n; i++) { k[i].score = scores[(size_t)i * stride]; k[i].idx = i; }
  qsort(k, n, sizeof(orc_sk), cmp_score_desc_idx_asc);
  for (int i = 0; i < n; i++) order[i] = k[i].idx;
  free(k);
}

/* a6: rnms_cuda (rnms_kernel.cu:204-265): sort, sweep, return ORIGINAL indices ascending */
static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}
int orc_rnms(const float* dets, int n, float thr, int64_t* keep_out) {
  if (n <= 0) return 0;
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* kp = (int32_t*)malloc(sizeof(int32_t) * n);
  float* sorted = (float*)malloc(sizeof(float) * 9 * (size_t)n);
  orc_sort_order(dets + 8, n, 9, order);
  for (int i = 0; i < n; i++) memcpy(sorted + (size_t)i * 9, dets + (size_t)order[i] * 9, 9 * sizeof(float));
  int nk = orc_nms_sorted_f32(sorted, n, 9, thr, 0, kp);
  for (int i = 0; i < nk; i++) keep_out[i] = order[kp[i]];
  qsort(keep_out, nk, sizeof(int64_t), cmp_i64);
  free(order); free(kp); free(sorted);
  return nk;
}

/* a7: py_cpu_nms_poly (DOTA_devkit/ResultMerge.py:18-41) over fp64 dets given an explicit visiting order
 * (the caller passes numpy's scores.argsort()[::-1], exactly the reference's order); keep is in visit order.
 * The reference keeps j iff iou_poly(kept, j) <= thresh. */
int orc_py_cpu_nms_poly(const double* dets, int n, const int64_t* order, double thr, int64_t* keep_out) {
  unsigned char* removed = (unsigned char*)calloc(n > 0 ? n : 1, 1);
  int nk = 0;
  for (int a = 0; a < n; a++) {
    int64_t i = order[a];
    if (removed[a]) continue;
    keep_out[nk++] = i;
    for (int b = a + 1; b < n; b++) {
      if (removed[b]) continue;
      double v = orc_polyiou_f64(dets + (size_t)i * 9, dets + (size_t)order[b] * 9);
      if (!(v <= thr)) removed[b] = 1;
    }
  }
  free(removed);
  return nk;
}

/* py_cpu_nms_poly_fast (DOTA_devkit/ResultMerge_multi_process.py:60-121): fp64 polyiou only for pairs whose horizontal
 * bounding boxes overlap; every other pair keeps its HBB overlap of 0.  Same visiting order convention as above. */
int orc_py_cpu_nms_poly_fast(const double* dets, int n, const int64_t* order, double thr, int64_t* keep_out) {
  unsigned char* removed = (unsigned char*)calloc(n > 0 ? n : 1, 1);
  double* bb = (double*)malloc(sizeof(double) * 5 * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    const double* d = dets + (size_t)i * 9;
    double xa = fmin(fmin(d[0], d[2]), fmin(d[4], d[6])), xb = fmax(fmax(d[0], d[2]), fmax(d[4], d[6]));
    double ya = fmin(fmin(d[1], d[3]), fmin(d[5], d[7])), yb = fmax(fmax(d[1], d[3]), fmax(d[5], d[7]));
    bb[5 * i] = xa; bb[5 * i + 1] = ya; bb[5 * i + 2] = xb; bb[5 * i + 3] = yb; bb[5 * i + 4] = (xb - xa + 1) * (yb - ya + 1);
  }
  int nk = 0;
  for (int a = 0; a < n; a++) {
    if (removed[a]) continue;
    int64_t i = order[a];
    keep_out[nk++] = i;
    for (int b = a + 1; b < n; b++) {
      if (removed[b]) continue;
      int64_t j = order[b];
      double w = fmax(0.0, fmin(bb[5 * i + 2], bb[5 * j + 2]) - fmax(bb[5 * i], bb[5 * j]));
      double h = fmax(0.0, fmin(bb[5 * i + 3], bb[5 * j + 3]) - fmax(bb[5 * i + 1], bb[5 * j + 1]));
      double hi = w * h;
      double ovr = hi / (bb[5 * i + 4] + bb[5 * j + 4] - hi);
      if (ovr > 0) ovr = orc_polyiou_f64(dets + (size_t)i * 9, dets + (size_t)j * 9);
      if (!(ovr <= thr)) removed[b] = 1;
    }
  }
  free(bb);
  free(removed);
  return nk;
}

/* ------------------------------------------------------------------------------------------------------- */
/* a8: poly_overlaps over (cx,cy,w,h,theta) boxes  (poly_overlaps_kernel.cu:280-328)                          */
/* ------------------------------------------------------------------------------------------------------- */
static void rotbox2poly(const float* dbox, f32_pt* ps) {
  float cs = cosf(dbox[4]);
  float ss = sinf(dbox[4]);
  float w = dbox[2], h = dbox[3];
  float x_ctr = dbox[0], y_ctr = dbox[1];
  /* mixed precision exactly as written in the reference: (w / 2.0) is double, the sum is rounded once */
  ps[0].x = (float)(x_ctr + cs * (w / 2.0) - ss * (-h / 2.0));
  ps[1].x = (float)(x_ctr + cs * (w / 2.0) - ss * (h / 2.0));
  ps[2].x = (float)(x_ctr + cs * (-w / 2.0) - ss * (h / 2.0));
  ps[3].x = (float)(x_ctr + cs * (-w / 2.0) - ss * (-h / 2.0));
  ps[0].y = (float)(y_ctr + ss * (w / 2.0) + cs * (-h / 2.0));
  ps[1].y = (float)(y_ctr + ss * (w / 2.0) + cs * (h / 2.0));
  ps[2].y = (float)(y_ctr + ss * (-w / 2.0) + cs * (h / 2.0));
  ps[3].y = (float)(y_ctr + ss * (-w / 2.0) + cs * (-h / 2.0));
}
void orc_rotbox2poly(const float* boxes, int n, float* polys8) {
  for (int i = 0; i < n; i++) {
    f32_pt ps[4];
    rotbox2poly(boxes + i * 5, ps);
    for (int k = 0; k < 4; k++) { polys8[i * 8 + 2 * k] = ps[k].x; polys8[i * 8 + 2 * k + 1] = ps[k].y; }
  }
}
void orc_poly_overlaps(const float* boxes, int n, const float* query, int k, float* out) {
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < k; j++) {
      f32_pt ps1[8], ps2[8];
      rotbox2poly(boxes + i * 5, ps1);
      rotbox2poly(query + j * 5, ps2);
      float inter = f32_inter(ps1, 4, ps2, 4, &g_max_clip_n, &g_clip_overflow);
      float uni = (float)fabs((double)f32_area(ps1, 4)) + (float)fabs((double)f32_area(ps2, 4)) - inter;
      out[(size_t)i * k + j] = (uni == 0) ? (inter + 1) / (uni + 1) : inter / uni;
    }
  }
}

/* ------------------------------------------------------------------------------------------------------- */
/* a4: minaerarect  (minarearect_kernel.cu:52-211 minBoundingRect, :343-452 Findminbox)                       */
/* ------------------------------------------------------------------------------------------------------- */
static float g_mbr_second_area;   /* second-smallest candidate area of the last min_bounding_rect call (tie diagnostics) */
static void min_bounding_rect(const f32_pt* ps, int n_points, float* minbox) {
  float edges_angles[24], unique_angles[24];
  const float pi = 3.1415926f;
  int n_edges = n_points - 1, n_unique = 0;
  for (int i = 0; i < n_edges; i++) {
    float ex = ps[i + 1].x - ps[i].x, ey = ps[i + 1].y - ps[i].y;
    float ang = (float)atan2((double)ey, (double)ex);
    if (ang >= 0) ang = (float)fmod((double)ang, (double)pi / 2);
    else ang = ang - (int)(ang / (pi / 2) - 1) * (pi / 2);
    edges_angles[i] = ang;
  }
  unique_angles[0] = edges_angles[0];
  n_unique = 1;
  for (int i = 1; i < n_edges; i++) {
    int dup = 0;
    for (int j = 0; j < n_unique; j++) if (edges_angles[i] == unique_angles[j]) dup++;
    if (!dup) unique_angles[n_unique++] = edges_angles[i];
  }
  float minarea = 1e12f;
  for (int i = 0; i < n_unique; i++) {
    float R00 = cosf(unique_angles[i]);
    float R01 = cosf(unique_angles[i] - pi / 2);
    float R10 = cosf(unique_angles[i] + pi / 2);
    float R11 = cosf(unique_angles[i]);
    float xmin = 1e12f, ymin = 1e12f, xmax = -1e12f, ymax = -1e12f;
    for (int j = 0; j < n_points; j++) {
      float rx = 0.0f, ry = 0.0f;
      rx = rx + R00 * ps[j].x; rx = rx + R01 * ps[j].y;
      ry = ry + R10 * ps[j].x; ry = ry + R11 * ps[j].y;
      if (!(isinf(rx) || isnan(rx))) { if (rx < xmin) xmin = rx; if (rx > xmax) xmax = rx; }
      if (!(isinf(ry) || isnan(ry))) { if (ry < ymin) ymin = ry; if (ry > ymax) ymax = ry; }
    }
    float area = (xmax - xmin) * (ymax - ymin);
    if (i == 0) g_mbr_second_area = 1e30f;
    if (area < minarea) { if (minarea < g_mbr_second_area && i > 0) g_mbr_second_area = minarea; }
    else if (area < g_mbr_second_area) g_mbr_second_area = area;
    if (area < minarea) {
      minarea = area;
      minbox[0] = unique_angles[i]; minbox[1] = xmin; minbox[2] = ymin; minbox[3] = xmax; minbox[4] = ymax;
    }
  }
}

static void find_min_box(const float* p, float* out8) {
  f32_pt convex[24], ps1[24];
  const float pi = 3.1415926f;
  int to_input[20];
  for (int i = 0; i < 9; i++) { convex[i].x = p[2 * i]; convex[i].y = p[2 * i + 1]; }
  int n1 = hf32_jarvis(convex, 9, to_input);
  for (int i = 0; i < n1; i++) ps1[i] = convex[i];
  ps1[n1] = convex[0];
  float mb[5] = {0, 0, 0, 0, 0};
  min_bounding_rect(ps1, n1 + 1, mb);
  float angle = mb[0], xmin = mb[1], ymin = mb[2], xmax = mb[3], ymax = mb[4];
  float R00 = cosf(angle), R01 = cosf(angle - pi / 2), R10 = cosf(angle + pi / 2), R11 = cosf(angle);
  /* corners (xmax,ymin),(xmin,ymin),(xmin,ymax),(xmax,ymax) as ROW vectors times R */
  const float cx[4] = {xmax, xmin, xmin, xmax}, cy[4] = {ymin, ymin, ymax, ymax};
  for (int c = 0; c < 4; c++) {
    float s0 = 0.0f, s1 = 0.0f;
    s0 = s0 + cx[c] * R00; s0 = s0 + cy[c] * R10;
    s1 = s1 + cx[c] * R01; s1 = s1 + cy[c] * R11;
    out8[2 * c] = s0; out8[2 * c + 1] = s1;
  }
}
/* Tie diagnostics (not a reference function): relative gap between the smallest and the second-smallest candidate
 * rectangle area of each point set.  The reference keeps the FIRST strict minimum (minarearect_kernel.cu:176-186), so a
 * gap at rounding level means the returned rectangle depends on the last ulp of cos / atan2. */
void orc_minarearect_margin(const float* pts, int n, float* margin) {
  float out8[8];
  for (int i = 0; i < n; i++) {
    find_min_box(pts + (size_t)i * 18, out8);
    float xs[4] = {out8[0], out8[2], out8[4], out8[6]}, ys[4] = {out8[1], out8[3], out8[5], out8[7]};
    /* area of the returned rectangle = |cross of two adjacent sides| */
    float a = fabsf((xs[1] - xs[0]) * (ys[2] - ys[1]) - (ys[1] - ys[0]) * (xs[2] - xs[1]));
    float second = g_mbr_second_area;
    margin[i] = (second >= 1e29f) ? INFINITY : (second - a) / fmaxf(a, 1e-12f);
  }
}
void orc_minarearect(const float* pts, int n, float* out) {
  for (int i = 0; i < n; i++) find_min_box(pts + (size_t)i * 18, out + (size_t)i * 8);
}

/* ------------------------------------------------------------------------------------------------------- */
/* a10: convex_iou  (convex_iou_kernel.cu:268-295 devrIoU; kernel :298-312 writes out[n][k])                   */
/* ------------------------------------------------------------------------------------------------------- */
static float convex_iou_one(const float* p, const float* q) {
  cvx_pt convex[24], ps1[24], ps2[8];
  int to_input[20];
  for (int i = 0; i < 9; i++) { convex[i].x = (double)p[2 * i]; convex[i].y = (double)p[2 * i + 1]; }
  int n1 = hf64_jarvis(convex, 9, to_input);
  for (int i = 0; i < n1; i++) ps1[i] = convex[i];
  for (int i = 0; i < 4; i++) { ps2[i].x = (double)q[2 * i]; ps2[i].y = (double)q[2 * i + 1]; }
  double inter = cvx_inter(ps1, n1, ps2, 4, &g_max_clip_n, &g_clip_overflow);
  double s_pred = cvx_area(ps1, n1);
  double uni = fabs(s_pred) + fabs(cvx_area(ps2, 4)) - inter;
  return (float)(inter / uni);
}
void orc_convex_iou(const float* pts, int n, const float* gts, int k, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++) out[(size_t)i * k + j] = convex_iou_one(pts + (size_t)i * 18, gts + (size_t)j * 8);
}
