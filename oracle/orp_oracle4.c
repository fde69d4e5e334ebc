/* oracle/orp_oracle4.c -- CPU ORACLE part 4 (TEST INFRASTRUCTURE ONLY): the assignment side of the APAA path.
 *   PointAssigner.assign                 mmdet/core/bbox/assigners/point_assigner.py:22-145
 *   MaxIoUAssigner.assign_wrt_overlaps   mmdet/core/bbox/assigners/max_iou_assigner.py:88-152
 *   get_adaptive_points_feature + feature_cosine_similarity   orientedreppoints_head.py:495-520, 576-600
 *   point_samples_selection              orientedreppoints_head.py:602-671
 * Sequential restatements (the reference's Python loops, one gt at a time).  Pinned against the reference's own
 * Python executed on CPU: tests/golden/apaa_py.npz (tests/golden/make_golden_py.py).
 * Ties the reference leaves to torch.topk / sort / max: smaller index first.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_point_assign(const float* points, int n, const float* gts, int k, float scale, int pos_num, int64_t* gt_inds) {
  for (int i = 0; i < n; i++) gt_inds[i] = 0;
  if (n == 0 || k == 0) return;
  int lvl_min = 1 << 30, lvl_max = -(1 << 30);
  for (int i = 0; i < n; i++) { int l = (int)log2f(points[3 * i + 2]); if (l < lvl_min) lvl_min = l; if (l > lvl_max) lvl_max = l; }
  float* assigned_dist = (float*)malloc(sizeof(float) * n);
  float* d = (float*)malloc(sizeof(float) * n);
  unsigned char* used = (unsigned char*)malloc(n);
  for (int i = 0; i < n; i++) assigned_dist[i] = INFINITY;
  for (int g = 0; g < k; g++) {
    const float* q = gts + 8 * (size_t)g;
    float xmin = q[0], xmax = q[0], ymin = q[1], ymax = q[1];
    for (int t = 1; t < 4; t++) {
      if (q[2 * t] < xmin) xmin = q[2 * t]; if (q[2 * t] > xmax) xmax = q[2 * t];
      if (q[2 * t + 1] < ymin) ymin = q[2 * t + 1]; if (q[2 * t + 1] > ymax) ymax = q[2 * t + 1];
    }
    float cx = (xmin + xmax) / 2, cy = (ymin + ymax) / 2;
    float w = xmax - xmin, h = ymax - ymin;
    if (w < 1e-6f) w = 1e-6f; if (h < 1e-6f) h = 1e-6f;
    int lvl = (int)((log2f(w / scale) + log2f(h / scale)) / 2);
    if (lvl < lvl_min) lvl = lvl_min; if (lvl > lvl_max) lvl = lvl_max;
    memset(used, 0, n);
    for (int i = 0; i < n; i++) {
      if ((int)log2f(points[3 * i + 2]) != lvl) { d[i] = -1; continue; }
      float dx = (points[3 * i] - cx) / w, dy = (points[3 * i + 1] - cy) / h;
      d[i] = sqrtf(dx * dx + dy * dy);
    }
    for (int r = 0; r < pos_num; r++) {        /* topk(pos_num, largest=False): smallest distance, then smallest index */
      int best = -1;
      for (int i = 0; i < n; i++) if (d[i] >= 0 && !used[i] && (best < 0 || d[i] < d[best])) best = i;
      if (best < 0) break;
      used[best] = 1;
      if (d[best] < assigned_dist[best]) { gt_inds[best] = g + 1; assigned_dist[best] = d[best]; }
    }
  }
  free(assigned_dist); free(d); free(used);
}

/* overlaps point-major [n,k]; torch.max semantics (NaN wins, first index on ties) */
static int better_f(float v, float cur) { return (v > cur) || (v != v && cur == cur); }
void orc_max_iou_assign(const float* ov, int n, int k, float pos_thr, float neg_lo, float neg_hi, float min_pos_iou,
                        int assign_all, int64_t* gt_inds, float* max_overlaps) {
  if (k == 0) { for (int i = 0; i < n; i++) { gt_inds[i] = 0; if (max_overlaps) max_overlaps[i] = 0; } return; }
  float* gt_max = (float*)malloc(sizeof(float) * k);
  int* gt_arg = (int*)malloc(sizeof(int) * k);
  for (int g = 0; g < k; g++) {
    float m = ov[g]; int a = 0;
    for (int i = 1; i < n; i++) { float v = ov[(size_t)i * k + g]; if (better_f(v, m)) { m = v; a = i; } }
    gt_max[g] = m; gt_arg[g] = a;
  }
  for (int i = 0; i < n; i++) {
    float m = ov[(size_t)i * k]; int arg = 0;
    for (int g = 1; g < k; g++) { float v = ov[(size_t)i * k + g]; if (better_f(v, m)) { m = v; arg = g; } }
    int64_t a = -1;
    if (m >= neg_lo && m < neg_hi) a = 0;
    if (m >= pos_thr) a = arg + 1;
    gt_inds[i] = a;
    if (max_overlaps) max_overlaps[i] = m;
  }
  for (int g = 0; g < k; g++) {
    if (!(gt_max[g] >= min_pos_iou)) continue;
    if (assign_all) { for (int i = 0; i < n; i++) if (ov[(size_t)i * k + g] == gt_max[g]) gt_inds[i] = g + 1; }
    else gt_inds[gt_arg[g]] = g + 1;
  }
  free(gt_max); free(gt_arg);
}

/* F.grid_sample(bilinear, zeros, align_corners=False) of feat [C,H,W] at image-space point (x,y), image = (W*stride, H*stride) */
static void sample_point(const float* feat, int C, int H, int W, float stride, float x, float y, float* out) {
  float ww = (float)W * stride, hh = (float)H * stride;
  float gx = x / (ww / 2.f) - 1.f, gy = y / (hh / 2.f) - 1.f;
  float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  float x0f = floorf(ix), y0f = floorf(iy);
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  float wx1 = ix - x0f, wx0 = 1.f - wx1, wy1 = iy - y0f, wy0 = 1.f - wy1;
  for (int c = 0; c < C; c++) {
    const float* f = feat + (size_t)c * H * W;
    float v = 0.f;
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) v += f[y0 * W + x0] * (wx0 * wy0);
    if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) v += f[y0 * W + x1] * (wx1 * wy0);
    if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) v += f[y1 * W + x0] * (wx0 * wy1);
    if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) v += f[y1 * W + x1] * (wx1 * wy1);
    out[c] = v;
  }
}
/* sampled [p,9,C] (for pinning get_adaptive_points_feature) */
void orc_sample_points(const float* feat, int C, int H, int W, float stride, const float* pts18, int p, float* out) {
  for (int i = 0; i < p; i++)
    for (int t = 0; t < 9; t++)
      sample_point(feat, C, H, W, stride, pts18[(size_t)i * 18 + 2 * t], pts18[(size_t)i * 18 + 2 * t + 1], out + ((size_t)i * 9 + t) * C);
}
/* feature_cosine_similarity on [p,9,C] */
void orc_feature_dissimilarity(const float* f, int p, int C, float* out) {
  float* mean = (float*)malloc(sizeof(float) * C);
  for (int i = 0; i < p; i++) {
    const float* fi = f + (size_t)i * 9 * C;
    double nm = 0;
    for (int c = 0; c < C; c++) { float s = 0; for (int t = 0; t < 9; t++) s += fi[t * C + c]; mean[c] = s / 9.f; nm += (double)mean[c] * mean[c]; }
    float norm_m = (float)sqrt(nm), cm = norm_m < 1e-2f ? 1e-2f : norm_m;
    float worst = -INFINITY;
    for (int t = 0; t < 9; t++) {
      double nk = 0, dk = 0;
      for (int c = 0; c < C; c++) { nk += (double)fi[t * C + c] * fi[t * C + c]; dk += (double)fi[t * C + c] * mean[c]; }
      float norm_k = (float)sqrt(nk), ck = norm_k < 1e-2f ? 1e-2f : norm_k;
      float uv = (float)dk / (ck * cm), nu = norm_k / ck, nv = norm_m / cm;
      float cs = uv / ((nu > 1e-6f ? nu : 1e-6f) * (nv > 1e-6f ? nv : 1e-6f));
      if (1.f - cs > worst) worst = 1.f - cs;
    }
    out[i] = worst;
  }
  free(mean);
}

/* point_samples_selection core: keep flags over the P positives */
void orc_apaa_select(const float* q, const int64_t* pos_gt, const int32_t* pos_lvl, int p, int num_gt, int num_level,
                     int per_level_k, double top_ratio, uint8_t* keep) {
  memset(keep, 0, p);
  unsigned char* used = (unsigned char*)malloc(p > 0 ? p : 1);
  for (int g = 1; g <= num_gt; g++) {
    float cq[256]; int ci[256]; int n = 0;
    memset(used, 0, p);
    for (int lv = 0; lv < num_level; lv++)
      for (int r = 0; r < per_level_k; r++) {
        int best = -1;
        for (int i = 0; i < p; i++)
          if (pos_gt[i] == g && pos_lvl[i] == lv && !used[i] && (best < 0 || q[i] < q[best])) best = i;
        if (best < 0) break;
        used[best] = 1; cq[n] = q[best]; ci[n] = best; n++;
      }
    if (n < 2) { for (int a = 0; a < n; a++) keep[ci[a]] = 1; continue; }
    for (int a = 1; a < n; a++) {           /* stable ascending sort */
      float vq = cq[a]; int vi = ci[a]; int b = a - 1;
      while (b >= 0 && cq[b] > vq) { cq[b + 1] = cq[b]; ci[b + 1] = ci[b]; b--; }
      cq[b + 1] = vq; ci[b + 1] = vi;
    }
    int topk = (int)ceil((double)n * top_ratio);
    for (int a = 0; a < topk && a < n; a++) keep[ci[a]] = 1;
  }
  free(used);
}
